// LDS atomic throughput on gfx950: cycles per wave instruction for ds_add_f64 / ds_add_u64 / ds_add_u32 / ds_add_f32 / ds_write_b64 on random and on
// conflict-free addresses of a 64 KB array, 1024 threads per workgroup, one workgroup per CU (what k_bucket_sum does per record).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_atomics.hip -o gpurun_out/lds_atomics && gpurun_out/lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE, bool RANDOM>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, int iters) {
    __shared__ double acc[8192];
    for (int i = threadIdx.x; i < 8192; i += 1024) acc[i] = 0.0;
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 97u + 12345u;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        x = x * 1664525u + 1013904223u;
        const uint32_t slot = RANDOM ? (x >> 19) : ((threadIdx.x + it * 1024) & 8191);
        if (MODE == 0) atomicAdd(&acc[slot], 1.0);
        else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned long long*>(&acc[slot]), 1ull);
        else if (MODE == 2) atomicAdd(reinterpret_cast<unsigned int*>(&acc[slot]), 1u);
        else if (MODE == 3) atomicAdd(reinterpret_cast<float*>(&acc[slot]), 1.0f);
        else acc[slot] = (double)x;
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (acc[threadIdx.x] == 123.456) out[0] = 0;
}
template <int MODE, bool RANDOM>
void run(const char* name) {
    unsigned long long* d; hipMalloc(&d, 256 * 8);
    const int iters = 2000;
    hipLaunchKernelGGL((k<MODE, RANDOM>), dim3(256), dim3(1024), 0, 0, d, iters);
    hipLaunchKernelGGL((k<MODE, RANDOM>), dim3(256), dim3(1024), 0, 0, d, iters);
    hipDeviceSynchronize();
    unsigned long long h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 256; i++) s += (double)h[i];
    s /= 256;
    printf("%-14s %-13s: %8.0f cycles for %d ops x 16 waves -> %.2f cycles per wave instruction, %.2f lane-ops per cycle per CU\n", name, RANDOM ? "random" : "conflict-free", s, iters, s / (iters * 16.0), iters * 1024.0 / s);
    hipFree(d);
}
int main() {
    run<0, true>("ds_add_f64"); run<0, false>("ds_add_f64");
    run<1, true>("ds_add_u64"); run<1, false>("ds_add_u64");
    run<2, true>("ds_add_u32"); run<2, false>("ds_add_u32");
    run<3, true>("ds_add_f32"); run<3, false>("ds_add_f32");
    run<4, true>("ds_write_b64"); run<4, false>("ds_write_b64");
    return 0;
}
