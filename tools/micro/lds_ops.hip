// LDS operation cost on gfx950: cycles per wave-instruction for the random-access patterns of the scoring kernel
// (512-thread workgroups, 2 per CU like k_score_slices).   hipcc --offload-arch=gfx950 -O3 lds_ops.hip -o lds_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int HT = 1536, TPB = 512, ITER = 2000;

template <int OP>
__global__ __launch_bounds__(TPB, 4) void k(uint32_t seed, unsigned long long* out, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* d = reinterpret_cast<double*>(smem);                 // [2*HT]
    uint32_t* u = reinterpret_cast<uint32_t*>(d + 2 * HT);      // [2*HT]
    for (int i = threadIdx.x; i < 2 * HT; i += TPB) { d[i] = 0.0; u[i] = 0xFFFFFFFFu; }
    __syncthreads();
    uint32_t x = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    uint32_t acc = 0;
    double accd = 0.0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; it++) {
        x = x * 1664525u + 1013904223u;
        const uint32_t h = (x >> 8) % HT;                        // random slot
        const uint32_t lin = (threadIdx.x + it * 7) % HT;       // stride-1 slot
        if (OP == 0) atomicAdd(&d[2 * h + (x & 1)], 1.0);                         // random ds_add_f64
        if (OP == 1) d[2 * h + (x & 1)] = (double)x;                              // random ds_write_b64
        if (OP == 2) acc += atomicCAS(&u[h], 0xFFFFFFFFu, x | 1u);                // random ds_cmpst_rtn_b32
        if (OP == 3) atomicAdd(reinterpret_cast<float*>(&u[h]), 1.0f);            // random ds_add_f32
        if (OP == 4) reinterpret_cast<uint16_t*>(u)[2 * h + (x & 1)] = (uint16_t)x;   // random ds_write_b16
        if (OP == 5) *reinterpret_cast<double2*>(&d[2 * lin]) = make_double2(1.0, 2.0);   // stride-1 ds_write_b128
        if (OP == 6) accd += d[2 * h];                                            // random ds_read_b64
        if (OP == 7) { const double2 v = *reinterpret_cast<const double2*>(&d[2 * h]); accd += v.x + v.y; }   // random ds_read_b128
        if (OP == 8) { const double2 v = *reinterpret_cast<const double2*>(&d[2 * lin]); accd += v.x + v.y; } // stride-1 ds_read_b128
        if (OP == 9) acc += atomicCAS(reinterpret_cast<unsigned long long*>(&d[h]), ~0ull, (unsigned long long)x) != 0;   // random ds_cmpst_rtn_b64
        if (OP == 10) u[h] = x;                                                   // random ds_write_b32
        if (OP == 11) atomicMax(&u[h], x);                                        // random ds_max_u32 (no return)
        if (OP == 12) atomicAdd(&d[2 * lin], 1.0);                                // stride-1 ds_add_f64
        if (OP == 13) atomicAdd(&u[h], x & 255u);                                 // random ds_add_u32 (no return)
        if (OP == 14) acc += atomicAdd(&u[h], x & 255u);                          // random ds_add_rtn_u32
        if (OP == 15) acc += u[h];                                                // random ds_read_b32
        if (OP == 16) atomicOr(&u[h], x);                                         // random ds_or_b32 (no return)
        if (OP == 17) atomicAdd(reinterpret_cast<float*>(&u[lin]), 1.0f);         // stride-1 ds_add_f32
        if (OP == 18) atomicAdd(&u[lin], 1u);                                     // stride-1 ds_add_u32
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (acc == 0x12345678u || accd == 1.2345) sink[0] = acc;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char* name) {
    unsigned long long* out; uint32_t* sink;
    const int blocks = 256 * 2;
    hipMalloc(&out, blocks * 8); hipMalloc(&sink, 4);
    const size_t lds = 2 * HT * 8 + 2 * HT * 4 + 36000;          // ~72 KB: 2 workgroups per CU
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(TPB), lds, 0, 1u, out, sink);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(TPB), lds, 0, 2u, out, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), out, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    // cycles (of the 100 MHz-independent shader clock counter) per iteration of one wave; 16 waves share the CU's LDS
    printf("%-28s %8.1f counter ticks per op per thread-iteration (16 waves/CU resident)\n", name, s / blocks / ITER);
    hipFree(out); hipFree(sink);
}

int main() {
    run<0>("random ds_add_f64");
    run<12>("stride-1 ds_add_f64");
    run<1>("random ds_write_b64");
    run<10>("random ds_write_b32");
    run<4>("random ds_write_b16");
    run<2>("random ds_cmpst_rtn_b32");
    run<9>("random ds_cmpst_rtn_b64");
    run<3>("random ds_add_f32");
    run<17>("stride-1 ds_add_f32");
    run<11>("random ds_max_u32");
    run<13>("random ds_add_u32");
    run<18>("stride-1 ds_add_u32");
    run<14>("random ds_add_rtn_u32");
    run<16>("random ds_or_b32");
    run<15>("random ds_read_b32");
    run<6>("random ds_read_b64");
    run<7>("random ds_read_b128");
    run<8>("stride-1 ds_read_b128");
    run<5>("stride-1 ds_write_b128");
    return 0;
}
