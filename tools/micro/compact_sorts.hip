// How to order / cut a candidate buffer of n <= 256 {u64 key, u32 doc} entries in LDS with a 256-thread workgroup (k_score_slices' topk_compact):
// the bitonic network (36 barrier-separated steps at 256 entries) against placement by counting (every entry counts the entries that precede it:
// 3 barriers), cycles per call of one workgroup alone on its CU, 64 calls back to back.
//   hipcc --offload-arch=gfx950 -O3 -w tools/micro/compact_sorts.hip -o /tmp/compact_sorts && /tmp/compact_sorts
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ bool better(uint64_t ka, uint32_t da, uint64_t kb, uint32_t db) { return ka > kb || (ka == kb && da < db); }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long* out, uint32_t n, int reps) {
    __shared__ uint64_t key[256];
    __shared__ uint32_t doc[256];
    __shared__ uint32_t place[256];
    const int tid = threadIdx.x;
    uint32_t x = tid * 2654435761u + 99u;
    unsigned long long tot = 0;
    for (int r = 0; r < reps; r++) {
        x = x * 1664525u + 1013904223u;
        key[tid] = ((uint64_t)x << 20) | (x >> 7); doc[tid] = x ^ 0x5555u;
        __syncthreads();
        const unsigned long long t0 = __builtin_readcyclecounter();
        if (MODE == 0) {
            uint32_t n2 = 64; while (n2 < n) n2 <<= 1;
            for (uint32_t i = n + tid; i < n2; i += 256) { key[i] = 0; doc[i] = 0xFFFFFFFFu; }
            lds_barrier();
            for (uint32_t size = 2; size <= n2; size <<= 1)
                for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
                    for (uint32_t i = tid; i < (n2 >> 1); i += 256) {
                        const uint32_t lo = 2 * i - (i & (stride - 1)), hi = lo + stride;
                        const bool desc = ((lo & size) == 0);
                        const uint64_t ka = key[lo], kb = key[hi]; const uint32_t da = doc[lo], db = doc[hi];
                        const bool swap = desc ? better(kb, db, ka, da) : better(ka, da, kb, db);
                        if (swap) { key[lo] = kb; key[hi] = ka; doc[lo] = db; doc[hi] = da; }
                    }
                    lds_barrier();
                }
        } else if (MODE == 1) {
            uint32_t n2 = 64; while (n2 < n) n2 <<= 1;
            place[tid] = 0;
            lds_barrier();
            const uint32_t i = tid & (n2 - 1), part = tid / n2, parts = 256 / n2;
            uint64_t mk = 0; uint32_t md = 0;
            if (i < n) {
                mk = key[i]; md = doc[i];
                const uint32_t per = (n + parts - 1) / parts, j0 = part * per, j1 = min(n, j0 + per);
                uint32_t cnt = 0;
                for (uint32_t j = j0; j < j1; j += 8) {
                    uint64_t kk[8]; uint32_t dd[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) { const uint32_t jj = min(j + u, j1 - 1); kk[u] = key[jj]; dd[u] = doc[jj]; }
#pragma unroll
                    for (int u = 0; u < 8; u++) cnt += (j + u < j1 && better(kk[u], dd[u], mk, md)) ? 1u : 0u;
                }
                if (cnt) atomicAdd(&place[i], cnt);
            }
            lds_barrier();
            if (tid < n) { const uint32_t pl = place[tid]; key[pl] = mk; doc[pl] = md; }
            lds_barrier();
        }
        if (MODE == 3) {
            // the same network, workgroup barriers only where a step crosses waves: thread i always handles pair i, so with strides <= 64 wave w
            // only ever touches entries 128w .. 128w+127 and its own LDS operations are ordered
            uint32_t n2 = 64; while (n2 < n) n2 <<= 1;
            for (uint32_t i = n + tid; i < n2; i += 256) { key[i] = 0; doc[i] = 0xFFFFFFFFu; }
            lds_barrier();
            for (uint32_t size = 2; size <= n2; size <<= 1)
                for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
                    for (uint32_t i = tid; i < (n2 >> 1); i += 256) {
                        const uint32_t lo = 2 * i - (i & (stride - 1)), hi = lo + stride;
                        const bool desc = ((lo & size) == 0);
                        const uint64_t ka = key[lo], kb = key[hi]; const uint32_t da = doc[lo], db = doc[hi];
                        const bool swap = desc ? better(kb, db, ka, da) : better(ka, da, kb, db);
                        if (swap) { key[lo] = kb; key[hi] = ka; doc[lo] = db; doc[hi] = da; }
                    }
                    const uint32_t next = stride > 1 ? (stride >> 1) : size;
                    if (stride >= 128 || next >= 128) lds_barrier(); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            lds_barrier();
        }
        if (MODE == 2) {
            // placement by counting with the compared entries BROADCAST from registers (v_readlane) instead of read from LDS by every lane:
            // wave w owns entries 64w .. 64w+63, a tile of 64 entries is loaded once per wave (one entry per lane) and handed round
            const int lane = tid & 63, wave = tid >> 6;
            const uint32_t i = 64u * wave + lane;
            const uint64_t mk = i < n ? key[i] : 0ull; const uint32_t md = i < n ? doc[i] : 0xFFFFFFFFu;
            uint32_t cnt = 0;
            const uint32_t tiles = (n + 63u) / 64u;
            for (uint32_t t = 0; t < tiles; t++) {
                const uint32_t j = 64u * t + lane;
                const uint64_t tk = j < n ? key[j] : 0ull; const uint32_t td = j < n ? doc[j] : 0xFFFFFFFFu;
                const uint32_t tlo = (uint32_t)tk, thi = (uint32_t)(tk >> 32);
#pragma unroll
                for (int u = 0; u < 64; u++) {
                    const uint64_t bk = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)thi, u) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)tlo, u);
                    const uint32_t bd = (uint32_t)__builtin_amdgcn_readlane((int)td, u);
                    const uint32_t bj = 64u * t + (uint32_t)u;
                    cnt += (better(bk, bd, mk, md) || (bk == mk && bd == md && bj < i)) ? 1u : 0u;
                }
            }
            lds_barrier();
            if (i < n) { key[cnt] = mk; doc[cnt] = md; }
            lds_barrier();
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        tot += t1 - t0;
        if (key[(tid + r) & 255] == 12345) out[1] = 1;
        __syncthreads();
    }
    if (tid == 0) out[0] = tot;
}
int main() {
    unsigned long long* d; hipMalloc(&d, 16);
    for (uint32_t n : {64u, 100u, 128u, 200u, 256u}) {
        unsigned long long h[2];
        hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), 0, 0, d, n, 64); hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), 0, 0, d, n, 64);
        hipDeviceSynchronize(); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        const double a = h[0] / 64.0;
        hipLaunchKernelGGL(k<1>, dim3(1), dim3(256), 0, 0, d, n, 64); hipLaunchKernelGGL(k<1>, dim3(1), dim3(256), 0, 0, d, n, 64);
        hipDeviceSynchronize(); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        const double b = h[0] / 64.0;
        hipLaunchKernelGGL(k<2>, dim3(1), dim3(256), 0, 0, d, n, 64); hipLaunchKernelGGL(k<2>, dim3(1), dim3(256), 0, 0, d, n, 64);
        hipDeviceSynchronize(); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        const double c = h[0] / 64.0;
        hipLaunchKernelGGL(k<3>, dim3(1), dim3(256), 0, 0, d, n, 64); hipLaunchKernelGGL(k<3>, dim3(1), dim3(256), 0, 0, d, n, 64);
        hipDeviceSynchronize(); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("n=%3u entries: bitonic network %7.0f cycles, placement by counting (LDS broadcast reads) %7.0f, (register tiles + v_readlane) %7.0f, bitonic with wave-local steps %7.0f\n", n, a, b, c, h[0] / 64.0);
    }
    return 0;
}
