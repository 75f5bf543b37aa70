// order.hpp — the result order of the retrieval path as one total order on (FinalRank, doc id):
// FinalRank descending (appendSort, retrieval/util.go:48-54), equal finals by ascending doc id (a valid
// linearisation of the reference's arrival-order-dependent ties), NaN finals last.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace ss {

// order-preserving map double -> u64: larger key = better; NaN lowest
__device__ __forceinline__ uint64_t fkey(double f) {
    if (f != f) return 0ull;
    const uint64_t b = (uint64_t)__double_as_longlong(f);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double funkey(uint64_t k) {
    const uint64_t b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __longlong_as_double((long long)b);
}
__device__ __forceinline__ bool better(uint64_t ka, uint32_t da, uint64_t kb, uint32_t db) {
    return ka > kb || (ka == kb && da < db);
}

}  // namespace ss
