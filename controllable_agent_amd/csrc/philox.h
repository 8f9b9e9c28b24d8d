// Philox4x32-10 counter-based RNG + Box-Muller (device side).  counter = (index, stream id, step counter, 0),
// key = (seed, rank): reproducible and independent of launch geometry.
#pragma once
#include <hip/hip_runtime.h>

namespace fbhip {

struct U4 { unsigned x, y, z, w; };

__device__ __forceinline__ U4 philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1) {
    const unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
        const unsigned hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        const unsigned n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    return U4{c0, c1, c2, c3};
}
__device__ __forceinline__ float u01(unsigned x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }   // (0,1)
__device__ __forceinline__ void box_muller(unsigned a, unsigned b, float& n0, float& n1) {
    const float r = sqrtf(-2.0f * logf(u01(a)));
    const float th = 6.283185307179586f * u01(b);
    n0 = r * cosf(th);
    n1 = r * sinf(th);
}

// stream ids (second counter word)
enum { STREAM_INDEX = 0, STREAM_PERM = 1, STREAM_MIX = 2, STREAM_Z = 3, STREAM_EPS_NEXT = 4, STREAM_EPS_ACTOR = 5,
       STREAM_ACT = 6, STREAM_FUTURE = 7, STREAM_ZU = 8, STREAM_RW = 9,
       STREAM_RWU = 10 };

}  // namespace fbhip
