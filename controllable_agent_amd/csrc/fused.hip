// head_kernel: the LAST layer of an embedding head in one launch (gfx950).
//
// Linear(H, z) of ForwardMap's F1 / F2 and of BackwardMap (fb_modules.py:78, :221), optionally followed by sqrt(d) * F.normalize
// (fb_modules.py:229): K = H is deep, N = z <= 64 is skinny.  The grouped GEMM of gemm.hip runs these as 32x32 tiles with the K
// range sliced across workgroups into a slab that a second launch (splitk_reduce_kernel) folds, and BackwardMap's projection is a
// third launch; every launch of this size costs its ~5 us fill / drain floor whatever it computes.  Here ONE workgroup owns 16
// batch rows and all N columns: its 8 waves split K (wave w contracts the 32-deep steps w, w + 8, ...: consecutive waves read
// adjacent 128-byte pieces of a row), the partial tiles are folded through LDS in wave order (deterministic), then bias, the row
// norm and the projection -- no slab in memory, no reduce launch, no l2norm launch.
//
// v_mfma_f32_16x16x4_f32 (exact fp32) with both operands loaded straight from global memory into MFMA fragments: X[rows, K] and
// W[N, K] are k-contiguous, a lane owns 8 consecutive k of one row (two float4 loads) and the 8 MFMAs of a step contract
// k = 32 t + 8 (lane >> 4) + m, m = 0..7, on both operands (any bijection of k works as long as A and B agree).  The product is
// computed TRANSPOSED (A = W rows, B = X rows), so an accumulator lane holds 4 consecutive output columns of one batch row: one
// float4 store.  Two register sets, ping-pong: the next step's fragments are in flight under the current step's MFMAs.
//
// Measured on MI355X (tools/fused_bench.py, profiles/r04a_fused_bench.txt), walker dims, B = 1024: four F heads (N = 50, K = 1024)
// 11.8 us against 12.8 + 5.2 (GEMM + reduce) inside the step; three B heads with the projection (K = 576) 8.8 us against
// 10.0 + 7.0 + 7.5.  At N = 100 (8 tiles per wave) it takes 25 us and loses: the launcher refuses N > 64 and the schedule keeps
// the grouped GEMM there.  Two things built beside it and NOT kept (round 4): staging the fragments through wave-private LDS patches
// from coalesced loads (14.4 us: more dependent stages per step, the direct loads were not the bound), and the same row ownership
// for the FIRST layers, Linear(in, H) + LayerNorm + tanh in one launch (23 us against 16 us for GEMM + LayerNorm on one trunk: the
// 64 workgroups of a trunk carry the whole tanh / store epilogue that the LayerNorm launch spreads over every CU).
#include "common.h"

namespace fbhip {

typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float comp(const float4& v, int m) { return m == 0 ? v.x : m == 1 ? v.y : m == 2 ? v.z : v.w; }

// acc[j] += W-fragment(j) x X-fragment over the 8 k of one step; T independent accumulators between two uses of the same one
template <int T>
__device__ __forceinline__ void mfma_step(floatx4 (&acc)[T], const float4 (&w0)[T], const float4 (&w1)[T], const float4& x0,
                                          const float4& x1) {
    // (hipcc's scheduler otherwise sinks every fragment load to just before its first use -- one load, a full s_waitcnt, four
    // DEPENDENT MFMAs, repeat: the fences pin "all loads of the next step, then T independent MFMAs per k")
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int j = 0; j < T; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(w0[j], m), comp(x0, m), acc[j], 0, 0, 0);
        if (T > 1) __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int j = 0; j < T; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(w1[j], m), comp(x1, m), acc[j], 0, 0, 0);
        if (T > 1) __builtin_amdgcn_sched_barrier(0);
    }
}

}  // namespace

// NTL = 16-column tiles of the output (N <= 16 NTL <= 64).  Waves 0 .. NTL-1 finalise one tile each.
template <int NTL>
__global__ void __launch_bounds__(512) head_kernel(const HeadGroup g) {
    const HeadProblem& p = g.p[blockIdx.y];
    const int rows = p.rows, N = p.N, K = p.K;
    const int row0 = blockIdx.x * 16;
    if (row0 >= rows) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int li = lane & 15, kk = lane >> 4;
    __shared__ __attribute__((aligned(16))) float red[8 * NTL * 256];        // [8 waves][NTL tiles][64 lanes] float4
    __shared__ float red2[4][16];

    // fragment sources; rows / columns past the edge read a CLAMPED (valid) address, their results are dropped at the store
    const float* xsrc = p.X + (size_t)min(row0 + li, rows - 1) * p.ldx;
    const float* wsrc[NTL];
#pragma unroll
    for (int j = 0; j < NTL; ++j) wsrc[j] = p.W + (size_t)min(16 * j + li, N - 1) * p.ldw;

    floatx4 acc[NTL];
#pragma unroll
    for (int j = 0; j < NTL; ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
    // the bias quad of the tile this wave finalises: requested now, used after the fold (not a round trip of its own at the end)
    const float4 bias4 = ldg4(p.bias + min(16 * wave + 4 * kk, ((N + 3) & ~3) - 4));

    const int KT = (K + 31) >> 5;
    {
        // K % 4 == 0: a quad starting below K lies inside the row; quads at or past K read a CLAMPED address and their X values are
        // zeroed when they are used (never at the load: a select on a value in flight would put the wait there).
        float4 xa0, xa1, wa0[NTL], wa1[NTL], xb0, xb1, wb0[NTL], wb1[NTL];
        auto load = [&](int t, float4& x0, float4& x1, float4 (&w0)[NTL], float4 (&w1)[NTL]) __attribute__((always_inline)) {
            const int k0 = min(32 * t + 8 * kk, K - 4), k1 = min(32 * t + 8 * kk + 4, K - 4);
            x0 = ldg4(xsrc + k0); x1 = ldg4(xsrc + k1);
#pragma unroll
            for (int j = 0; j < NTL; ++j) { w0[j] = ldg4(wsrc[j] + k0); w1[j] = ldg4(wsrc[j] + k1); }
        };
        auto step = [&](int t, const float4& x0, const float4& x1, const float4 (&w0)[NTL], const float4 (&w1)[NTL]) __attribute__((always_inline)) {
            const int k0 = 32 * t + 8 * kk;
            __builtin_amdgcn_sched_barrier(0);
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 y0 = k0 < K ? x0 : z, y1 = k0 + 4 < K ? x1 : z;
            mfma_step<NTL>(acc, w0, w1, y0, y1);
        };
        load(wave, xa0, xa1, wa0, wa1);
        for (int t = wave; t < KT; t += 16) {
            load(t + 8, xb0, xb1, wb0, wb1);
            step(t, xa0, xa1, wa0, wa1);
            if (t + 8 >= KT) break;
            load(t + 16, xa0, xa1, wa0, wa1);
            step(t + 8, xb0, xb1, wb0, wb1);
        }
    }

    // fold the 8 partial tiles of every output tile in wave order; wave w < NTL finalises tile w
#pragma unroll
    for (int j = 0; j < NTL; ++j)
        *reinterpret_cast<float4*>(red + ((wave * NTL + j) * 64 + lane) * 4) = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
    __syncthreads();
    // yv = y[row0 + li][c0 .. c0 + 3], c0 = 16 wave + 4 kk
    const int row = row0 + li, c0 = 16 * wave + 4 * kk;
    const bool mine = wave < NTL && c0 < N;
    float4 yv = make_float4(0.f, 0.f, 0.f, 0.f);
    float sq = 0.f;
    if (mine) {
        yv = *reinterpret_cast<const float4*>(red + ((0 * NTL + wave) * 64 + lane) * 4);
#pragma unroll
        for (int w = 1; w < 8; ++w) {
            const float4 u = *reinterpret_cast<const float4*>(red + ((w * NTL + wave) * 64 + lane) * 4);
            yv.x += u.x; yv.y += u.y; yv.z += u.z; yv.w += u.w;
        }
        const float4 b = bias4;                                           // (c0 < N here: the clamp above did not move it)
        yv.x += b.x;
        yv.y = c0 + 1 < N ? yv.y + b.y : 0.f;
        yv.z = c0 + 2 < N ? yv.z + b.z : 0.f;
        yv.w = c0 + 3 < N ? yv.w + b.w : 0.f;
        sq = (yv.x * yv.x + yv.y * yv.y) + (yv.z * yv.z + yv.w * yv.w);
    }
    float den = 1.f;
    if (p.out2 != nullptr) {                                    // sqrt(d) * F.normalize(y): the row norm spans the finalising waves
        sq += __shfl_xor(sq, 16);
        sq += __shfl_xor(sq, 32);
        if (wave < 4 && kk == 0) red2[wave][li] = sq;          // (waves >= NTL hold 0)
        __syncthreads();
        const float tot = (red2[0][li] + red2[1][li]) + (red2[2][li] + red2[3][li]);
        const float nrm = sqrtf(tot);
        den = fmaxf(nrm, 1e-12f);                               // F.normalize eps
        if (wave == 0 && kk == 0 && row < rows && p.norms != nullptr) p.norms[row] = nrm;
    }
    if (!mine || row >= rows) return;
    *reinterpret_cast<float4*>(p.C + (size_t)row * p.ldc + c0) = yv;
    if (p.out2 != nullptr)
        *reinterpret_cast<float4*>(p.out2 + (size_t)row * p.ldo + c0) =
            make_float4(p.scale * (yv.x / den), p.scale * (yv.y / den), p.scale * (yv.z / den), p.scale * (yv.w / den));
}

bool head_ok(const HeadProblem& p) {
    auto al = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    const int Np = (p.N + 3) & ~3;
    return p.rows >= 1 && p.N >= 1 && p.N <= 64 && p.K >= 4 && (p.K & 3) == 0 && p.ldx >= p.K && (p.ldx & 3) == 0 &&
           p.ldw >= p.K && (p.ldw & 3) == 0 && p.ldc >= Np && (p.ldc & 3) == 0 && al(p.X) && al(p.W) && al(p.bias) && al(p.C) &&
           p.bias != nullptr && (p.out2 == nullptr || (p.ldo >= Np && (p.ldo & 3) == 0 && al(p.out2)));
}

hipError_t launch_head_group(const HeadGroup& g, hipStream_t s) {
    if (g.n < 1) return hipSuccess;
    int maxrows = 0, maxN = 0;
    for (int i = 0; i < g.n; ++i) {
        if (!head_ok(g.p[i])) return hipErrorInvalidValue;
        maxrows = g.p[i].rows > maxrows ? g.p[i].rows : maxrows;
        maxN = g.p[i].N > maxN ? g.p[i].N : maxN;
    }
    dim3 grid((maxrows + 15) / 16, g.n), block(512);
    if (maxN <= 16) hipLaunchKernelGGL(head_kernel<1>, grid, block, 0, s, g);
    else if (maxN <= 32) hipLaunchKernelGGL(head_kernel<2>, grid, block, 0, s, g);
    else hipLaunchKernelGGL(head_kernel<4>, grid, block, 0, s, g);
    return hipGetLastError();
}

}  // namespace fbhip
