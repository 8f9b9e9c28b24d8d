// The a-wide seams of the actor as 16-row MFMA tiles (gfx950), round 6.
//
// policy_head_kernel and actor_head_bwd_kernel (rowops.hip) give one WAVE one batch row and keep the a-wide weight slices
// ([a, H] of the policy head W4, [H, a] of the action columns of forward_net's first layer W1) in LDS: every workgroup of 4-8 rows
// refills a 48-98 KB image before it can start (quadruped: 1024 workgroups x 48 KB for the policy head, 256 x 96 KB for the
// backward seam) -- 47 us and 55 us per launch inside the step at B = 2048, a = 12 (profiles/r05z_quadruped_kernel_stats.txt)
// for bytes that stream in 3 us.  Here ONE workgroup owns 16 batch rows, the way head_kernel (fused.hip) does:
//   * the H-deep contractions (premu = p . W4^T;  d action = g . W1[:, action columns]) are v_mfma_f32_16x16x4_f32 with both
//     operands loaded straight from global memory / L2 as fragments, the 8 waves split K = H, partial tiles folded through
//     4 KB of LDS in wave order (deterministic);
//   * the a-deep rank updates (pre += W1[:, action] . action;  d p = d premu . W4) are the same MFMA with K = 16 >= a: each wave
//     owns H / 8 consecutive hidden units of the 16 rows, A fragments come from global, B fragments are the lane's OWN registers
//     (an accumulator lane holds 4 consecutive a-indices of one batch row = the k-bijection k -> 4 (lane >> 4) + step);
//   * LayerNorm statistics of a row span the 8 waves: wave-shuffle over the 4 lanes of a row, then an 8-way LDS fold.
// No weight image in LDS, no refill, 4x / 2x fewer workgroup-level passes over the weight slices.
// Math is rowops.hip's (fb_modules.py:112-126, :190; utils.py:171-185; LayerNorm + tanh as ln_tanh_fwd / bwd_kernel), the
// fp32 summation ORDER differs (MFMA k order, 8-way folds): covered by the same parity tolerances.
#include "common.h"

namespace fbhip {

typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float comp(const float4& v, int m) { return m == 0 ? v.x : m == 1 ? v.y : m == 2 ? v.z : v.w; }
__device__ __forceinline__ float quad_sum(float v) {             // over the 4 lanes (li, kk = 0..3) that share a batch row
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

// C[16 x 16] += W-rows x X-rows over K, K split over the 8 waves in 32-deep steps (head_kernel's scheme); returns this wave's
// partial tile: lane (li, kk) holds columns 4 kk .. 4 kk + 3 (rows of W) of batch row li (row of X)
__device__ __forceinline__ floatx4 deep_dot_partial(const float* __restrict__ xsrc, const float* __restrict__ wsrc, int K, int wave,
                                                    int kk) {
    floatx4 acc = floatx4{0.f, 0.f, 0.f, 0.f};
    const int KT = (K + 31) >> 5;
    float4 xa0, xa1, wa0, wa1, xb0, xb1, wb0, wb1;
    auto load = [&](int t, float4& x0, float4& x1, float4& w0, float4& w1) __attribute__((always_inline)) {
        const int k0 = min(32 * t + 8 * kk, K - 4), k1 = min(32 * t + 8 * kk + 4, K - 4);
        x0 = ldg4(xsrc + k0); x1 = ldg4(xsrc + k1);
        w0 = ldg4(wsrc + k0); w1 = ldg4(wsrc + k1);
    };
    auto step = [&](int t, const float4& x0, const float4& x1, const float4& w0, const float4& w1) __attribute__((always_inline)) {
        const int k0 = 32 * t + 8 * kk;
        __builtin_amdgcn_sched_barrier(0);
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 y0 = k0 < K ? x0 : z, y1 = k0 + 4 < K ? x1 : z;
#pragma unroll
        for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(w0, m), comp(y0, m), acc, 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(w1, m), comp(y1, m), acc, 0, 0, 0);
    };
    if (wave < KT) {
        load(wave, xa0, xa1, wa0, wa1);
        for (int t = wave; t < KT; t += 16) {
            load(min(t + 8, KT - 1), xb0, xb1, wb0, wb1);
            step(t, xa0, xa1, wa0, wa1);
            if (t + 8 >= KT) break;
            load(min(t + 16, KT - 1), xa0, xa1, wa0, wa1);
            step(t + 8, xb0, xb1, wb0, wb1);
        }
    }
    return acc;
}

// fold the 8 waves' partial tiles in wave order; EVERY wave gets the total (each needs it as the B operand of its rank update)
__device__ __forceinline__ float4 fold8(float* red /* [8][64] float4 */, const floatx4& acc, int wave, int lane) {
    *reinterpret_cast<float4*>(red + (wave * 64 + lane) * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    __syncthreads();
    float4 y = *reinterpret_cast<const float4*>(red + lane * 4);
#pragma unroll
    for (int w = 1; w < 8; ++w) {
        const float4 u = *reinterpret_cast<const float4*>(red + (w * 64 + lane) * 4);
        y.x += u.x; y.y += u.y; y.z += u.z; y.w += u.w;
    }
    return y;
}

// row statistic spanning the 8 waves: ``v`` is this lane's partial over its own elements of row li
__device__ __forceinline__ float row_total(float* red2 /* [8][16] */, float v, int wave, int li, int kk) {
    v = quad_sum(v);
    if (kk == 0) red2[wave * 16 + li] = v;
    __syncthreads();
    float t = red2[li];
#pragma unroll
    for (int w = 1; w < 8; ++w) t += red2[w * 16 + li];
    return t;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------
// policy head (+ optionally the first layer of the trunk that consumes the action): see policy_head_kernel (rowops.hip) for the
// contract of every operand.  TruncatedNormal actor only (na == a <= 16).  TPW = H / 128: 16-wide tiles of the hidden row per wave.
template <int TPW>
__global__ void __launch_bounds__(512) policy_head_tile_kernel(const PolicyHeadJobs jobs, const float* __restrict__ W4, int ldw4,
                                                               const float* __restrict__ b4, int ldpre, int ldn, float stddev,
                                                               float clip, int ldmu, int rows, int H, int a) {
    const PolicyHeadJob& jb = jobs.j[blockIdx.y];
    const int row0 = blockIdx.x * 16;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int li = lane & 15, kk = lane >> 4;
    const int row = min(row0 + li, rows - 1);
    const bool live = row0 + li < rows;
    const bool first = jb.base != nullptr;
    __shared__ __attribute__((aligned(16))) float red[8 * 64 * 4];
    __shared__ float red2[2][8 * 16];

    // ---- the first layer's operands: requested NOW, consumed after the head (one round trip for everything)
    const int nb = wave * 16 * TPW;                              // this wave's span of hidden units
    float4 pre[TPW], gm[TPW], bt[TPW];
    float w1[TPW][4];
    if (first) {
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int n0 = nb + 16 * t;
            pre[t] = ldg4(jb.base + (size_t)row * jb.ldb + n0 + 4 * kk);
            gm[t] = ldg4(jb.gamma + n0 + 4 * kk);
            bt[t] = ldg4(jb.beta + n0 + 4 * kk);
            // A fragment of the rank-a update: W1[n0 + li][aoff + 4 kk + m], m = 0..3 (columns past a: clamped, their B is zero)
            // (one 16-byte load where the quad lies inside the a columns -- 4-byte aligned only: the action columns start at obs_dim;
            //  gfx950 global loads need no natural alignment -- else clamped scalars)
            const float* q = jb.W1a + (size_t)(n0 + li) * jb.ldw1;
            if (4 * kk + 3 < a) {
                float4 v;
                __builtin_memcpy(&v, q + 4 * kk, 16);
                w1[t][0] = v.x; w1[t][1] = v.y; w1[t][2] = v.z; w1[t][3] = v.w;
            } else {
#pragma unroll
                for (int m = 0; m < 4; ++m) w1[t][m] = q[min(4 * kk + m, a - 1)];
            }
        }
    }
    float nz[4], bias[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int c = min(4 * kk + m, a - 1);
        bias[m] = b4[c];
        nz[m] = jb.noise != nullptr ? jb.noise[(size_t)row * ldn + c] : 0.f;
    }

    // ---- premu = p . W4^T + b4: 8 waves split K = H
    const floatx4 part = deep_dot_partial(jb.P + (size_t)row * jb.ldp, W4 + (size_t)min(li, a - 1) * ldw4, H, wave, kk);
    const float4 tot = fold8(red, part, wave, lane);
    float act[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int c = 4 * kk + m;
        const float v = comp(tot, m) + bias[m];
        const float mu = tanhf(v);                                 // Actor: mu = tanh(policy(h))   (fb_modules.py:124)
        float av = mu;
        if (jb.noise != nullptr) {                                 // TruncatedNormal.sample (utils.py:176-185)
            float e = nz[m] * stddev;
            if (clip >= 0.f) e = fminf(fmaxf(e, -clip), clip);
            const float lo = (float)(-1.0 + 1e-6), hi = (float)(1.0 - 1e-6);
            av = fminf(fmaxf(mu + e, lo), hi);
        }
        act[m] = c < a ? av : 0.f;
        if (wave == 0 && live && c < a) {
            jb.premu[(size_t)row * ldpre + c] = v;
            if (jb.mu != nullptr) jb.mu[(size_t)row * ldmu + c] = mu;
            if (jb.action != nullptr) jb.action[(size_t)row * jb.lda + c] = av;
        }
    }
    if (!first) return;                                            // (uniform per workgroup)

    // ---- pre += W1[:, action columns] . action   (K = 16 >= a: 4 MFMAs per 16 x 16 tile, B = this lane's own action quad)
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        floatx4 c4 = floatx4{pre[t].x, pre[t].y, pre[t].z, pre[t].w};
#pragma unroll
        for (int m = 0; m < 4; ++m) c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[t][m], act[m], c4, 0, 0, 0);
        pre[t] = make_float4(c4[0], c4[1], c4[2], c4[3]);
    }
    // ---- LayerNorm + tanh over the row (two passes, like ln_tanh_fwd_kernel: mean, then the centred squares)
    float sm = 0.f;
#pragma unroll
    for (int t = 0; t < TPW; ++t) sm += (pre[t].x + pre[t].y) + (pre[t].z + pre[t].w);
    const float mean = row_total(red2[0], sm, wave, li, kk) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const float4 cc = make_float4(pre[t].x - mean, pre[t].y - mean, pre[t].z - mean, pre[t].w - mean);
        q += (cc.x * cc.x + cc.y * cc.y) + (cc.z * cc.z + cc.w * cc.w);
    }
    const float var = row_total(red2[1], q, wave, li, kk) / (float)H;     // biased, like nn.LayerNorm
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    if (!live) return;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int n0 = nb + 16 * t + 4 * kk;
        float4 o;
        o.x = tanhf((pre[t].x - mean) * rstd * gm[t].x + bt[t].x);
        o.y = tanhf((pre[t].y - mean) * rstd * gm[t].y + bt[t].y);
        o.z = tanhf((pre[t].z - mean) * rstd * gm[t].z + bt[t].z);
        o.w = tanhf((pre[t].w - mean) * rstd * gm[t].w + bt[t].w);
        *reinterpret_cast<float4*>(jb.t1 + (size_t)row * jb.ldt1 + n0) = o;
        if (jb.stats != nullptr) *reinterpret_cast<float4*>(const_cast<float*>(jb.base) + (size_t)row * jb.ldb + n0) = pre[t];
    }
    if (jb.stats != nullptr && wave == 0 && kk == 0) {
        jb.stats[2 * row] = mean;
        jb.stats[2 * row + 1] = rstd;
    }
}

// FBHIP_HEAD_TILES (read at every launch: A/B runs and tests force either form): a bit mask, 1 = policy head, 2 = backward seam;
// unset = both where the launch fills the chip (>= 2048 batch rows: 128 workgroups of 16 rows per job; measured same-box, round 6:
// quadruped B = 2048 +1.9 %, walker B = 1024 -1.3 % -- 64 workgroups per job leave the serial latency of one tile chain exposed)
static bool tiles_on(int which, int rows) {
    const char* e = getenv("FBHIP_HEAD_TILES");
    if (e && e[0] >= '0' && e[0] <= '3') return ((e[0] - '0') & which) != 0;
    return rows >= 2048;
}

bool policy_head_tiles_ok(const PolicyHeadJobs& jobs, int ldw4, int rows, int H, int a, int na, const Squash& sq) {
    if (!tiles_on(1, rows) || sq.on || na != a || a < 1 || a > 16 || (H & 3) || H < 32 || (ldw4 & 3) || rows < 1) return false;
    for (int i = 0; i < jobs.n; ++i) {
        const PolicyHeadJob& j = jobs.j[i];
        if ((j.ldp & 3) || ((uintptr_t)j.P & 15)) return false;
        if (j.base != nullptr && !(H == 512 || H == 1024)) return false;      // (H = 2048: 16 tiles per wave spill; the row kernel stays)
    }
    return true;
}

hipError_t launch_policy_head_tiles(const PolicyHeadJobs& jobs, const float* W4, int ldw4, const float* b4, int ldpre, int ldn,
                                    float stddev, float clip, int ldmu, int rows, int H, int a, hipStream_t s) {
    if (((uintptr_t)W4 & 15)) return hipErrorInvalidValue;
    dim3 grid((rows + 15) / 16, jobs.n), block(512);
    bool first = false;
    for (int i = 0; i < jobs.n; ++i) first = first || jobs.j[i].base != nullptr;
    if (!first || H == 512) hipLaunchKernelGGL(policy_head_tile_kernel<4>, grid, block, 0, s, jobs, W4, ldw4, b4, ldpre, ldn, stddev, clip, ldmu, rows, H, a);
    else hipLaunchKernelGGL(policy_head_tile_kernel<8>, grid, block, 0, s, jobs, W4, ldw4, b4, ldpre, ldn, stddev, clip, ldmu, rows, H, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------
// d action -> d premu -> d p with the LayerNorm + tanh backward of forward_net's obs_action trunk in front: see
// actor_head_bwd_kernel (rowops.hip, LNE > 0) for the contract.  H = 128 TPW.
template <int TPW>
__global__ void __launch_bounds__(512) actor_head_bwd_tile_kernel(const float* __restrict__ dt1, int ldt, const float* __restrict__ lnY,
                                                                  int ldy, const float* __restrict__ lnX, int ldx,
                                                                  const float* __restrict__ lnStats, const float* __restrict__ lnGamma,
                                                                  const float* __restrict__ W1a, int ldw1, const float* __restrict__ mu,
                                                                  int ldmu, const float* __restrict__ W4, int ldw4,
                                                                  const float* __restrict__ P, int ldp_, float* __restrict__ dpremu,
                                                                  int ldd, float* __restrict__ dp, int lddp, int rows, int H, int a) {
    const int row0 = blockIdx.x * 16;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int li = lane & 15, kk = lane >> 4;
    const int row = min(row0 + li, rows - 1);
    const bool live = row0 + li < rows;
    __shared__ __attribute__((aligned(16))) float red[8 * 64 * 4];
    __shared__ float red2[2][8 * 16];
    const int nb = wave * 16 * TPW;

    // ---- everything of the rows, and the A fragments of both products, in flight at once
    float4 g[TPW], h[TPW], pv[TPW];
    float wa[TPW][4], wc[TPW][4];
    const float mean = lnStats[2 * row], rstd = lnStats[2 * row + 1];
    float mu_[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) mu_[m] = mu[(size_t)row * ldmu + min(4 * kk + m, a - 1)];
    {
        float4 dy[TPW], yv[TPW], xv[TPW], gam[TPW];
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int n0 = nb + 16 * t + 4 * kk;
            dy[t] = ldg4(dt1 + (size_t)row * ldt + n0);
            yv[t] = ldg4(lnY + (size_t)row * ldy + n0);
            xv[t] = ldg4(lnX + (size_t)row * ldx + n0);
            gam[t] = ldg4(lnGamma + n0);
            pv[t] = ldg4(P + (size_t)row * ldp_ + n0);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                // d action[j] = sum_n g[n] W1[n][aoff + j]:  A[M = j = li][k] with n = n0' + 4 kk + m
                wa[t][m] = W1a[(size_t)(nb + 16 * t + 4 * kk + m) * ldw1 + min(li, a - 1)];
                // d p[n] = sum_j d premu[j] W4[j][n]:        A[M = n = n0' + li][k] with j = 4 kk + m
                wc[t][m] = W4[(size_t)min(4 * kk + m, a - 1) * ldw4 + nb + 16 * t + li];
            }
        }
        // du = dy (1 - y^2); g = du gamma; xhat = (x - mean) rstd      (ln_tanh_bwd_kernel, same math)
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            g[t] = make_float4(dy[t].x * (1.f - yv[t].x * yv[t].x) * gam[t].x, dy[t].y * (1.f - yv[t].y * yv[t].y) * gam[t].y,
                               dy[t].z * (1.f - yv[t].z * yv[t].z) * gam[t].z, dy[t].w * (1.f - yv[t].w * yv[t].w) * gam[t].w);
            h[t] = make_float4((xv[t].x - mean) * rstd, (xv[t].y - mean) * rstd, (xv[t].z - mean) * rstd, (xv[t].w - mean) * rstd);
        }
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        s1 += (g[t].x + g[t].y) + (g[t].z + g[t].w);
        s2 += (g[t].x * h[t].x + g[t].y * h[t].y) + (g[t].z * h[t].z + g[t].w * h[t].w);
    }
    // (both statistics behind ONE barrier: two slots of red2)
    s1 = quad_sum(s1); s2 = quad_sum(s2);
    if (kk == 0) { red2[0][wave * 16 + li] = s1; red2[1][wave * 16 + li] = s2; }
    __syncthreads();
    float m1 = red2[0][li], m2 = red2[1][li];
#pragma unroll
    for (int w = 1; w < 8; ++w) { m1 += red2[0][w * 16 + li]; m2 += red2[1][w * 16 + li]; }
    m1 /= (float)H; m2 /= (float)H;
    // dx = rstd (g - mean(g) - xhat mean(g xhat)); d action partial over this wave's hidden units (two accumulator chains)
    floatx4 acc0 = floatx4{0.f, 0.f, 0.f, 0.f}, acc1 = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const float4 dx = make_float4(rstd * (g[t].x - m1 - h[t].x * m2), rstd * (g[t].y - m1 - h[t].y * m2),
                                      rstd * (g[t].z - m1 - h[t].z * m2), rstd * (g[t].w - m1 - h[t].w * m2));
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (t & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[t][m], comp(dx, m), acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[t][m], comp(dx, m), acc0, 0, 0, 0);
        }
    }
    const floatx4 part = floatx4{acc0[0] + acc1[0], acc0[1] + acc1[1], acc0[2] + acc1[2], acc0[3] + acc1[3]};
    const float4 dact = fold8(red, part, wave, lane);               // lane (li, kk): d action[4 kk .. 4 kk + 3] of row li
    // d premu = d action (1 - mu^2): straight-through clamp + tanh (utils.py:171-174, fb_modules.py:124)
    float dpm[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int c = 4 * kk + m;
        const float v = comp(dact, m) * (1.f - mu_[m] * mu_[m]);
        dpm[m] = c < a ? v : 0.f;
        if (wave == 0 && live && c < a) dpremu[(size_t)row * ldd + c] = v;
    }
    if (!live) return;
    // d p = (d premu . W4) relu'(p)
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        floatx4 c4 = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 4; ++m) c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[t][m], dpm[m], c4, 0, 0, 0);
        const float4 o = make_float4(pv[t].x > 0.f ? c4[0] : 0.f, pv[t].y > 0.f ? c4[1] : 0.f, pv[t].z > 0.f ? c4[2] : 0.f,
                                     pv[t].w > 0.f ? c4[3] : 0.f);
        *reinterpret_cast<float4*>(dp + (size_t)row * lddp + nb + 16 * t + 4 * kk) = o;
    }
}

bool actor_head_bwd_tiles_ok(int ldt, int ldy, int ldx, int ldp_, int lddp, int rows, int H, int a, const void* p0, const void* p1,
                             const void* p2, const void* p3, const void* p4, const void* p5) {
    auto al = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    return tiles_on(2, rows) && a >= 1 && a <= 16 && (H == 512 || H == 1024) && rows >= 1 && !((ldt | ldy | ldx | ldp_ | lddp) & 3) &&
           al(p0) && al(p1) && al(p2) && al(p3) && al(p4) && al(p5);
}

hipError_t launch_actor_head_bwd_tiles(const float* dt1, int ldt, const float* W1a, int ldw1, const float* mu, int ldmu, const float* W4,
                                       int ldw4, const float* P, int ldp_, float* dpremu, int ldd, float* dp, int lddp, int rows, int H,
                                       int a, hipStream_t s, const float* lnY, int ldy, const float* lnX, int ldx, const float* lnStats,
                                       const float* lnGamma) {
    dim3 grid((rows + 15) / 16), block(512);
#define AHT(T) hipLaunchKernelGGL(actor_head_bwd_tile_kernel<T>, grid, block, 0, s, dt1, ldt, lnY, ldy, lnX, ldx, lnStats, lnGamma, W1a, \
                                  ldw1, mu, ldmu, W4, ldw4, P, ldp_, dpremu, ldd, dp, lddp, rows, H, a)
    if (H == 512) AHT(4); else AHT(8);
#undef AHT
    return hipGetLastError();
}

}  // namespace fbhip
