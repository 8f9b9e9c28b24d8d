// Row-wise kernels of the FB-DDPG step for gfx950: one 64-lane wavefront per row, wave-shuffle reductions.
//   ln_tanh_fwd / ln_tanh_bwd : the "ntanh" non-linearity  LayerNorm(eps=1e-5, affine) + Tanh (fb_modules.py:49-50)
//   l2norm_fwd / l2norm_bwd   : sqrt(d) * F.normalize(x, dim=1)  (fb_modules.py:229, fb_ddpg.py:226,484)
//   policy_sample             : mu = tanh(.) and TruncatedNormal.sample with straight-through clamp (utils.py:171-185)
//   actor_loss                : Q = min(F1.z, F2.z), loss = -mean Q and dF_i (fb_ddpg.py:400-406)
#include "common.h"
#include "fbhip.h"

namespace fbhip {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// load 4 consecutive floats starting at element n0 of a row; elements >= n read as 0
__device__ __forceinline__ float4 ld4(const float* __restrict__ row, int n0, int n, bool vec) {
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n0 < n) {
        if (vec && n0 + 3 < n) {
            x = *reinterpret_cast<const float4*>(row + n0);
        } else {
            x.x = row[n0];
            if (n0 + 1 < n) x.y = row[n0 + 1];
            if (n0 + 2 < n) x.z = row[n0 + 2];
            if (n0 + 3 < n) x.w = row[n0 + 3];
        }
    }
    return x;
}
__device__ __forceinline__ void st4(float* __restrict__ row, int n0, int n, bool vec, float4 x) {
    if (n0 < n) {
        if (vec && n0 + 3 < n) {
            *reinterpret_cast<float4*>(row + n0) = x;
        } else {
            row[n0] = x.x;
            if (n0 + 1 < n) row[n0 + 1] = x.y;
            if (n0 + 2 < n) row[n0 + 2] = x.z;
            if (n0 + 3 < n) row[n0 + 3] = x.w;
        }
    }
}
__device__ __forceinline__ float4 mask4(float4 x, int n0, int n) {
    if (n0 >= n) x.x = 0.f;
    if (n0 + 1 >= n) x.y = 0.f;
    if (n0 + 2 >= n) x.z = 0.f;
    if (n0 + 3 >= n) x.w = 0.f;
    return x;
}
// branch-free variants for rows whose first ``np`` (multiple of 4) elements are readable: quads at or beyond np read a
// clamped (valid) address and are zeroed by mask4 -- no control flow between the loads, so they all stay in flight
__device__ __forceinline__ float4 ld4c(const float* __restrict__ row, int n0, int np) {
    return *reinterpret_cast<const float4*>(row + (n0 < np ? n0 : np - 4));
}

static inline bool aligned16(const void* p, int ld) { return (((uintptr_t)p & 15) == 0) && ((ld & 3) == 0); }

// ------------------------------------------------------------------------------------------------------
template <int MAXQ, bool FAST>
__global__ void __launch_bounds__(256) ln_tanh_fwd_kernel(const LnFwdGroup g) {
    const LnFwdProblem& p = g.p[blockIdx.y];
    const float* __restrict__ x = p.x;
    float* __restrict__ y = p.y;
    const int ldx = p.ldx, ldy = p.ldy, rows = p.rows, n = p.n, vx = p.vx, vy = p.vy, vp = p.vp, np = p.np;
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * ldx;
    float4 v[MAXQ], gm[MAXQ], bt[MAXQ];
    if constexpr (FAST) {
        // every load of the row (and its gamma / beta) is issued before the first use
#pragma unroll
        for (int i = 0; i < MAXQ; ++i) {
            const int n0 = 4 * (lane + 64 * i);
            v[i] = ld4c(xr, n0, np); gm[i] = ld4c(p.gamma, n0, np); bt[i] = ld4c(p.beta, n0, np);
        }
#pragma unroll
        for (int i = 0; i < MAXQ; ++i) v[i] = mask4(v[i], 4 * (lane + 64 * i), n);
    } else {
#pragma unroll
        for (int i = 0; i < MAXQ; ++i) v[i] = ld4(xr, 4 * (lane + 64 * i), n, vx);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXQ; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = wave_sum(s) / (float)n;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXQ; ++i) {
        const int n0 = 4 * (lane + 64 * i);
        float4 c = mask4(make_float4(v[i].x - mean, v[i].y - mean, v[i].z - mean, v[i].w - mean), n0, n);
        q += (c.x * c.x + c.y * c.y) + (c.z * c.z + c.w * c.w);
    }
    const float var = wave_sum(q) / (float)n;            // biased, like nn.LayerNorm
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    float* yr = y + (size_t)row * ldy;
#pragma unroll
    for (int i = 0; i < MAXQ; ++i) {
        const int n0 = 4 * (lane + 64 * i);
        if (n0 < n) {
            if constexpr (!FAST) { gm[i] = ld4(p.gamma, n0, n, vp); bt[i] = ld4(p.beta, n0, n, vp); }
            float4 o;
            o.x = tanhf((v[i].x - mean) * rstd * gm[i].x + bt[i].x);
            o.y = tanhf((v[i].y - mean) * rstd * gm[i].y + bt[i].y);
            o.z = tanhf((v[i].z - mean) * rstd * gm[i].z + bt[i].z);
            o.w = tanhf((v[i].w - mean) * rstd * gm[i].w + bt[i].w);
            if constexpr (FAST) {
                // the quad lies inside [0, np): elements in [n, np) are written as 0 (pad columns stay zero)
                *reinterpret_cast<float4*>(yr + n0) = mask4(o, n0, n);
            } else {
                st4(yr, n0, n, vy, o);
            }
        }
    }
    if (lane == 0) {
        p.stats[2 * row] = mean;
        p.stats[2 * row + 1] = rstd;
    }
}

hipError_t launch_ln_tanh_fwd_group(LnFwdGroup g, hipStream_t s) {
    if (g.n < 1) return hipSuccess;
    int maxrows = 0, maxn = 0;
    bool fast = true;
    for (int i = 0; i < g.n; ++i) {
        LnFwdProblem& p = g.p[i];
        if (p.n > 2048) return hipErrorInvalidValue;
        p.vx = aligned16(p.x, p.ldx); p.vy = aligned16(p.y, p.ldy);
        p.vp = aligned16(p.gamma, 4) && aligned16(p.beta, 4);
        fast = fast && p.vx && p.vy && p.vp && p.np >= p.n && (p.np & 3) == 0 && p.np >= 4 && p.np <= p.ldx && p.np <= p.ldy;
        maxrows = p.rows > maxrows ? p.rows : maxrows;
        maxn = p.n > maxn ? p.n : maxn;
    }
    if (maxrows <= 0) return hipSuccess;
    dim3 grid((maxrows + 3) / 4, g.n), block(256);
    const int q = (maxn + 255) / 256;
#define LN_FWD(Q)                                                                              \
    if (fast) hipLaunchKernelGGL((ln_tanh_fwd_kernel<Q, true>), grid, block, 0, s, g);         \
    else hipLaunchKernelGGL((ln_tanh_fwd_kernel<Q, false>), grid, block, 0, s, g)
    if (q <= 1) { LN_FWD(1); } else if (q <= 2) { LN_FWD(2); } else if (q <= 3) { LN_FWD(3); } else if (q <= 4) { LN_FWD(4); } else { LN_FWD(8); }
#undef LN_FWD
    return hipGetLastError();
}

hipError_t launch_ln_tanh_fwd(const float* x, int ldx, const float* gamma, const float* beta, float* y, int ldy,
                              float* stats, int rows, int n, hipStream_t s) {
    LnFwdGroup g{};
    g.n = 1;
    g.p[0] = LnFwdProblem{x, ldx, gamma, beta, y, ldy, stats, rows, n, 0, 0, 0, (n & 3) == 0 ? n : 0};
    return launch_ln_tanh_fwd_group(g, s);
}

// ------------------------------------------------------------------------------------------------------
// backward of y = tanh(gamma * xhat + beta):  du = dy (1 - y^2); g = du * gamma;
// dx = rstd (g - mean(g) - xhat mean(g xhat));  dgamma = sum_rows du xhat;  dbeta = sum_rows du.
// Each workgroup owns LN_BWD_ROWS_PER_BLOCK rows (2 per wave) and emits one partial row of (dgamma, dbeta);
// a second tiny kernel folds the partial rows in a fixed order (deterministic, no atomics).
template <int MAXQ, bool FAST>
__global__ void __launch_bounds__(256) ln_tanh_bwd_kernel(const LnBwdGroup grp) {
    extern __shared__ float lds[];          // [4 waves][n] (used twice: dgamma then dbeta)
    const LnBwdProblem& p = grp.p[blockIdx.y];
    const int rows = p.rows, n = p.n;
    if (blockIdx.x * LN_BWD_ROWS_PER_BLOCK >= rows) return;
    const float* __restrict__ gamma = p.gamma;
    float* __restrict__ partials = p.partials;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float4 pg[MAXQ], pb[MAXQ];
#pragma unroll
    for (int i = 0; i < MAXQ; ++i) pg[i] = pb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float inv_n = 1.0f / (float)n;
    constexpr int RPW = LN_BWD_ROWS_PER_BLOCK / 4;          // rows per wave
    if constexpr (FAST) {
        // all loads of BOTH rows of this wave (3 streams + gamma) are issued before the first use; dx may alias dy
        // (in place), so the compiler cannot hoist the second row's loads over the first row's stores by itself
        const int np = p.np;
        float4 d[RPW][MAXQ], yy[RPW][MAXQ], xx[RPW][MAXQ], gm[MAXQ];
        float mean[RPW], rstd[RPW];
        int rowi[RPW];
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int row = blockIdx.x * LN_BWD_ROWS_PER_BLOCK + rr * 4 + wid;
            rowi[rr] = row;
            const int rc = row < rows ? row : rows - 1;     // clamped: results of a row past the end are dropped
            mean[rr] = p.stats[2 * rc]; rstd[rr] = p.stats[2 * rc + 1];
#pragma unroll
            for (int i = 0; i < MAXQ; ++i) {
                const int n0 = 4 * (lane + 64 * i);
                d[rr][i] = ld4c(p.dy + (size_t)rc * p.lddy, n0, np);
                yy[rr][i] = ld4c(p.y + (size_t)rc * p.ldy, n0, np);
                xx[rr][i] = ld4c(p.x + (size_t)rc * p.ldx, n0, np);
            }
        }
#pragma unroll
        for (int i = 0; i < MAXQ; ++i) gm[i] = ld4c(gamma, 4 * (lane + 64 * i), np);
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const bool live = rowi[rr] < rows;
            float4 g[MAXQ], xh[MAXQ];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < MAXQ; ++i) {
                const int n0 = 4 * (lane + 64 * i);
                const float4 dd = mask4(d[rr][i], n0, live ? n : 0), y4 = yy[rr][i], x4 = xx[rr][i];
                float4 du, h;
                du.x = dd.x * (1.f - y4.x * y4.x); du.y = dd.y * (1.f - y4.y * y4.y);
                du.z = dd.z * (1.f - y4.z * y4.z); du.w = dd.w * (1.f - y4.w * y4.w);
                h = mask4(make_float4((x4.x - mean[rr]) * rstd[rr], (x4.y - mean[rr]) * rstd[rr], (x4.z - mean[rr]) * rstd[rr],
                                      (x4.w - mean[rr]) * rstd[rr]), n0, n);
                xh[i] = h;
                g[i] = make_float4(du.x * gm[i].x, du.y * gm[i].y, du.z * gm[i].z, du.w * gm[i].w);
                s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
                s2 += (g[i].x * h.x + g[i].y * h.y) + (g[i].z * h.z + g[i].w * h.w);
                pg[i].x += du.x * h.x; pg[i].y += du.y * h.y; pg[i].z += du.z * h.z; pg[i].w += du.w * h.w;
                pb[i].x += du.x; pb[i].y += du.y; pb[i].z += du.z; pb[i].w += du.w;
            }
            const float m1 = wave_sum(s1) * inv_n, m2 = wave_sum(s2) * inv_n;
            if (live) {
                float* dxr = p.dx + (size_t)rowi[rr] * p.lddx;
#pragma unroll
                for (int i = 0; i < MAXQ; ++i) {
                    const int n0 = 4 * (lane + 64 * i);
                    if (n0 < n) {
                        float4 o;
                        o.x = rstd[rr] * (g[i].x - m1 - xh[i].x * m2);
                        o.y = rstd[rr] * (g[i].y - m1 - xh[i].y * m2);
                        o.z = rstd[rr] * (g[i].z - m1 - xh[i].z * m2);
                        o.w = rstd[rr] * (g[i].w - m1 - xh[i].w * m2);
                        *reinterpret_cast<float4*>(dxr + n0) = mask4(o, n0, n);     // pad columns [n, np) stay zero
                    }
                }
            }
        }
    } else {
#pragma unroll 1
        for (int rr = 0; rr < RPW; ++rr) {
            const int row = blockIdx.x * LN_BWD_ROWS_PER_BLOCK + rr * 4 + wid;
            if (row >= rows) break;
            const float mean = p.stats[2 * row], rstd = p.stats[2 * row + 1];
            const float* dyr = p.dy + (size_t)row * p.lddy;
            const float* yr = p.y + (size_t)row * p.ldy;
            const float* xr = p.x + (size_t)row * p.ldx;
            float4 g[MAXQ], xh[MAXQ];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < MAXQ; ++i) {
                const int n0 = 4 * (lane + 64 * i);
                const float4 d = ld4(dyr, n0, n, p.vdy), yy = ld4(yr, n0, n, p.vy), xx = ld4(xr, n0, n, p.vx);
                const float4 gm = ld4(gamma, n0, n, p.vp);
                float4 du, h;
                du.x = d.x * (1.f - yy.x * yy.x); du.y = d.y * (1.f - yy.y * yy.y);
                du.z = d.z * (1.f - yy.z * yy.z); du.w = d.w * (1.f - yy.w * yy.w);
                h = mask4(make_float4((xx.x - mean) * rstd, (xx.y - mean) * rstd, (xx.z - mean) * rstd,
                                      (xx.w - mean) * rstd), n0, n);
                xh[i] = h;
                g[i] = make_float4(du.x * gm.x, du.y * gm.y, du.z * gm.z, du.w * gm.w);
                s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
                s2 += (g[i].x * h.x + g[i].y * h.y) + (g[i].z * h.z + g[i].w * h.w);
                pg[i].x += du.x * h.x; pg[i].y += du.y * h.y; pg[i].z += du.z * h.z; pg[i].w += du.w * h.w;
                pb[i].x += du.x; pb[i].y += du.y; pb[i].z += du.z; pb[i].w += du.w;
            }
            const float m1 = wave_sum(s1) * inv_n, m2 = wave_sum(s2) * inv_n;
            float* dxr = p.dx + (size_t)row * p.lddx;
#pragma unroll
            for (int i = 0; i < MAXQ; ++i) {
                const int n0 = 4 * (lane + 64 * i);
                float4 o;
                o.x = rstd * (g[i].x - m1 - xh[i].x * m2);
                o.y = rstd * (g[i].y - m1 - xh[i].y * m2);
                o.z = rstd * (g[i].z - m1 - xh[i].z * m2);
                o.w = rstd * (g[i].w - m1 - xh[i].w * m2);
                st4(dxr, n0, n, p.vdx, o);
            }
        }
    }
    if (partials == nullptr) return;
    float* out = partials + (size_t)blockIdx.x * 2 * n;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MAXQ; ++i) {
            const int n0 = 4 * (lane + 64 * i);
            const float4 v = pass == 0 ? pg[i] : pb[i];
            if (n0 < n) lds[wid * n + n0] = v.x;
            if (n0 + 1 < n) lds[wid * n + n0 + 1] = v.y;
            if (n0 + 2 < n) lds[wid * n + n0 + 2] = v.z;
            if (n0 + 3 < n) lds[wid * n + n0 + 3] = v.w;
        }
        __syncthreads();
        for (int j = threadIdx.x; j < n; j += 256)
            out[pass * n + j] = (lds[j] + lds[n + j]) + (lds[2 * n + j] + lds[3 * n + j]);
    }
}

// out[j] = sum_b partials[b][j], j < 2n.  Workgroup = 8 row-groups x 32 columns: every thread folds nb/8 partial
// rows (coalesced 128-byte reads), the 8 group sums are combined in a fixed order through LDS.
__global__ void __launch_bounds__(256) ln_colreduce_kernel(const LnBwdGroup grp) {
    __shared__ float red[8][32];
    const LnBwdProblem& p = grp.p[blockIdx.y];
    if (p.partials == nullptr) return;
    const int n = p.n, nb = (p.rows + LN_BWD_ROWS_PER_BLOCK - 1) / LN_BWD_ROWS_PER_BLOCK;
    const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int j = blockIdx.x * 32 + c;
    if (blockIdx.x * 32 >= 2 * n) return;
    float s = 0.f;
    if (j < 2 * n)
        for (int b0 = rg; b0 < nb; b0 += 64) {              // eight partial rows in flight at once (same summation order)
            float x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = p.partials[(size_t)min(b0 + 8 * u, nb - 1) * 2 * n + j];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (b0 + 8 * u < nb) ? x[u] : 0.f;
        }
    red[rg][c] = s;
    __syncthreads();
    if (rg == 0 && j < 2 * n) {
        float t = red[0][c];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += red[k][c];
        if (j < n) p.dgamma[j] = t; else p.dbeta[j - n] = t;
    }
}

hipError_t launch_ln_tanh_bwd_group(LnBwdGroup g, hipStream_t s, ColReduceJobs* defer) {
    if (g.n < 1) return hipSuccess;
    int maxrows = 0, maxn = 0;
    bool any_param = false, fast = true;
    for (int i = 0; i < g.n; ++i) {
        LnBwdProblem& p = g.p[i];
        if (p.n > 2048) return hipErrorInvalidValue;
        const bool want = p.dgamma != nullptr && p.dbeta != nullptr;
        if (want && p.partials == nullptr) return hipErrorInvalidValue;
        if (!want) p.partials = nullptr;
        any_param |= want;
        p.vdy = aligned16(p.dy, p.lddy); p.vy = aligned16(p.y, p.ldy); p.vx = aligned16(p.x, p.ldx);
        p.vdx = aligned16(p.dx, p.lddx); p.vp = aligned16(p.gamma, 4);
        fast = fast && p.vdy && p.vy && p.vx && p.vdx && p.vp && p.np >= p.n && (p.np & 3) == 0 && p.np >= 4 &&
               p.np <= p.lddy && p.np <= p.ldy && p.np <= p.ldx && p.np <= p.lddx && p.rows >= 1;
        maxrows = p.rows > maxrows ? p.rows : maxrows;
        maxn = p.n > maxn ? p.n : maxn;
    }
    if (maxrows <= 0) return hipSuccess;
    const int nb = (maxrows + LN_BWD_ROWS_PER_BLOCK - 1) / LN_BWD_ROWS_PER_BLOCK;
    dim3 grid(nb, g.n), block(256);
    const size_t shmem = (size_t)4 * maxn * sizeof(float);
    const int q = (maxn + 255) / 256;
    if (q > 4) fast = false;                       // the two-rows-in-flight variant is built for n <= 1024
#define LN_BWD(Q)                                                                                  \
    if (fast) hipLaunchKernelGGL((ln_tanh_bwd_kernel<Q, true>), grid, block, shmem, s, g);         \
    else hipLaunchKernelGGL((ln_tanh_bwd_kernel<Q, false>), grid, block, shmem, s, g)
    if (q <= 1) { LN_BWD(1); } else if (q <= 2) { LN_BWD(2); } else if (q <= 3) { LN_BWD(3); } else if (q <= 4) { LN_BWD(4); }
    else hipLaunchKernelGGL((ln_tanh_bwd_kernel<8, false>), grid, block, shmem, s, g);
#undef LN_BWD
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || !any_param) return e;
    if (defer != nullptr) {                        // hand the column reduces to the next split-K reduce launch
        int room = CR_MAX - defer->count;
        int want = 0;
        for (int i = 0; i < g.n; ++i) want += g.p[i].partials != nullptr ? 1 : 0;
        if (want <= room) {
            for (int i = 0; i < g.n; ++i) {
                const LnBwdProblem& p = g.p[i];
                if (p.partials == nullptr) continue;
                const int k = defer->count++;
                defer->partials[k] = p.partials; defer->dgamma[k] = p.dgamma; defer->dbeta[k] = p.dbeta;
                defer->rows[k] = p.rows; defer->n[k] = p.n;
                if (k == 0) defer->block_start[0] = 0;
                defer->block_start[k + 1] = defer->block_start[k] + (2 * p.n + 31) / 32;
            }
            return hipSuccess;
        }
    }
    hipLaunchKernelGGL(ln_colreduce_kernel, dim3((2 * maxn + 31) / 32, g.n), dim3(256), 0, s, g);
    return hipGetLastError();
}

hipError_t launch_ln_tanh_bwd(const float* dy, int lddy, const float* y, int ldy, const float* x, int ldx,
                              const float* stats, const float* gamma, float* dx, int lddx, float* dgamma,
                              float* dbeta, float* partials, int rows, int n, hipStream_t s) {
    LnBwdGroup g{};
    g.n = 1;
    g.p[0] = LnBwdProblem{dy, lddy, y, ldy, x, ldx, stats, gamma, dx, lddx, dgamma, dbeta, partials, rows, n, 0, 0, 0, 0, 0,
                          (n & 3) == 0 ? n : 0};
    return launch_ln_tanh_bwd_group(g, s);
}

// ------------------------------------------------------------------------------------------------------
constexpr int L2_MAXE = 4;      // d <= 256
__global__ void __launch_bounds__(256) l2norm_fwd_kernel(const L2Group grp) {
    const L2Problem& p = grp.p[blockIdx.y];
    const int rows = p.rows, d = p.d;
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* yr = p.y + (size_t)row * p.ldy;
    float v[L2_MAXE];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < L2_MAXE; ++i) {
        const int j = lane + 64 * i;
        v[i] = j < d ? yr[j] : 0.f;
        s += v[i] * v[i];
    }
    const float nrm = sqrtf(wave_sum(s));
    const float den = fmaxf(nrm, 1e-12f);                 // F.normalize eps
    float* o = p.out + (size_t)row * p.ldo;
#pragma unroll
    for (int i = 0; i < L2_MAXE; ++i) {
        const int j = lane + 64 * i;
        if (j < d) o[j] = p.scale * (v[i] / den);
    }
    if (p.norms != nullptr && lane == 0) p.norms[row] = nrm;
}

hipError_t launch_l2norm_fwd_group(const L2Group& g, hipStream_t s) {
    int maxrows = 0;
    for (int i = 0; i < g.n; ++i) {
        if (g.p[i].d > 64 * L2_MAXE) return hipErrorInvalidValue;
        maxrows = g.p[i].rows > maxrows ? g.p[i].rows : maxrows;
    }
    if (g.n < 1 || maxrows <= 0) return hipSuccess;
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((maxrows + 3) / 4, g.n), dim3(256), 0, s, g);
    return hipGetLastError();
}

hipError_t launch_l2norm_fwd(const float* y, int ldy, float* out, int ldo, float* norms, int rows, int d,
                             float scale, hipStream_t s) {
    L2Group g{};
    g.n = 1;
    g.p[0] = L2Problem{y, ldy, out, ldo, norms, rows, d, scale};
    return launch_l2norm_fwd_group(g, s);
}

// dy = (sqrt(d)/||y||) (dB - yhat (yhat . dB)),  yhat = y/||y||
__global__ void __launch_bounds__(256) l2norm_bwd_kernel(const float* __restrict__ dB, int lddb,
                                                         const float* __restrict__ y, int ldy,
                                                         const float* __restrict__ norms, float* __restrict__ dy,
                                                         int lddy, int rows, int d, float scale) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float nrm = fmaxf(norms[row], 1e-12f);
    const float inv = 1.0f / nrm;
    float yh[L2_MAXE], g[L2_MAXE];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < L2_MAXE; ++i) {
        const int j = lane + 64 * i;
        yh[i] = j < d ? y[(size_t)row * ldy + j] * inv : 0.f;
        g[i] = j < d ? dB[(size_t)row * lddb + j] : 0.f;
        dot += yh[i] * g[i];
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int i = 0; i < L2_MAXE; ++i) {
        const int j = lane + 64 * i;
        if (j < d) dy[(size_t)row * lddy + j] = scale * inv * (g[i] - yh[i] * dot);
    }
}

hipError_t launch_l2norm_bwd(const float* dB, int lddb, const float* y, int ldy, const float* norms, float* dy,
                             int lddy, int rows, int d, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    if (d > 64 * L2_MAXE) return hipErrorInvalidValue;
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, dB, lddb, y, ldy, norms, dy, lddy,
                       rows, d, sqrtf((float)d));
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) policy_sample_kernel(const float* __restrict__ pre, int ldp,
                                                            const float* __restrict__ noise, int ldn, float stddev,
                                                            float clip, float* __restrict__ mu, int ldmu,
                                                            float* __restrict__ action, int lda, int rows, int a,
                                                            const Squash sq) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * a) return;
    const int r = idx / a, c = idx % a;
    const float loc = pre[(size_t)r * ldp + c];
    const float m = tanhf(loc);                    // Actor: mu; DiagGaussianActor: dist.mean = tanh(loc)
    if (mu != nullptr) mu[(size_t)r * ldmu + c] = m;
    float act = m;
    if (sq.on) {
        if (noise != nullptr) {                    // SquashedNormal.sample / rsample: tanh(loc + scale eps), no clamp
            const float log_std = sq.lo + 0.5f * (sq.hi - sq.lo) * (tanhf(pre[(size_t)r * ldp + a + c]) + 1.f);
            act = tanhf(loc + expf(log_std) * noise[(size_t)r * ldn + c]);
        }
    } else if (noise != nullptr) {
        float e = noise[(size_t)r * ldn + c] * stddev;
        if (clip >= 0.f) e = fminf(fmaxf(e, -clip), clip);
        const float lo = (float)(-1.0 + 1e-6), hi = (float)(1.0 - 1e-6);
        act = fminf(fmaxf(m + e, lo), hi);
    }
    if (action != nullptr) action[(size_t)r * lda + c] = act;
}

hipError_t launch_policy_sample(const float* pre, int ldp, const float* noise, int ldn, float stddev, float clip,
                                float* mu, int ldmu, float* action, int lda, int rows, int a, Squash sq, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(policy_sample_kernel, dim3((rows * a + 255) / 256), dim3(256), 0, s, pre, ldp, noise, ldn,
                       stddev, clip, mu, ldmu, action, lda, rows, a, sq);
    return hipGetLastError();
}

// torch.nn.functional.softplus (beta 1, threshold 20)
__device__ __forceinline__ float softplus20(float x) { return x > 20.f ? x : log1pf(expf(x)); }

// Backward of  L = mean_rows( temp * sum_j log_prob_j - Q )  at the DiagGaussianActor head, given dQ-part d action
// (already scaled by -1/B through dF): with u = loc + std eps (eps constant), action = tanh(u),
//   log_prob_j = Normal(loc, std).log_prob(u) - 2 (log 2 - u - softplus(-2u))       (utils.py:212-215)
//   d log_prob / du = 2 tanh(u);  the Normal term is -eps^2/2 - log std: no loc gradient, -1/std wrt std
// =>  du = dact (1 - tanh(u)^2) + (temp/B) 2 tanh(u);  dloc = du;  dlog_std = du eps std - temp/B;
//     draw = dlog_std (hi - lo)/2 (1 - tanh(raw)^2)
__global__ void __launch_bounds__(256) squash_head_bwd_kernel(const float* __restrict__ dact, int ldd,
                                                              const float* __restrict__ pre, int ldp,
                                                              const float* __restrict__ noise, int ldn,
                                                              float* __restrict__ dpre, int ldo, int rows, int a,
                                                              const Squash sq) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * a) return;
    const int r = idx / a, c = idx % a;
    const float loc = pre[(size_t)r * ldp + c], t = tanhf(pre[(size_t)r * ldp + a + c]), e = noise[(size_t)r * ldn + c];
    const float half = 0.5f * (sq.hi - sq.lo), std = expf(sq.lo + half * (t + 1.f));
    const float th = tanhf(loc + std * e), tb = sq.temp / (float)rows;
    const float du = dact[(size_t)r * ldd + c] * (1.f - th * th) + tb * 2.f * th;
    dpre[(size_t)r * ldo + c] = du;
    dpre[(size_t)r * ldo + a + c] = (du * e * std - tb) * half * (1.f - t * t);
}

hipError_t launch_squash_head_bwd(const float* dact, int ldd, const float* pre, int ldp, const float* noise, int ldn,
                                  float* dpre, int ldo, int rows, int a, Squash sq, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    if (!sq.on || noise == nullptr) return hipErrorInvalidValue;
    hipLaunchKernelGGL(squash_head_bwd_kernel, dim3((rows * a + 255) / 256), dim3(256), 0, s, dact, ldd, pre, ldp, noise,
                       ldn, dpre, ldo, rows, a, sq);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------
// One wavefront per row (4 rows per workgroup); per-workgroup partial sums of (min Q, log-prob) go to ``part`` and
// a one-wave finalize kernel folds them in a fixed order (deterministic).  metrics: ACTOR_LOSS, Q, ACTOR_LOGPROB.
__global__ void __launch_bounds__(256) actor_loss_kernel(const float* __restrict__ F1, const float* __restrict__ F2,
                                                         int ldf, const float* __restrict__ z, int ldz,
                                                         const float* __restrict__ mu, int ldmu,
                                                         const float* __restrict__ act, int lda, float stddev,
                                                         float* __restrict__ dF1, float* __restrict__ dF2,
                                                         float* __restrict__ part, int rows, int d, int a,
                                                         const Squash sq, const float* __restrict__ pre, int ldp,
                                                         const float* __restrict__ noise, int ldn) {
    __shared__ float red[4][3];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wid;
    const float inv_b = 1.0f / (float)rows;
    float qmin = 0.f, lp = 0.f, q1win = 0.f;
    if (row < rows) {
        float zz[L2_MAXE], q1 = 0.f, q2 = 0.f;
#pragma unroll
        for (int i = 0; i < L2_MAXE; ++i) {
            const int j = lane + 64 * i;
            zz[i] = j < d ? z[(size_t)row * ldz + j] : 0.f;
            if (j < d) {
                q1 += F1[(size_t)row * ldf + j] * zz[i];
                q2 += F2[(size_t)row * ldf + j] * zz[i];
            }
        }
        q1 = wave_sum(q1);
        q2 = wave_sum(q2);
        // torch.min(Q1, Q2) backward: all of the gradient to the strict arg-min, split evenly on exact ties
        const float w1 = q1 < q2 ? 1.f : (q1 == q2 ? 0.5f : 0.f), w2 = 1.f - w1;
#pragma unroll
        for (int i = 0; i < L2_MAXE; ++i) {
            const int j = lane + 64 * i;
            if (j < d) {
                dF1[(size_t)row * ldf + j] = -zz[i] * inv_b * w1;
                dF2[(size_t)row * ldf + j] = -zz[i] * inv_b * w2;
            }
        }
        if (lane < a && sq.on) {                   // SquashedNormal.log_prob(action) with the cached pre-image (utils.py:212-215)
            const float loc = pre[(size_t)row * ldp + lane], e = noise[(size_t)row * ldn + lane];
            const float log_std = sq.lo + 0.5f * (sq.hi - sq.lo) * (tanhf(pre[(size_t)row * ldp + a + lane]) + 1.f);
            const float u = loc + expf(log_std) * e;
            lp = -0.5f * e * e - log_std - 0.91893853320467274178f - 2.f * (0.69314718055994530942f - u - softplus20(-2.f * u));
        } else if (lane < a) {
            const float df = act[(size_t)row * lda + lane] - mu[(size_t)row * ldmu + lane];
            lp = -(df * df) / (2.f * stddev * stddev) - logf(stddev) - 0.91893853320467274178f;
        }
        lp = wave_sum(lp);
        qmin = fminf(q1, q2);
        q1win = q1 > q2 ? 1.f : 0.f;               // additional_metric: q1_success = (Q1 > Q2).mean()  (fb_ddpg.py:403-404, 417)
    }
    if (lane == 0) { red[wid][0] = qmin; red[wid][1] = lp; red[wid][2] = q1win; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[3 * blockIdx.x] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
        part[3 * blockIdx.x + 1] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
        part[3 * blockIdx.x + 2] = (red[0][2] + red[1][2]) + (red[2][2] + red[3][2]);
    }
}

// The same loss WITHOUT materialising F1, F2 (the update's actor phase).  With V_i = z . W4_i  ([B,d] x [d,H], a GEMM that
// needs neither the ForwardMap pass nor the action, so it rides in the phase's first round):
//   Q_i[s] = F_i[s] . z[s] = (p_i[s] W4_i^T + b4_i) . z[s] = p_i[s] . V_i[s] + b4_i . z[s]
//   d p_i[s] = (dF_i[s] W4_i) * relu'(p_i) = -(w_i[s] / B) V_i[s] * (p_i[s] > 0)          (dF_i = -z w_i / B)
// i.e. the heads' output GEMM + its split-K reduce, the loss kernel and the heads' data-gradient GEMM collapse into this
// one row kernel: one wavefront per row reads p [B, 2H] and V [B, 2H] and overwrites V with d p.
// QM: float4 quads per lane and half row, H <= 256 QM: the rows of p and V are read ONCE, every load in flight before the first
// use (as loops over k: 2 x H / 256 dependent round trips)
template <int QM>
__global__ void __launch_bounds__(256) actor_q_kernel(const float* __restrict__ P, int ldp_, float* __restrict__ V, int ldv,
                                                      const float* __restrict__ z, int ldz,
                                                      const float* __restrict__ b41, const float* __restrict__ b42,
                                                      const float* __restrict__ mu, int ldmu, const float* __restrict__ act,
                                                      int lda, float stddev, float* __restrict__ part, int rows, int H, int d,
                                                      int a, const Squash sq, const float* __restrict__ pre, int ldp,
                                                      const float* __restrict__ noise, int ldn, StepState* adv, int adv_which) {
    __shared__ float red[4][3];
    if (adv != nullptr && blockIdx.x == 0 && threadIdx.x == 255) step_advance_device(adv, adv_which);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wid;
    const float inv_b = 1.0f / (float)rows;
    float qmin = 0.f, lp = 0.f, q1win = 0.f;
    if (row < rows) {
        const float4* p1 = reinterpret_cast<const float4*>(P + (size_t)row * ldp_);
        const float4* p2 = reinterpret_cast<const float4*>(P + (size_t)row * ldp_ + H);
        float4* v1 = reinterpret_cast<float4*>(V + (size_t)row * ldv);
        float4* v2 = reinterpret_cast<float4*>(V + (size_t)row * ldv + H);
        float q1 = 0.f, q2 = 0.f;
        const int nq = H / 4;
        float4 A1[QM], C1[QM], A2[QM], C2[QM];
#pragma unroll
        for (int i = 0; i < QM; ++i) {
            const int k = min(lane + 64 * i, nq - 1);
            A1[i] = p1[k]; C1[i] = v1[k]; A2[i] = p2[k]; C2[i] = v2[k];
        }
#pragma unroll
        for (int i = 0; i < QM; ++i) {
            if (lane + 64 * i < nq) {
                q1 += A1[i].x * C1[i].x + A1[i].y * C1[i].y + A1[i].z * C1[i].z + A1[i].w * C1[i].w;
                q2 += A2[i].x * C2[i].x + A2[i].y * C2[i].y + A2[i].z * C2[i].z + A2[i].w * C2[i].w;
            }
        }
        for (int j = lane; j < d; j += 64) {
            const float zz = z[(size_t)row * ldz + j];
            q1 += b41[j] * zz;
            q2 += b42[j] * zz;
        }
        q1 = wave_sum(q1);
        q2 = wave_sum(q2);
        // torch.min(Q1, Q2) backward: all of the gradient to the strict arg-min, split evenly on exact ties
        const float w1 = q1 < q2 ? 1.f : (q1 == q2 ? 0.5f : 0.f), w2 = 1.f - w1;
        const float s1 = -inv_b * w1, s2 = -inv_b * w2;
#pragma unroll
        for (int i = 0; i < QM; ++i) {
            const int k = lane + 64 * i;
            if (k < nq) {
                const float4 a1 = A1[i], a2 = A2[i];
                float4 c1 = C1[i], c2 = C2[i];
                c1.x = a1.x > 0.f ? s1 * c1.x : 0.f; c1.y = a1.y > 0.f ? s1 * c1.y : 0.f;
                c1.z = a1.z > 0.f ? s1 * c1.z : 0.f; c1.w = a1.w > 0.f ? s1 * c1.w : 0.f;
                c2.x = a2.x > 0.f ? s2 * c2.x : 0.f; c2.y = a2.y > 0.f ? s2 * c2.y : 0.f;
                c2.z = a2.z > 0.f ? s2 * c2.z : 0.f; c2.w = a2.w > 0.f ? s2 * c2.w : 0.f;
                v1[k] = c1; v2[k] = c2;
            }
        }
        if (lane < a && sq.on) {                   // SquashedNormal.log_prob(action) with the cached pre-image (utils.py:212-215)
            const float loc = pre[(size_t)row * ldp + lane], e = noise[(size_t)row * ldn + lane];
            const float log_std = sq.lo + 0.5f * (sq.hi - sq.lo) * (tanhf(pre[(size_t)row * ldp + a + lane]) + 1.f);
            const float u = loc + expf(log_std) * e;
            lp = -0.5f * e * e - log_std - 0.91893853320467274178f - 2.f * (0.69314718055994530942f - u - softplus20(-2.f * u));
        } else if (lane < a) {
            const float df = act[(size_t)row * lda + lane] - mu[(size_t)row * ldmu + lane];
            lp = -(df * df) / (2.f * stddev * stddev) - logf(stddev) - 0.91893853320467274178f;
        }
        lp = wave_sum(lp);
        qmin = fminf(q1, q2);
        q1win = q1 > q2 ? 1.f : 0.f;
    }
    if (lane == 0) { red[wid][0] = qmin; red[wid][1] = lp; red[wid][2] = q1win; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[3 * blockIdx.x] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
        part[3 * blockIdx.x + 1] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
        part[3 * blockIdx.x + 2] = (red[0][2] + red[1][2]) + (red[2][2] + red[3][2]);
    }
}

// ``host`` != nullptr: these are the LAST metrics of the update -- the kernel also publishes all FBHIP_NUM_METRICS to the host (what
// metrics_publish_kernel does, see there; one launch less on the metrics-on critical path)
__global__ void __launch_bounds__(64) actor_loss_finalize_kernel(const float* __restrict__ part, int nblk, int rows,
                                                                 float* __restrict__ metrics, int m_loss, int m_q,
                                                                 int m_lp, float temp /* 0: loss = -mean Q */,
                                                                 float* host, unsigned int* dseq) {
    const int lane = threadIdx.x;
    // (the other metrics were written by earlier launches: requested before the fold below)
    float mine = (host != nullptr && lane < FBHIP_NUM_METRICS) ? metrics[lane] : 0.f;
    double q = 0.0, l = 0.0, w = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 64) { q += (double)part[3 * b]; l += (double)part[3 * b + 1]; w += (double)part[3 * b + 2]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { q += __shfl_xor(q, o); l += __shfl_xor(l, o); w += __shfl_xor(w, o); }
    const float v_w = (float)(w / rows), v_loss = (float)((temp * l - q) / rows) /* fb_ddpg.py:406 */, v_q = (float)(q / rows),
                v_lp = (float)(l / rows);
    if (threadIdx.x == 0) {
        metrics[FBHIP_M_Q1_SUCCESS] = v_w;
        metrics[m_loss] = v_loss;
        metrics[m_q] = v_q;
        metrics[m_lp] = v_lp;
    }
    if (host == nullptr) return;
    mine = lane == FBHIP_M_Q1_SUCCESS ? v_w : lane == m_loss ? v_loss : lane == m_q ? v_q : lane == m_lp ? v_lp : mine;
    if (lane < FBHIP_NUM_METRICS) host[lane] = mine;
    __threadfence_system();
    if (lane == 0) {
        const unsigned int sq = *dseq + 1u;
        *dseq = sq;
        __hip_atomic_store(reinterpret_cast<unsigned int*>(host + FBHIP_NUM_METRICS), sq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The step's metrics to the HOST, from inside the step: the 32 device floats are written into pinned host memory, then a sequence
// number behind a system-scope fence.  Enqueued right after the LAST metric of an update is final (actor_loss_finalize_kernel; the
// FB metrics for an agent without an actor) -- i.e. before the actor's backward pass and optimiser step: the host that called
// update() with use_tb / use_hiplog (README.md:50; fb_ddpg.py:356-377, 413-418) spins on the number (fbhip_wait_metrics), returns
// the dict and enqueues the NEXT update while this one's tail still runs.  No D2H copy command, no stream synchronise.
__global__ void __launch_bounds__(64) metrics_publish_kernel(const float* __restrict__ metrics, float* host, unsigned int* dseq) {
    const int lane = threadIdx.x;
    if (lane < FBHIP_NUM_METRICS) host[lane] = metrics[lane];
    __threadfence_system();
    if (lane == 0) {
        const unsigned int sq = *dseq + 1u;
        *dseq = sq;
        __hip_atomic_store(reinterpret_cast<unsigned int*>(host + FBHIP_NUM_METRICS), sq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

hipError_t launch_metrics_publish(const float* metrics, float* host, unsigned int* dseq, hipStream_t s) {
    if (metrics == nullptr || host == nullptr || dseq == nullptr) return hipErrorInvalidValue;
    hipLaunchKernelGGL(metrics_publish_kernel, dim3(1), dim3(64), 0, s, metrics, host, dseq);
    return hipGetLastError();
}

hipError_t launch_actor_loss(const float* F1, const float* F2, int ldf, const float* z, int ldz, const float* mu,
                             int ldmu, const float* action, int lda, float stddev, float* dF1, float* dF2,
                             float* metrics, float* scratch, int rows, int d, int a, hipStream_t s, Squash sq,
                             const float* pre, int ldp, const float* noise, int ldn) {
    if (d > 64 * L2_MAXE || a > 64 || scratch == nullptr) return hipErrorInvalidValue;
    if (sq.on && (pre == nullptr || noise == nullptr)) return hipErrorInvalidValue;
    const int nblk = (rows + 3) / 4;
    hipLaunchKernelGGL(actor_loss_kernel, dim3(nblk), dim3(256), 0, s, F1, F2, ldf, z, ldz, mu, ldmu, action, lda,
                       stddev, dF1, dF2, scratch, rows, d, a, sq, pre, ldp, noise, ldn);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || metrics == nullptr) return e;       // loss / Q / log-prob are metrics only
    hipLaunchKernelGGL(actor_loss_finalize_kernel, dim3(1), dim3(64), 0, s, scratch, nblk, rows, metrics, 15, 16, 17,
                       sq.on ? sq.temp : 0.f, (float*)nullptr, (unsigned int*)nullptr);
    return hipGetLastError();
}

hipError_t launch_actor_q(const float* P, int ldp_, float* V, int ldv, const float* z, int ldz, const float* b41,
                          const float* b42, const float* mu, int ldmu, const float* action, int lda, float stddev,
                          float* metrics, float* scratch, int rows, int H, int d, int a, Squash sq, const float* pre, int ldp,
                          const float* noise, int ldn, hipStream_t s, StepState* adv, int adv_which, float* pub_host,
                          unsigned int* pub_seq) {
    if ((H & 3) || (ldp_ & 3) || (ldv & 3) || a > 64 || scratch == nullptr) return hipErrorInvalidValue;
    if (sq.on && (pre == nullptr || noise == nullptr)) return hipErrorInvalidValue;
    const int nblk = (rows + 3) / 4;
    if (H > 2048 || ((uintptr_t)P & 15) || ((uintptr_t)V & 15)) return hipErrorInvalidValue;
#define AQ_LAUNCH(QM)                                                                                                     \
    hipLaunchKernelGGL(actor_q_kernel<QM>, dim3(nblk), dim3(256), 0, s, P, ldp_, V, ldv, z, ldz, b41, b42, mu, ldmu, action, lda, \
                       stddev, scratch, rows, H, d, a, sq, pre, ldp, noise, ldn, adv, adv_which)
    if (H <= 256) { AQ_LAUNCH(1); } else if (H <= 512) { AQ_LAUNCH(2); } else if (H <= 1024) { AQ_LAUNCH(4); } else { AQ_LAUNCH(8); }
#undef AQ_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || metrics == nullptr) return e;       // loss / Q / log-prob are metrics only
    hipLaunchKernelGGL(actor_loss_finalize_kernel, dim3(1), dim3(64), 0, s, scratch, nblk, rows, metrics, 15, 16, 17,
                       sq.on ? sq.temp : 0.f, pub_seq != nullptr ? pub_host : nullptr, pub_seq);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------
// The policy head as one row kernel: premu = p . W4^T + b4 (NA = a or 2a outputs over K = H), then exactly policy_sample_kernel.
// Replaces an a-wide GEMM (+ its share of a split-K reduce) and the sample launch; W4 ([NA, H], 24 KB at walker dims) sits in
// LDS.  NA is a template parameter and every load is unconditional (see actor_head_bwd_kernel).
// Up to PH_MAX_JOBS row sets share the launch (blockIdx.y): the target chain's actor(next_obs) and update_actor's actor(obs)
// reach their heads in the same round.
// MAXQ > 0: jobs may carry the first layer of the trunk that consumes the action (PolicyHeadJob::base; H = 256 MAXQ): policy head,
// sample, rank-a update of the pre-activation, LayerNorm and tanh for one row per wave -- three launches of the critical path
// (policy head, K = 32 GEMM, LayerNorm) in one.
template <int NA, int MAXQ>
__global__ void __launch_bounds__(256) policy_head_kernel(const PolicyHeadJobs jobs, const float* __restrict__ W4, int ldw4,
                                                          const float* __restrict__ b4, int ldpre, int ldn, float stddev,
                                                          float clip, int ldmu, int rows, int H, int a, const Squash sq) {
    const PolicyHeadJob& jb = jobs.j[blockIdx.y];
    const float* __restrict__ P = jb.P;
    const int ldp_ = jb.ldp;
    float* __restrict__ premu = jb.premu;
    const float* __restrict__ noise = jb.noise;
    float* __restrict__ mu = jb.mu;
    float* __restrict__ action = jb.action;
    const int lda = jb.lda;
    extern __shared__ float ph_lds[];              // [NA][H]; later [a][H]: the action columns of the first layer's weight, transposed
    const bool first = MAXQ > 0 && jb.base != nullptr;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int row = min(blockIdx.x * 4 + wid, rows - 1);            // surplus waves redo the last row (same stores)
    float4 pre[MAXQ > 0 ? MAXQ : 1], gm[MAXQ > 0 ? MAXQ : 1], bt[MAXQ > 0 ? MAXQ : 1];
    float xrow[MAXQ > 0 ? 4 * MAXQ : 1];                              // MAXQ > 0: the row of P, requested before the LDS fill
    if constexpr (MAXQ > 0) {
#pragma unroll
        for (int u = 0; u < 4 * MAXQ; ++u) xrow[u] = P[(size_t)row * ldp_ + min(lane + 64 * u, H - 1)];
    }
    float w1r[MAXQ > 0 ? MAXQ : 1][NA];                              // W1[n][aoff + j], n = threadIdx.x + 256 i: parked in registers
    if constexpr (MAXQ > 0) {
        if (first) {
            // the action columns of the first layer's weight go through the SAME LDS bytes as W4, after the head's dot products
            // (two more barriers, half the LDS: at quadruped dims 48 KB instead of 96): loaded now, written to LDS later
#pragma unroll
            for (int i = 0; i < MAXQ; ++i)
#pragma unroll
                for (int jj = 0; jj < NA; ++jj) w1r[i][jj] = jb.W1a[(size_t)(threadIdx.x + 256 * i) * jb.ldw1 + min(jj, a - 1)];
            // the row's base values and the LayerNorm parameters: in flight under the head's dot products
#pragma unroll
            for (int i = 0; i < MAXQ; ++i) {
                const int n0 = 4 * (lane + 64 * i);
                pre[i] = *reinterpret_cast<const float4*>(jb.base + (size_t)row * jb.ldb + n0);
                gm[i] = *reinterpret_cast<const float4*>(jb.gamma + n0);
                bt[i] = *reinterpret_cast<const float4*>(jb.beta + n0);
            }
        }
    }
    for (int k4 = threadIdx.x; k4 < H / 4; k4 += 256) {
        float4 v[NA];
#pragma unroll
        for (int jj = 0; jj < NA; ++jj) v[jj] = reinterpret_cast<const float4*>(W4 + (size_t)jj * ldw4)[k4];
#pragma unroll
        for (int jj = 0; jj < NA; ++jj) reinterpret_cast<float4*>(ph_lds + (size_t)jj * H)[k4] = v[jj];
    }
    __syncthreads();
    constexpr int U = MAXQ > 0 ? 4 * MAXQ : 4;      // MAXQ > 0: H = 256 MAXQ, the whole row of P in flight at once (one round trip)
    float acc[NA];
#pragma unroll
    for (int jj = 0; jj < NA; ++jj) acc[jj] = 0.f;
    for (int k0 = lane; k0 < H; k0 += 64 * U) {
        float x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr (MAXQ > 0) x[u] = xrow[u];                   // (k0 == lane: the loop runs once)
            else x[u] = P[(size_t)row * ldp_ + min(k0 + 64 * u, H - 1)];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + 64 * u;
            const float xv = k < H ? x[u] : 0.f;
            const int kc = min(k, H - 1);
#pragma unroll
            for (int jj = 0; jj < NA; ++jj) acc[jj] += xv * ph_lds[(size_t)jj * H + kc];
        }
    }
    const float bias_l = b4[min(lane, NA - 1)];
    const float nz = noise != nullptr ? noise[(size_t)row * ldn + min(lane, a - 1)] : 0.f;
    float mine = 0.f, mine_raw = 0.f;              // lane j < a keeps pre[j] (and pre[a + j] when the head is 2a wide)
#pragma unroll
    for (int jj = 0; jj < NA; ++jj) {
        const float v = wave_sum(acc[jj]) + __shfl(bias_l, jj);
        mine = (jj == lane) ? v : mine;
        mine_raw = (jj == lane + a) ? v : mine_raw;
    }
    if (lane < a) {
        premu[(size_t)row * ldpre + lane] = mine;
        if (NA > a) premu[(size_t)row * ldpre + a + lane] = mine_raw;
        const float m = tanhf(mine);               // Actor: mu; DiagGaussianActor: dist.mean = tanh(loc)
        if (mu != nullptr) mu[(size_t)row * ldmu + lane] = m;
        float act = m;
        if (sq.on) {
            if (noise != nullptr) {
                const float log_std = sq.lo + 0.5f * (sq.hi - sq.lo) * (tanhf(mine_raw) + 1.f);
                act = tanhf(mine + expf(log_std) * nz);
            }
        } else if (noise != nullptr) {
            float e = nz * stddev;
            if (clip >= 0.f) e = fminf(fmaxf(e, -clip), clip);
            const float lo = (float)(-1.0 + 1e-6), hi = (float)(1.0 - 1e-6);
            act = fminf(fmaxf(m + e, lo), hi);
        }
        if (action != nullptr) action[(size_t)row * lda + lane] = act;
        mine = act;
    }
    if constexpr (MAXQ > 0) {
        if (!first) return;                        // (uniform per workgroup: every wave of it takes the barriers below)
        __syncthreads();                           // all waves are done with W4
        float* sW1 = ph_lds;
#pragma unroll
        for (int i = 0; i < MAXQ; ++i)
#pragma unroll
            for (int jj = 0; jj < NA; ++jj)
                if (jj < a) sW1[(size_t)jj * H + threadIdx.x + 256 * i] = w1r[i][jj];
        __syncthreads();
        // pre += W1[:, aoff + j] * action[j]   (lane j < a holds action[j] in ``mine``)
#pragma unroll
        for (int jj = 0; jj < NA; ++jj) {
            if (jj < a) {
                const float aj = __shfl(mine, jj);
#pragma unroll
                for (int i = 0; i < MAXQ; ++i) {
                    const float4 wv = *reinterpret_cast<const float4*>(sW1 + (size_t)jj * H + 4 * (lane + 64 * i));
                    pre[i].x += aj * wv.x; pre[i].y += aj * wv.y; pre[i].z += aj * wv.z; pre[i].w += aj * wv.w;
                }
            }
        }
        // LayerNorm + tanh: ln_tanh_fwd_kernel's arithmetic (n = H, every quad whole)
        float sm = 0.f;
#pragma unroll
        for (int i = 0; i < MAXQ; ++i) sm += (pre[i].x + pre[i].y) + (pre[i].z + pre[i].w);
        const float mean = wave_sum(sm) / (float)H;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXQ; ++i) {
            const float4 cc = make_float4(pre[i].x - mean, pre[i].y - mean, pre[i].z - mean, pre[i].w - mean);
            q += (cc.x * cc.x + cc.y * cc.y) + (cc.z * cc.z + cc.w * cc.w);
        }
        const float var = wave_sum(q) / (float)H;            // biased, like nn.LayerNorm
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
        for (int i = 0; i < MAXQ; ++i) {
            const int n0 = 4 * (lane + 64 * i);
            float4 o;
            o.x = tanhf((pre[i].x - mean) * rstd * gm[i].x + bt[i].x);
            o.y = tanhf((pre[i].y - mean) * rstd * gm[i].y + bt[i].y);
            o.z = tanhf((pre[i].z - mean) * rstd * gm[i].z + bt[i].z);
            o.w = tanhf((pre[i].w - mean) * rstd * gm[i].w + bt[i].w);
            *reinterpret_cast<float4*>(jb.t1 + (size_t)row * jb.ldt1 + n0) = o;
            if (jb.stats != nullptr) *reinterpret_cast<float4*>(const_cast<float*>(jb.base) + (size_t)row * jb.ldb + n0) = pre[i];
        }
        if (jb.stats != nullptr && lane == 0) {
            jb.stats[2 * row] = mean;
            jb.stats[2 * row + 1] = rstd;
        }
    }
}

bool policy_first_ok(int H, int a, int na) {
    return policy_head_ok(H, na) && (H == 512 || H == 1024 || H == 2048) && a >= 1 && a <= na;
}

// exact widths with an instantiation: the walker / quadruped / test actors and their 2a-wide boltzmann heads
bool policy_head_ok(int H, int na) {
    const bool inst = na == 3 || na == 6 || na == 12 || na == 24;
    return inst && (H & 3) == 0 && (size_t)na * H * sizeof(float) <= 48 * 1024;
}

#define PH_FOR_EACH(X) X(3, 0) X(6, 0) X(12, 0) X(24, 0) X(3, 2) X(6, 2) X(12, 2) X(24, 2) X(3, 4) X(6, 4) X(12, 4) X(24, 4) X(3, 8) X(6, 8) X(12, 8) X(24, 8)
hipError_t policy_head_prepare(int H, int a, int na) {
    (void)H; (void)a; (void)na;                    // na * H floats <= 48 KB (policy_head_ok): no limit to raise
    return hipSuccess;
}

hipError_t launch_policy_head(const PolicyHeadJobs& jobs, const float* W4, int ldw4, const float* b4, int ldpre, int ldn,
                              float stddev, float clip, int ldmu, int rows, int H, int a, int na, Squash sq, hipStream_t s) {
    if (!policy_head_ok(H, na) || (na != a && na != 2 * a) || (ldw4 & 3) || jobs.n < 1 || jobs.n > PH_MAX_JOBS)
        return hipErrorInvalidValue;
    if (policy_head_tiles_ok(jobs, ldw4, rows, H, a, na, sq)) {       // 16-row MFMA tiles, no LDS image (headtiles.hip)
        for (int i = 0; i < jobs.n; ++i) {
            const PolicyHeadJob& j = jobs.j[i];
            if (j.base != nullptr && (!policy_first_ok(H, a, na) || j.W1a == nullptr || j.gamma == nullptr || j.beta == nullptr || j.t1 == nullptr ||
                                      (j.ldb & 3) || (j.ldt1 & 3) || ((uintptr_t)j.base & 15) || ((uintptr_t)j.t1 & 15) ||
                                      ((uintptr_t)j.gamma & 15) || ((uintptr_t)j.beta & 15)))
                return hipErrorInvalidValue;
        }
        return launch_policy_head_tiles(jobs, W4, ldw4, b4, ldpre, ldn, stddev, clip, ldmu, rows, H, a, s);
    }
    bool first = false;
    for (int i = 0; i < jobs.n; ++i) {
        const PolicyHeadJob& j = jobs.j[i];
        if (j.base == nullptr) continue;
        first = true;
        if (!policy_first_ok(H, a, na) || j.W1a == nullptr || j.gamma == nullptr || j.beta == nullptr || j.t1 == nullptr ||
            (j.ldb & 3) || (j.ldt1 & 3) || ((uintptr_t)j.base & 15) || ((uintptr_t)j.t1 & 15) || ((uintptr_t)j.gamma & 15) ||
            ((uintptr_t)j.beta & 15))
            return hipErrorInvalidValue;
    }
    const int q = first ? H / 256 : 0;
    const size_t lds = (size_t)na * H * sizeof(float);
#define PH_LAUNCH(NB, Q)                                                                                                  \
    if (na == NB && q == Q) {                                                                                             \
        hipLaunchKernelGGL((policy_head_kernel<NB, Q>), dim3((rows + 3) / 4, jobs.n), dim3(256), lds, s, jobs, W4, ldw4, b4, \
                           ldpre, ldn, stddev, clip, ldmu, rows, H, a, sq);                                               \
        return hipGetLastError();                                                                                         \
    }
    PH_FOR_EACH(PH_LAUNCH)
#undef PH_LAUNCH
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------------------
// The seam between forward_net's data gradient and the actor's backward pass in update_actor, default (TruncatedNormal) actor:
//   d action = dt1 . W1[:, action columns]      [B,H] x [H,a]     (the last step of forward_net's dgrad, fb_modules.py:190)
//   d premu  = d action * (1 - mu^2)            straight-through clamp + tanh (utils.py:171-174)
//   d p      = (d premu . W4) * relu'(p)        [B,a] x [a,H]     (the first step of the actor's dgrad, fb_modules.py:119)
// Two GEMMs with an a-wide (6) output / contraction, each with a split-K reduce launch, are one wavefront per row here; the
// two [H, a] / [a, H] weight slices (24 KB each at walker dims) sit in LDS.  a <= 16.
constexpr int AHB_MAXA = 16;
// AHB_N: compile-time bound of a (8 / 16): the accumulator loops are fully unrolled and predicated on jj < a, so a 6-wide
// head must not pay for 16 lanes of them
// AHB_N == a exactly when EXACT (instantiated for 1..8, 12, 16): every per-output loop is straight-line code without
// predicates; other widths up to 16 take <16, false>, whose loops are predicated on jj < a
// LNE > 0: ``dt1`` is dL/d(tanh output) of forward_net's obs_action trunk and the LayerNorm+tanh backward of that row (no
// parameter gradients: forward_net's are discarded in update_actor) happens here too, LNE = ceil(H / 64) elements per lane --
// one more launch off the phase's dependency chain.  LNE == 0: ``dt1`` already is the gradient wrt the pre-LayerNorm values.
// The workgroup is 4 or 8 waves (blockDim.x = 256 / 512; one row per wave): where the two weight slices allow only ONE workgroup per
// CU (a = 12, H = 1024: 98 KB) and there are more than 256 x 4 rows, eight waves share a fill -- 256 workgroups, one round, instead of
// two rounds of 512 four-wave ones.
template <int AHB_N, bool EXACT, int LNE>
__global__ void __launch_bounds__(LNE <= 16 ? 512 : 256) actor_head_bwd_kernel(const float* __restrict__ dt1, int ldt,
                                                             const float* __restrict__ lnY, int ldy,
                                                             const float* __restrict__ lnX, int ldx,
                                                             const float* __restrict__ lnStats,
                                                             const float* __restrict__ lnGamma,
                                                             const float* __restrict__ W1a, int ldw1,
                                                             const float* __restrict__ mu, int ldmu,
                                                             const float* __restrict__ W4, int ldw4,
                                                             const float* __restrict__ P, int ldp_,
                                                             float* __restrict__ dpremu, int ldd,
                                                             float* __restrict__ dp, int lddp, int rows, int H, int a) {
    extern __shared__ float ahb_lds[];             // [a][H] (W1 action columns, transposed) then [a][H] (W4)
    float* sW1 = ahb_lds;
    float* sW4 = ahb_lds + (size_t)a * H;
    // LNE > 0: everything this wave needs of its row is requested BEFORE the LDS fill (one round trip for both)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int NT = blockDim.x;
    const int row = min((int)(blockIdx.x * (NT >> 6)) + wid, rows - 1);   // surplus waves redo the last row (no early exit: same stores)
    float dyv[LNE > 0 ? LNE : 1], yv[LNE > 0 ? LNE : 1], xv[LNE > 0 ? LNE : 1], gam[LNE > 0 ? LNE : 1], pva[LNE > 0 ? LNE : 1];
    float mean = 0.f, rstd = 0.f;
    if constexpr (LNE > 0) {
        mean = lnStats[2 * row]; rstd = lnStats[2 * row + 1];
#pragma unroll
        for (int i = 0; i < LNE; ++i) {
            const int mc = min(lane + 64 * i, H - 1);
            dyv[i] = dt1[(size_t)row * ldt + mc]; yv[i] = lnY[(size_t)row * ldy + mc]; xv[i] = lnX[(size_t)row * ldx + mc];
            gam[i] = lnGamma[mc];
            pva[i] = P[(size_t)row * ldp_ + mc];
        }
    }
    // Every global load below is UNCONDITIONAL (clamped index, mask at the use): hipcc otherwise waits for each load right
    // where a select meets it, and a 24-load fill becomes 24 round trips (measured: 27 us for this kernel).
    for (int m = threadIdx.x; m < H; m += NT) {
        float v[AHB_N];
#pragma unroll
        for (int jj = 0; jj < AHB_N; ++jj) v[jj] = W1a[(size_t)m * ldw1 + min(jj, a - 1)];
#pragma unroll
        for (int jj = 0; jj < AHB_N; ++jj)
            if (EXACT || jj < a) sW1[(size_t)jj * H + m] = v[jj];
    }
    {
        float4 v[AHB_N];
        const int k4 = min((int)threadIdx.x, H / 4 - 1);           // H / 4 <= 256 is not required: loop below covers the rest
#pragma unroll
        for (int jj = 0; jj < AHB_N; ++jj) v[jj] = reinterpret_cast<const float4*>(W4 + (size_t)min(jj, a - 1) * ldw4)[k4];
        if ((int)threadIdx.x < H / 4) {
#pragma unroll
            for (int jj = 0; jj < AHB_N; ++jj)
                if (EXACT || jj < a) reinterpret_cast<float4*>(sW4 + (size_t)jj * H)[k4] = v[jj];
        }
        for (int kk = threadIdx.x + NT; kk < H / 4; kk += NT)
#pragma unroll
            for (int jj = 0; jj < AHB_N; ++jj)
                if (EXACT || jj < a) reinterpret_cast<float4*>(sW4 + (size_t)jj * H)[kk] = reinterpret_cast<const float4*>(W4 + (size_t)jj * ldw4)[kk];
    }
    __syncthreads();
    constexpr int U = 4;                                            // elements per lane per batch of loads
    float acc[AHB_N];
#pragma unroll
    for (int jj = 0; jj < AHB_N; ++jj) acc[jj] = 0.f;
    if constexpr (LNE > 0) {
        // du = dy (1 - y^2); g = du gamma; dx = rstd (g - mean(g) - xhat mean(g xhat))      (ln_tanh_bwd_kernel, same math)
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LNE; ++i) {
            const bool in = lane + 64 * i < H;
            const float du = dyv[i] * (1.f - yv[i] * yv[i]);
            const float h = in ? (xv[i] - mean) * rstd : 0.f;
            const float g = in ? du * gam[i] : 0.f;
            dyv[i] = g; xv[i] = h;                 // reuse the registers: g and xhat
            s1 += g; s2 += g * h;
        }
        const float m1 = wave_sum(s1) / (float)H, m2 = wave_sum(s2) / (float)H;
#pragma unroll
        for (int i = 0; i < LNE; ++i) {
            const int m = lane + 64 * i, mc = min(m, H - 1);
            const float gv = m < H ? rstd * (dyv[i] - m1 - xv[i] * m2) : 0.f;
#pragma unroll
            for (int jj = 0; jj < AHB_N; ++jj)
                if (EXACT || jj < a) acc[jj] += gv * sW1[(size_t)jj * H + mc];
        }
    } else {
    for (int m0 = lane; m0 < H; m0 += 64 * U) {
        float g[U];
#pragma unroll
        for (int u = 0; u < U; ++u) g[u] = dt1[(size_t)row * ldt + min(m0 + 64 * u, H - 1)];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m = m0 + 64 * u;
            const float gv = m < H ? g[u] : 0.f;
            const int mc = min(m, H - 1);
#pragma unroll
            for (int jj = 0; jj < AHB_N; ++jj)
                if (EXACT || jj < a) acc[jj] += gv * sW1[(size_t)jj * H + mc];
        }
    }
    }
    const float mu_l = mu[(size_t)row * ldmu + min(lane, a - 1)];
#pragma unroll
    for (int jj = 0; jj < AHB_N; ++jj) {
        if (EXACT || jj < a) {
            const float m_ = __shfl(mu_l, jj);
            acc[jj] = wave_sum(acc[jj]) * (1.f - m_ * m_);
        }
    }
    if (lane < a) {
        float v = 0.f;
#pragma unroll
        for (int jj = 0; jj < AHB_N; ++jj) v = (jj == lane) ? acc[jj] : v;
        dpremu[(size_t)row * ldd + lane] = v;
    }
    if constexpr (LNE > 0) {
#pragma unroll
        for (int i = 0; i < LNE; ++i) {
            const int k = lane + 64 * i;
            const int kc = min(k, H - 1);
            float v = 0.f;
#pragma unroll
            for (int jj = 0; jj < AHB_N; ++jj)
                if (EXACT || jj < a) v += acc[jj] * sW4[(size_t)jj * H + kc];
            if (k < H) dp[(size_t)row * lddp + k] = pva[i] > 0.f ? v : 0.f;
        }
        return;
    }
    for (int k0 = lane; k0 < H; k0 += 64 * U) {
        float pv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) pv[u] = P[(size_t)row * ldp_ + min(k0 + 64 * u, H - 1)];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + 64 * u;
            const int kc = min(k, H - 1);
            float v = 0.f;
#pragma unroll
            for (int jj = 0; jj < AHB_N; ++jj)
                if (EXACT || jj < a) v += acc[jj] * sW4[(size_t)jj * H + kc];
            if (k < H) dp[(size_t)row * lddp + k] = pv[u] > 0.f ? v : 0.f;
        }
    }
}

bool actor_head_bwd_ok(int H, int a);
// Measured (same-box A/B): walker (a = 6, H = 1024, 48 KB of LDS) +2.5 % of the update over the two GEMMs + two reduces;
// quadruped (a = 12, B = 2048, 96 KB: one workgroup per CU) +0.2 %.  An earlier version with runtime-width predicated loops
// and select-guarded loads took 27 us instead of 12 and LOST at quadruped dims -- hence the exact-width instantiations.
bool actor_head_bwd_ok(int H, int a) {
    constexpr long cap = 128L * 1024;
    return a <= AHB_MAXA && (long)(2 * a * H * sizeof(float)) <= cap;
}

hipError_t actor_head_bwd_prepare(int H, int a) {
    if (!actor_head_bwd_ok(H, a)) return hipSuccess;
    const size_t bytes = (size_t)2 * a * H * sizeof(float);
    if (bytes <= 48 * 1024) return hipSuccess;
#define AHB_ATTR1(NA, EX, LN)                                                                                            \
    {                                                                                                                    \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&actor_head_bwd_kernel<NA, EX, LN>),             \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);                     \
        if (e != hipSuccess) return e;                                                                                   \
    }
#define AHB_ATTR(NA, EX) AHB_ATTR1(NA, EX, 0) AHB_ATTR1(NA, EX, 16) AHB_ATTR1(NA, EX, 32)
    switch (a) {
        case 1: AHB_ATTR(1, true); break;
        case 2: AHB_ATTR(2, true); break;
        case 3: AHB_ATTR(3, true); break;
        case 4: AHB_ATTR(4, true); break;
        case 5: AHB_ATTR(5, true); break;
        case 6: AHB_ATTR(6, true); break;
        case 7: AHB_ATTR(7, true); break;
        case 8: AHB_ATTR(8, true); break;
        case 12: AHB_ATTR(12, true); break;
        case 16: AHB_ATTR(16, true); break;
        default: AHB_ATTR(16, false); break;
    }
#undef AHB_ATTR
#undef AHB_ATTR1
    return hipSuccess;
}

hipError_t launch_actor_head_bwd(const float* dt1, int ldt, const float* W1a, int ldw1, const float* mu, int ldmu,
                                 const float* W4, int ldw4, const float* P, int ldp_, float* dpremu, int ldd, float* dp,
                                 int lddp, int rows, int H, int a, hipStream_t s, const float* lnY, int ldy, const float* lnX,
                                 int ldx, const float* lnStats, const float* lnGamma) {
    if (!actor_head_bwd_ok(H, a)) return hipErrorInvalidValue;
    const bool ln = lnY != nullptr;
    if (ln && (H > 2048 || !lnX || !lnStats || !lnGamma)) return hipErrorInvalidValue;
    if (ln && getenv("FBHIP_AHB_WAVES") == nullptr &&
        actor_head_bwd_tiles_ok(ldt, ldy, ldx, ldp_, lddp, rows, H, a, dt1, lnY, lnX, P, dp, lnGamma))       // headtiles.hip
        return launch_actor_head_bwd_tiles(dt1, ldt, W1a, ldw1, mu, ldmu, W4, ldw4, P, ldp_, dpremu, ldd, dp, lddp, rows, H, a, s, lnY,
                                           ldy, lnX, ldx, lnStats, lnGamma);
    const int lne = !ln ? 0 : (H <= 1024 ? 16 : 32);
    // eight waves per workgroup where the LDS image leaves room for one workgroup per CU only and four-wave workgroups would need a
    // second round (FBHIP_AHB_WAVES=4 / 8 forces either)
    const char* fe = getenv("FBHIP_AHB_WAVES");                      // (read at every launch: tests force either form)
    const int force = fe ? atoi(fe) : 0;
    const size_t lds_bytes = (size_t)2 * a * H * sizeof(float);
    const int nw = lne > 16 ? 4 : (force == 4 || force == 8 ? force : ((lds_bytes > 80 * 1024 && rows > 4 * 256) ? 8 : 4));   // (H > 1024: a row's registers need the four-wave budget)
#define AHB_LAUNCH1(NA, EX, LN)                                                                                          \
    hipLaunchKernelGGL((actor_head_bwd_kernel<NA, EX, LN>), dim3((rows + nw - 1) / nw), dim3(64 * nw), (size_t)2 * a * H * sizeof(float), \
                       s, dt1, ldt, lnY, ldy, lnX, ldx, lnStats, lnGamma, W1a, ldw1, mu, ldmu, W4, ldw4, P, ldp_, dpremu, ldd,  \
                       dp, lddp, rows, H, a)
#define AHB_LAUNCH(NA, EX)                                                                                               \
    {                                                                                                                    \
        if (lne == 0) AHB_LAUNCH1(NA, EX, 0);                                                                            \
        else if (lne == 16) AHB_LAUNCH1(NA, EX, 16);                                                                     \
        else AHB_LAUNCH1(NA, EX, 32);                                                                                    \
    }
    switch (a) {
        case 1: AHB_LAUNCH(1, true); break;
        case 2: AHB_LAUNCH(2, true); break;
        case 3: AHB_LAUNCH(3, true); break;
        case 4: AHB_LAUNCH(4, true); break;
        case 5: AHB_LAUNCH(5, true); break;
        case 6: AHB_LAUNCH(6, true); break;
        case 7: AHB_LAUNCH(7, true); break;
        case 8: AHB_LAUNCH(8, true); break;
        case 12: AHB_LAUNCH(12, true); break;
        case 16: AHB_LAUNCH(16, true); break;
        default: AHB_LAUNCH(16, false); break;
    }
#undef AHB_LAUNCH
#undef AHB_LAUNCH1
    return hipGetLastError();
}

// dst[r] = [A[r, :na] | B[r, :nb]]  -- builds an [obs|z] / [obs|action] panel for the inference entry points
__global__ void __launch_bounds__(256) concat2_kernel(float* __restrict__ dst, int ld, const float* __restrict__ A,
                                                      int lda, int na, const float* __restrict__ Bsrc, int ldb, int nb,
                                                      int rows) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    for (int j = lane; j < na; j += 64) dst[(size_t)r * ld + j] = A[(size_t)r * lda + j];
    if (Bsrc != nullptr)
        for (int j = lane; j < nb; j += 64) dst[(size_t)r * ld + na + j] = Bsrc[(size_t)r * ldb + j];
}

hipError_t launch_concat2(float* dst, int ld, const float* A, int lda, int na, const float* B, int ldb, int nb,
                          int rows, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(concat2_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, dst, ld, A, lda, na, B, ldb, nb, rows);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------
// DiscreteFBAgent (discrete_fb.py:277-311): the ForwardMap heads emit one embedding per action, [rows, d * A] with element
// (k, a) at column k * A + a.  One wavefront per row.
//   SELECT (target side, :289-303): Q_i[a] = sum_k Fall_i[k, a] z[k];  nq = min(Q_1, Q_2);
//       greedy:     a* = argmax_a nq (first maximum);  out_i[k] = Fall_i[k, a*];  nextq = nq[a*]
//       boltzmann:  pi = softmax(nq / temp);  out_i[k] = sum_a pi[a] Fall_i[k, a];  nextq = sum_a pi[a] nq[a]
//   GATHER (online side, :309-311): out_i[k] = Fall_i[k, action]
// lanes are (a, part): a = lane % AP (AP = A rounded up to a power of two), the 64 / AP parts split k; a butterfly over
// the parts leaves the full Q_i[a] on every lane (fixed order: deterministic)
// (blockIdx.y picks the job: the target-side selection and the online-side gather of one update share a launch)
__global__ void __launch_bounds__(256) discrete_head_kernel(const DiscreteHeadJobs jobs, int ldfa, int ldz, int ldo, int rows,
                                                            int d, int A, int AP, int boltz, float inv_temp) {
    __shared__ float s_pi[4][64];
    const DiscreteHeadJob& jb = jobs.j[blockIdx.y];
    const float* __restrict__ z = jb.z;
    const float* __restrict__ act_idx = jb.act_idx;
    float* __restrict__ out1 = jb.out1;
    float* __restrict__ out2 = jb.out2;
    float* __restrict__ nextq = jb.nextq;
    int32_t* __restrict__ act_out = jb.act_out;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, row = blockIdx.x * 4 + wid;
    if (row >= rows) return;
    const float* f1 = jb.Fall1 + (size_t)row * ldfa;
    const float* f2 = jb.Fall2 + (size_t)row * ldfa;
    if (jb.gather) {
        const int a = (int)act_idx[row];
        for (int k = lane; k < d; k += 64) {
            out1[(size_t)row * ldo + k] = f1[k * A + a];
            out2[(size_t)row * ldo + k] = f2[k * A + a];
        }
        return;
    }
    const int a = lane & (AP - 1), part = lane / AP, parts = 64 / AP;
    const int ac = a < A ? a : A - 1;
    float q1 = 0.f, q2 = 0.f;
    for (int k = part; k < d; k += parts) {
        const float zz = z[(size_t)row * ldz + k];
        q1 += f1[k * A + ac] * zz;
        q2 += f2[k * A + ac] * zz;
    }
    for (int o = AP; o < 64; o <<= 1) { q1 += __shfl_xor(q1, o); q2 += __shfl_xor(q2, o); }
    const float nq = a < A ? fminf(q1, q2) : -INFINITY;
    if (!boltz) {
        float best = nq;
        int bi = a;
        for (int o = 1; o < AP; o <<= 1) {
            const float ov = __shfl_xor(best, o);
            const int oi = __shfl_xor(bi, o);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        for (int k = lane; k < d; k += 64) {
            out1[(size_t)row * ldo + k] = f1[k * A + bi];
            out2[(size_t)row * ldo + k] = f2[k * A + bi];
        }
        if (lane == 0) {
            if (nextq) nextq[row] = best;
            if (act_out) act_out[row] = bi;
        }
        return;
    }
    const float xs = a < A ? nq * inv_temp : -INFINITY;
    float m = xs;
    for (int o = 1; o < AP; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
    const float e = a < A ? expf(xs - m) : 0.f;
    float se = e;
    for (int o = 1; o < AP; o <<= 1) se += __shfl_xor(se, o);
    const float pi = e / se;
    float pq = a < A ? pi * nq : 0.f;
    for (int o = 1; o < AP; o <<= 1) pq += __shfl_xor(pq, o);
    if (lane < AP) s_pi[wid][lane] = pi;
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k < d; k += 64) {
        float t1 = 0.f, t2 = 0.f;
        for (int b = 0; b < A; ++b) {
            const float w = s_pi[wid][b];
            t1 += w * f1[k * A + b];
            t2 += w * f2[k * A + b];
        }
        out1[(size_t)row * ldo + k] = t1;
        out2[(size_t)row * ldo + k] = t2;
    }
    if (lane == 0) {
        if (nextq) nextq[row] = pq;
        if (act_out) {                       // (the arg-max is still what ``act`` wants)
            int bi = 0;
            for (int b = 1; b < A; ++b) if (s_pi[wid][b] > s_pi[wid][bi]) bi = b;
            act_out[row] = bi;
        }
    }
}

// backward of the gather (discrete_fb.py:310): d Fall_i[k, a] = (a == action) ? dF_i[k] : 0
__global__ void __launch_bounds__(256) discrete_scatter_kernel(const float* __restrict__ dF1, const float* __restrict__ dF2,
                                                               int ldf, const float* __restrict__ act_idx,
                                                               float* __restrict__ dFall1, float* __restrict__ dFall2,
                                                               int ldfa, int rows, int d, int A) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int a = (int)act_idx[row];
    for (int j = lane; j < d * A; j += 64) {
        const int k = j / A;
        const bool hit = (j - k * A) == a;
        dFall1[(size_t)row * ldfa + j] = hit ? dF1[(size_t)row * ldf + k] : 0.f;
        dFall2[(size_t)row * ldfa + j] = hit ? dF2[(size_t)row * ldf + k] : 0.f;
    }
}

hipError_t launch_discrete_select(const float* Fall1, const float* Fall2, int ldfa, const float* z, int ldz, float* out1,
                                  float* out2, int ldo, float* nextq, int32_t* act_out, int rows, int d, int A, int boltz,
                                  float temp, hipStream_t s) {
    DiscreteHeadJobs jobs{};
    jobs.n = 1;
    jobs.j[0] = DiscreteHeadJob{Fall1, Fall2, z, nullptr, out1, out2, nextq, act_out, 0};
    return launch_discrete_heads(jobs, ldfa, ldz, ldo, rows, d, A, boltz, temp, s);
}
hipError_t launch_discrete_heads(const DiscreteHeadJobs& jobs, int ldfa, int ldz, int ldo, int rows, int d, int A, int boltz,
                                 float temp, hipStream_t s) {
    if (A < 1 || A > 64 || jobs.n < 1 || jobs.n > 2) return hipErrorInvalidValue;
    int AP = 1;
    while (AP < A) AP <<= 1;
    hipLaunchKernelGGL(discrete_head_kernel, dim3((rows + 3) / 4, jobs.n), dim3(256), 0, s, jobs, ldfa, ldz, ldo, rows, d, A, AP,
                       boltz, 1.0f / temp);
    return hipGetLastError();
}
hipError_t launch_discrete_gather(const float* Fall1, const float* Fall2, int ldfa, const float* act_idx, float* out1,
                                  float* out2, int ldo, int rows, int d, int A, hipStream_t s) {
    DiscreteHeadJobs jobs{};
    jobs.n = 1;
    jobs.j[0] = DiscreteHeadJob{Fall1, Fall2, nullptr, act_idx, out1, out2, nullptr, nullptr, 1};
    return launch_discrete_heads(jobs, ldfa, 0, ldo, rows, d, A, 0, 1.f, s);
}
hipError_t launch_discrete_scatter(const float* dF1, const float* dF2, int ldf, const float* act_idx, float* dFall1,
                                   float* dFall2, int ldfa, int rows, int d, int A, hipStream_t s) {
    hipLaunchKernelGGL(discrete_scatter_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, dF1, dF2, ldf, act_idx, dFall1, dFall2,
                       ldfa, rows, d, A);
    return hipGetLastError();
}

}  // namespace fbhip
