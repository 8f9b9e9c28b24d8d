// Fused Adam + soft-target EMA over a flat fp32 segment, and the tiny in-graph step-state kernel.
//   torch.optim.Adam defaults (fb_ddpg.py:146-151): betas (0.9, 0.999), eps 1e-8, no weight decay:
//     m <- m + (1-b1)(g - m);  v <- b2 v + (1-b2) g g;  p <- p - (lr/(1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
//   utils.soft_update_params (utils.py:66-69):  target <- tau p + (1-tau) target
// The reference issues 28 x (lerp, mul, addcmul, sqrt, div, add, addcdiv) + 28 x 4 EMA launches per FB step;
// here it is ONE pass over {p, g, m, v, target}: 5 reads + 4 writes of 16 B per lane.  Fusing the EMA into the
// FB Adam pass is legal because nothing between fb_opt.step() and soft_update_params reads a target net or
// writes forward_net/backward_net.
#include "common.h"
#include "fbhip.h"

namespace fbhip {

__global__ void step_advance_kernel(StepState* st, int which) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    step_advance_device(st, which);
}

hipError_t launch_step_advance(StepState* st, int which, hipStream_t s) {
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(64), 0, s, st, which);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) adam_ema_kernel(float4* __restrict__ p, const float4* __restrict__ g,
                                                       float4* __restrict__ m, float4* __restrict__ v,
                                                       float4* __restrict__ tgt, int64_t nquad, float lr, float lr2,
                                                       int64_t split_quad, float grad_scale, float tau,
                                                       const StepState* __restrict__ st, int which, int t_explicit,
                                                       float tau2, int ema_before2) {
    double bc1, bc2s;
    if (st != nullptr) {
        bc1 = which == 0 ? st->fb_bc1 : st->actor_bc1;
        bc2s = which == 0 ? st->fb_bc2_sqrt : st->actor_bc2_sqrt;
    } else {
        bc1 = 1.0 - pow(0.9, (double)t_explicit);
        bc2s = sqrt(1.0 - pow(0.999, (double)t_explicit));
    }
    const float ss1 = (float)((double)lr / bc1), ss2 = (float)((double)lr2 / bc1);
    const float bc2f = (float)bc2s;
    const float w1 = (float)(1.0 - 0.9), w2 = (float)(1.0 - 0.999), b2 = 0.999f, eps = 1e-8f;
    const float omt = (float)(1.0 - (double)tau);
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nquad; i += stride) {
        const bool second = i >= split_quad;
        const float ss = second ? ss2 : ss1;
        float4 gg = g[i], mm = m[i], vv = v[i], pp = p[i];
        float4 tt = (tgt != nullptr ? tgt : p)[i];          // (in flight with the other four streams, not behind the stores)
        const float4 p0 = pp;
#define ADAM1(c)                                                         \
        {                                                                \
            const float gr = gg.c * grad_scale;                          \
            mm.c = mm.c + w1 * (gr - mm.c);                              \
            vv.c = vv.c * b2 + (w2 * gr) * gr;                           \
            const float den = sqrtf(vv.c) / bc2f + eps;                  \
            pp.c = pp.c - ss * (mm.c / den);                             \
        }
        ADAM1(x) ADAM1(y) ADAM1(z) ADAM1(w)
#undef ADAM1
        m[i] = mm; v[i] = vv; p[i] = pp;
        if (tgt != nullptr) {
            // (sf.py:237 TransitionLatentModel: the learner's own target net is moved inside its forward(), i.e. towards the
            // parameters BEFORE phi_opt.step(), at its own fixed rate)
            const float4 src = (second && ema_before2) ? p0 : pp;
            const float ta = second ? tau2 : tau, om = second ? (float)(1.0 - (double)tau2) : omt;
            tt.x = ta * src.x + om * tt.x; tt.y = ta * src.y + om * tt.y;
            tt.z = ta * src.z + om * tt.z; tt.w = ta * src.w + om * tt.w;
            tgt[i] = tt;
        }
    }
}

hipError_t launch_adam_ema(float* p, const float* g, float* m, float* v, float* target, int64_t numel, float lr,
                           float lr2, int64_t split, float grad_scale, float tau, const StepState* st, int which,
                           int t_explicit, hipStream_t s, float tau2, int ema_before2) {
    if (numel <= 0) return hipSuccess;
    if ((numel & 3) || (split & 3)) return hipErrorInvalidValue;
    const int64_t nquad = numel / 4;
    int blocks = (int)((nquad + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adam_ema_kernel, dim3(blocks), dim3(256), 0, s, (float4*)p, (const float4*)g, (float4*)m,
                       (float4*)v, (float4*)target, nquad, lr, lr2, split / 4, grad_scale, tau, st, which, t_explicit,
                       tau2 < 0.f ? tau : tau2, ema_before2);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Extra metrics of fb_ddpg.py:356-377 that are not by-products of the pairwise kernel: F1.mean(), B.mean(),
// mean row norms of B and z, and max|.| / Frobenius of (B^T B / batch - I).  One workgroup; metric-only path.
__global__ void __launch_bounds__(1024) extra_metrics_kernel(const float* __restrict__ F1, const float* __restrict__ Bm,
                                                             const float* __restrict__ z, int ld, int rows, int d,
                                                             const float* __restrict__ cov /*[d,d] = B^T B*/,
                                                             int ldc, float* __restrict__ metrics, int cov_rows) {
    __shared__ double red[16][6];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    double sF = 0, sB = 0, sBn = 0, sZn = 0, sL2 = 0, mx = 0;
    for (int r = wid; r < rows; r += 16) {
        float f = 0.f, b = 0.f, b2 = 0.f, z2 = 0.f;
        for (int j = lane; j < d; j += 64) {
            const float bv = Bm[(size_t)r * ld + j], zv = z[(size_t)r * ld + j];
            f += F1[(size_t)r * ld + j]; b += bv; b2 += bv * bv; z2 += zv * zv;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            f += __shfl_xor(f, o); b += __shfl_xor(b, o); b2 += __shfl_xor(b2, o); z2 += __shfl_xor(z2, o);
        }
        sF += f; sB += b; sBn += sqrtf(b2); sZn += sqrtf(z2);
    }
    for (int e = tid; e < d * d; e += 1024) {
        const int i = e / d, j = e % d;
        const float v = cov[(size_t)i * ldc + j] / (float)cov_rows - (i == j ? 1.f : 0.f);
        sL2 += (double)v * v;
        mx = fmax(mx, (double)fabsf(v));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sL2 += __shfl_xor(sL2, o);
        mx = fmax(mx, __shfl_xor(mx, o));
    }
    if (lane == 0) { red[wid][0] = sF; red[wid][1] = sB; red[wid][2] = sBn; red[wid][3] = sZn; red[wid][4] = sL2; red[wid][5] = mx; }
    __syncthreads();
    if (tid == 0) {
        double a[6] = {0, 0, 0, 0, 0, 0};
        for (int w = 0; w < 16; ++w) {
            for (int k = 0; k < 5; ++k) a[k] += red[w][k];
            a[5] = fmax(a[5], red[w][5]);
        }
        metrics[FBHIP_M_F1] = (float)(a[0] / ((double)rows * d));
        metrics[FBHIP_M_B] = (float)(a[1] / ((double)rows * d));
        metrics[FBHIP_M_B_NORM] = (float)(a[2] / rows);
        metrics[FBHIP_M_Z_NORM] = (float)(a[3] / rows);
        metrics[FBHIP_M_ORTH_L2] = (float)(sqrt(a[4]) / sqrt((double)d));
        metrics[FBHIP_M_ORTH_LINF] = (float)a[5];
    }
}

// The same metrics from many workgroups (round 6: the one-workgroup form above walks 64 rows per wave behind a shuffle reduction
// each -- 75 us at B = 1024 on the critical path of every metrics-on update, README.md:50's invocation).  A workgroup owns 64 rows,
// FOUR LANES per row (independent loads, two shuffles per row), and leaves four fp64 partial sums; the workgroup that draws the
// last ticket folds all partials IN INDEX ORDER (deterministic whatever the arrival order), adds the B^T B statistics and writes
// the six metrics.  ``part`` = [gridDim.x][4] doubles, ``ticket`` = one zero-initialised counter, reset by the last workgroup.
constexpr int XM_ROWS = 64;
__global__ void __launch_bounds__(256) extra_metrics_wide_kernel(const float* __restrict__ F1, const float* __restrict__ Bm,
                                                                  const float* __restrict__ z, int ld, int rows, int d,
                                                                  const float* __restrict__ cov, int ldc, float* __restrict__ metrics,
                                                                  int cov_rows, double* __restrict__ part, unsigned int* ticket) {
    __shared__ double red[4][4];
    __shared__ double red2[4][2];
    __shared__ unsigned int s_last;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, q = lane & 3;
    const int r = blockIdx.x * XM_ROWS + wid * 16 + (lane >> 2);
    const int rc = min(r, rows - 1);
    float f = 0.f, b = 0.f, b2 = 0.f, z2 = 0.f;
    for (int j = q; j < d; j += 4) {
        const float bv = Bm[(size_t)rc * ld + j], zv = z[(size_t)rc * ld + j];
        f += F1[(size_t)rc * ld + j]; b += bv; b2 += bv * bv; z2 += zv * zv;
    }
#pragma unroll
    for (int o = 1; o <= 2; o <<= 1) { f += __shfl_xor(f, o); b += __shfl_xor(b, o); b2 += __shfl_xor(b2, o); z2 += __shfl_xor(z2, o); }
    const bool mine = q == 0 && r < rows;
    double sF = mine ? (double)f : 0.0, sB = mine ? (double)b : 0.0, sBn = mine ? (double)sqrtf(b2) : 0.0, sZn = mine ? (double)sqrtf(z2) : 0.0;
#pragma unroll
    for (int o = 4; o <= 32; o <<= 1) { sF += __shfl_xor(sF, o); sB += __shfl_xor(sB, o); sBn += __shfl_xor(sBn, o); sZn += __shfl_xor(sZn, o); }
    if (lane == 0) { red[wid][0] = sF; red[wid][1] = sB; red[wid][2] = sBn; red[wid][3] = sZn; }
    __syncthreads();
    if (tid < 4) part[4 * blockIdx.x + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    __threadfence();
    __syncthreads();
    if (tid == 0) s_last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    // ---- the last workgroup: B^T B / rows - I, then the fold
    double sL2 = 0.0, mx = 0.0;
    for (int e = tid; e < d * d; e += 256) {
        const int i = e / d, j = e % d;
        const float v = cov[(size_t)i * ldc + j] / (float)cov_rows - (i == j ? 1.f : 0.f);
        sL2 += (double)v * v;
        mx = fmax(mx, (double)fabsf(v));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sL2 += __shfl_xor(sL2, o); mx = fmax(mx, __shfl_xor(mx, o)); }
    if (lane == 0) { red2[wid][0] = sL2; red2[wid][1] = mx; }
    __syncthreads();
    if (tid == 0) {
        double a[4] = {0, 0, 0, 0};
        const volatile double* pv = part;
        for (unsigned int g = 0; g < gridDim.x; ++g)
            for (int k = 0; k < 4; ++k) a[k] += pv[4 * g + k];
        const double l2 = (red2[0][0] + red2[1][0]) + (red2[2][0] + red2[3][0]);
        const double linf = fmax(fmax(red2[0][1], red2[1][1]), fmax(red2[2][1], red2[3][1]));
        metrics[FBHIP_M_F1] = (float)(a[0] / ((double)rows * d));
        metrics[FBHIP_M_B] = (float)(a[1] / ((double)rows * d));
        metrics[FBHIP_M_B_NORM] = (float)(a[2] / rows);
        metrics[FBHIP_M_Z_NORM] = (float)(a[3] / rows);
        metrics[FBHIP_M_ORTH_L2] = (float)(sqrt(l2) / sqrt((double)d));
        metrics[FBHIP_M_ORTH_LINF] = (float)linf;
        *ticket = 0u;                            // (the next launch -- the next graph replay -- starts from zero again)
    }
}

hipError_t launch_extra_metrics(const float* F1, const float* Bm, const float* z, int ld, int rows, int d,
                                const float* cov, int ldc, float* metrics, hipStream_t s, int cov_rows, double* part,
                                unsigned int* ticket) {
    const int nblk = (rows + XM_ROWS - 1) / XM_ROWS;
    if (part != nullptr && ticket != nullptr && nblk >= 2 && nblk <= EXTRA_METRICS_MAX_BLOCKS) {
        hipLaunchKernelGGL(extra_metrics_wide_kernel, dim3(nblk), dim3(256), 0, s, F1, Bm, z, ld, rows, d, cov, ldc, metrics,
                           cov_rows > 0 ? cov_rows : rows, part, ticket);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(extra_metrics_kernel, dim3(1), dim3(1024), 0, s, F1, Bm, z, ld, rows, d, cov, ldc, metrics,
                       cov_rows > 0 ? cov_rows : rows);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// q_loss branch of update_fb (fb_ddpg.py:330-340).
//   cov = B^T B / batch;  inv_cov = inverse(cov)                       -> inverse_kernel (d x d, one workgroup)
//   implicit_reward = rowsum((B inv_cov) * z);  next_Q = min(tF1.z, tF2.z);  target_Q = ir + discount * next_Q
//   q_loss = mse(F1.z, target_Q) + mse(F2.z, target_Q);  dF_i += coef * 2 (F_i.z - target_Q) / batch * z
// Gauss-Jordan with partial pivoting in fp64 (the matrix is tiny; fp64 keeps the result at the conditioning of the fp32 input
// rather than adding a second fp32 round-off on top of torch.inverse's), by ONE workgroup of 1024 threads that keeps the matrix
// in REGISTERS: thread (ri, cj) = (t % 32, t / 32) owns the 4 x 4 elements of rows ri + 32 a, columns cj + 32 b (d <= 128).
// Per pivot step k only two vectors cross threads, through LDS: column k (written by its 32 owners, one half-wave, which also
// pick the pivot among the rows not used so far with five xor-shuffles and compute 1 / pivot once) and the scaled pivot row
// (written by its 32 owners after the first barrier); then every thread reads 4 + 4 values and updates its 16 elements.
// Rows are never exchanged: the pivot order sigma is remembered and the result is scattered as
//   inverse[a][b] = M_final[sigma(a)][sigma^-1(b)]      (Gauss-Jordan without pivoting applied to the row-permuted matrix).
// d = 100, measured inside the SF step where it runs beside the previous step's actor phase (profiles/r02g_sf_lap_mix_*): 0.24 ms,
// i.e. 2.4 us per pivot step (two barriers of 16 waves + the shuffle reduction on a CU it shares with GEMM workgroups).  The
// first version (256 threads over an LDS-resident matrix, an fp64 division per element, four barriers per step) took 0.85 ms --
// more than a whole FB update -- and a 1024-thread LDS-resident one 0.34 ms.
constexpr int INV_T = 1024;
__device__ __forceinline__ double rcp_f64(double x) {          // v_rcp_f64 seed + two Newton steps: ~1 ulp, no div expansion
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    r = fma(r, fma(-x, r, 1.0), r);
    return r;
}
__global__ void __launch_bounds__(INV_T) inverse_kernel(const float* __restrict__ A, int lda, int d, float scale,
                                                        float* __restrict__ out, int ldo) {
    __shared__ double colk[2][128], rowk[128], s_rkk[2];
    __shared__ int s_p[2], piv[128], pinv[128];
    const int tid = threadIdx.x, ri = tid & 31, cj = tid >> 5, wave = tid >> 6;
    double m[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int i = ri + 32 * a, j = cj + 32 * b;
            m[a][b] = (i < d && j < d) ? (double)A[(size_t)i * lda + j] * (double)scale : 0.0;
        }
    unsigned used = 0;                          // bit a: row ri + 32 a has been a pivot row
    for (int k = 0; k < d; ++k) {
        const int par = k & 1, kb = k >> 5, kc = k & 31;
        if (wave == (kc >> 1)) {                // the wave that holds column k (the other half-wave runs along, writes nothing)
            const bool owner = cj == kc;
            double v[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) v[a] = kb == 0 ? m[a][0] : kb == 1 ? m[a][1] : kb == 2 ? m[a][2] : m[a][3];
            double bv = 0.0, babs = -1.0;
            int bi = 0x7fffffff;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int i = ri + 32 * a;
                if (owner) colk[par][i] = v[a];
                const double av = (i < d && !((used >> a) & 1u)) ? fabs(v[a]) : -1.0;
                if (av > babs) { babs = av; bv = v[a]; bi = i; }
            }
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) {   // within the half-wave; lowest row on ties
                const double ov = __shfl_xor(bv, o), oa = __shfl_xor(babs, o);
                const int oi = __shfl_xor(bi, o);
                if (oa > babs || (oa == babs && oi < bi)) { babs = oa; bv = ov; bi = oi; }
            }
            if ((unsigned)bi >= (unsigned)d) bi = k;           // (only with NaNs in the input: garbage out, but in bounds)
            if (owner && ri == 0) { s_p[par] = bi; s_rkk[par] = rcp_f64(bv); piv[k] = bi; pinv[bi] = k; }
        }
        __syncthreads();
        const int p = s_p[par];
        const double rkk = s_rkk[par];
        if (ri == (p & 31)) {                   // the owners of the pivot row: scaled row -> LDS
            const int pa = p >> 5;
            used |= 1u << pa;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int j = cj + 32 * b;
                const double x = pa == 0 ? m[0][b] : pa == 1 ? m[1][b] : pa == 2 ? m[2][b] : m[3][b];
                rowk[j] = (j == k) ? rkk : x * rkk;
            }
        }
        __syncthreads();
        double rj[4], ca[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) rj[b] = rowk[cj + 32 * b];
#pragma unroll
        for (int a = 0; a < 4; ++a) ca[a] = colk[par][ri + 32 * a];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const bool isp = (ri + 32 * a) == p, isk = (cj + 32 * b) == k;
                m[a][b] = isp ? rj[b] : (isk ? -ca[a] * rkk : fma(-ca[a], rj[b], m[a][b]));
            }
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int i = ri + 32 * a, j = cj + 32 * b;
            if (i < d && j < d) out[(size_t)pinv[i] * ldo + piv[j]] = (float)m[a][b];
        }
}

hipError_t launch_inverse(const float* A, int lda, int d, float scale, float* out, int ldo, hipStream_t s) {
    if (d < 1 || d > 128) return hipErrorInvalidValue;
    hipLaunchKernelGGL(inverse_kernel, dim3(1), dim3(INV_T), 0, s, A, lda, d, scale, out, ldo);
    return hipGetLastError();
}

hipError_t inverse_prepare() { return hipSuccess; }      // (static LDS only)

// one wavefront per row; partial sums of the two squared errors go to ``part`` (2 floats per workgroup)
__global__ void __launch_bounds__(256) qloss_kernel(const float* __restrict__ F1, const float* __restrict__ F2,
                                                    const float* __restrict__ tF1, const float* __restrict__ tF2,
                                                    const float* __restrict__ BinvC, const float* __restrict__ z,
                                                    int ld, const float* __restrict__ discount, float coef,
                                                    float* __restrict__ dF1, float* __restrict__ dF2,
                                                    float* __restrict__ part, int rows, int d, int norm_rows,
                                                    const float* __restrict__ nextq) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, row = blockIdx.x * 4 + wid;
    float sq = 0.f;
    if (row < rows) {
        float q1 = 0.f, q2 = 0.f, n1 = 0.f, n2 = 0.f, ir = 0.f;
        for (int j = lane; j < d; j += 64) {
            const size_t o = (size_t)row * ld + j;
            const float zz = z[o];
            q1 += F1[o] * zz; q2 += F2[o] * zz; n1 += tF1[o] * zz; n2 += tF2[o] * zz; ir += BinvC[o] * zz;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            q1 += __shfl_xor(q1, off); q2 += __shfl_xor(q2, off); n1 += __shfl_xor(n1, off);
            n2 += __shfl_xor(n2, off); ir += __shfl_xor(ir, off);
        }
        const float tq = ir + discount[row] * (nextq != nullptr ? nextq[row] : fminf(n1, n2));
        const float e1 = q1 - tq, e2 = q2 - tq;
        const float g1 = coef * 2.f * e1 / (float)norm_rows, g2 = coef * 2.f * e2 / (float)norm_rows;
        for (int j = lane; j < d; j += 64) {
            const size_t o = (size_t)row * ld + j;
            const float zz = z[o];
            dF1[o] += g1 * zz;
            dF2[o] += g2 * zz;
        }
        sq = e1 * e1 + e2 * e2;
    }
    if (lane == 0) red[wid] = sq;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(64) qloss_finalize_kernel(const float* __restrict__ part, int nblk, int rows,
                                                            float coef, float* __restrict__ metrics) {
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 64) s += (double)part[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) {
        const float q = (float)(s / rows);
        metrics[FBHIP_M_Q_LOSS] = q;
        metrics[FBHIP_M_FB_LOSS] += coef * q;        // fb_loss += q_loss_coef * q_loss (fb_ddpg.py:340)
    }
}

hipError_t launch_qloss(const float* F1, const float* F2, const float* tF1, const float* tF2, const float* BinvC,
                        const float* z, int ld, const float* discount, float coef, float* dF1, float* dF2,
                        float* metrics, float* scratch, int rows, int d, hipStream_t s, int norm_rows, const float* nextq) {
    const int nblk = (rows + 3) / 4;
    if (norm_rows <= 0) norm_rows = rows;        // > rows: these rows are a block of a larger (global) batch
    hipLaunchKernelGGL(qloss_kernel, dim3(nblk), dim3(256), 0, s, F1, F2, tF1, tF2, BinvC, z, ld, discount, coef, dF1,
                       dF2, scratch, rows, d, norm_rows, nextq);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(qloss_finalize_kernel, dim3(1), dim3(64), 0, s, scratch, nblk, norm_rows, coef, metrics);
    return hipGetLastError();
}

}  // namespace fbhip
