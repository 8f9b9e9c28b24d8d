// Row kernels of the successor-feature sibling (url_benchmark/agent/sf.py, SFAgent.update_sf :594-664 and the feature
// learners ICM :194-213 / Laplacian :100-116) for gfx950.  Everything dense -- the Actor / ForwardMap ("successor_net") /
// feature_net passes and the inverse-dynamics mlp -- runs on the grouped GEMM, LayerNorm and projection kernels of the FB
// step; what is specific to SFAgent is three small reductions, one wavefront per batch row, with fixed-order folds
// (deterministic, no atomics):
//
//   sf_loss_kernel   target_F = phi(next_goal) + discount * next_F[argmin_i next_F_i . z]          (sf.py:614-616)
//                    q_loss:  sum_i mse(F_i . z, target_F . z)   else  sum_i mse(F_i, target_F)      (sf.py:619-626)
//                    and its gradient wrt F1, F2 (what sf_loss.backward() hands to successor_net's heads)
//   icm_loss_kernel  pred = tanh(pre);  loss = mean((action - pred)^2);  d pre                       (sf.py:207-210)
//   lap_kernel       mean((phi - next_phi)^2) and its gradient wrt both, on top of the orthonormality gradient that
//                    pairwise_kernel (called with zero F panels) left in d phi                        (sf.py:104-114)
#include "common.h"
#include "fbhip.h"

namespace fbhip {

namespace {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

constexpr int SF_PART = 6;      // per-workgroup partial sums: squared error, target_F, F1, phi, |phi|, |z|

__global__ void __launch_bounds__(256) sf_loss_kernel(const float* __restrict__ F1, const float* __restrict__ F2,
                                                      const float* __restrict__ nF1, const float* __restrict__ nF2,
                                                      const float* __restrict__ phi, const float* __restrict__ z, int ld,
                                                      const float* __restrict__ discount, int q_loss,
                                                      float* __restrict__ dF1, float* __restrict__ dF2,
                                                      float* __restrict__ part, int rows, int d) {
    __shared__ float red[4][SF_PART];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, row = blockIdx.x * 4 + wid;
    float acc[SF_PART] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (row < rows) {
        const size_t base = (size_t)row * ld;
        float q1 = 0.f, q2 = 0.f, n1 = 0.f, n2 = 0.f, pz = 0.f, pp = 0.f, zz2 = 0.f, sF1 = 0.f, sphi = 0.f;
        for (int j = lane; j < d; j += 64) {
            const float zz = z[base + j], ph = phi[base + j], f1 = F1[base + j];
            q1 += f1 * zz; q2 += F2[base + j] * zz; n1 += nF1[base + j] * zz; n2 += nF2[base + j] * zz;
            pz += ph * zz; pp += ph * ph; zz2 += zz * zz; sF1 += f1; sphi += ph;
        }
        q1 = wsum(q1); q2 = wsum(q2); n1 = wsum(n1); n2 = wsum(n2); pz = wsum(pz); pp = wsum(pp); zz2 = wsum(zz2);
        sF1 = wsum(sF1); sphi = wsum(sphi);
        const bool first = n1 < n2;                                   // torch.where((next_Q1 < next_Q2), next_F1, next_F2)
        const float gam = discount[row];
        const float* __restrict__ nF = first ? nF1 : nF2;
        float sq = 0.f, stf = 0.f;
        if (q_loss) {                                                 // scalar regression on Q = F . z
            const float tq = pz + gam * (first ? n1 : n2);           // target_F . z
            const float e1 = q1 - tq, e2 = q2 - tq;
            const float g1 = 2.f * e1 / (float)rows, g2 = 2.f * e2 / (float)rows;
            for (int j = lane; j < d; j += 64) {
                const float zz = z[base + j];
                dF1[base + j] = g1 * zz;
                dF2[base + j] = g2 * zz;
                stf += phi[base + j] + gam * nF[base + j];
            }
            sq = e1 * e1 + e2 * e2;                                   // / rows in the finalize
        } else {                                                      // regression in feature space, mean over rows * d
            const float sc = 2.f / ((float)rows * (float)d);
            for (int j = lane; j < d; j += 64) {
                const float tf = phi[base + j] + gam * nF[base + j];
                const float e1 = F1[base + j] - tf, e2 = F2[base + j] - tf;
                dF1[base + j] = sc * e1;
                dF2[base + j] = sc * e2;
                sq += e1 * e1 + e2 * e2;
                stf += tf;
            }
            sq = wsum(sq);
        }
        stf = wsum(stf);
        acc[0] = sq; acc[1] = stf; acc[2] = sF1; acc[3] = sphi; acc[4] = sqrtf(pp); acc[5] = sqrtf(zz2);
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < SF_PART; ++k) red[wid][k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < SF_PART) {
        const int k = threadIdx.x;
        part[(size_t)blockIdx.x * SF_PART + k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
    }
}

__global__ void __launch_bounds__(64) sf_loss_finalize_kernel(const float* __restrict__ part, int nblk, int rows, int d,
                                                              int q_loss, float* __restrict__ metrics) {
    double s[SF_PART] = {0, 0, 0, 0, 0, 0};
    for (int b = threadIdx.x; b < nblk; b += 64)
#pragma unroll
        for (int k = 0; k < SF_PART; ++k) s[k] += (double)part[(size_t)b * SF_PART + k];
#pragma unroll
    for (int k = 0; k < SF_PART; ++k)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s[k] += __shfl_xor(s[k], o);
    if (threadIdx.x == 0) {
        const double n = (double)rows, nd = n * (double)d;
        metrics[FBHIP_M_SF_LOSS] = (float)(s[0] / (q_loss ? n : nd));
        metrics[FBHIP_M_SF_TARGET_F] = (float)(s[1] / nd);
        metrics[FBHIP_M_F1] = (float)(s[2] / nd);
        metrics[FBHIP_M_SF_PHI] = (float)(s[3] / nd);
        metrics[FBHIP_M_SF_PHI_NORM] = (float)(s[4] / n);
        metrics[FBHIP_M_Z_NORM] = (float)(s[5] / n);
    }
}

// one thread per element of the [rows, a] inverse-dynamics output; per-workgroup partial sums folded in a fixed order
__global__ void __launch_bounds__(256) icm_loss_kernel(const float* __restrict__ pre, int ldp, const float* __restrict__ action,
                                                       int lda, float* __restrict__ dpre, int ldd, int rows, int a,
                                                       int squash, float* __restrict__ part) {
    __shared__ float red[4];
    const float sc = 2.f / ((float)rows * (float)a);
    const int e = blockIdx.x * 256 + threadIdx.x;
    float sq = 0.f;
    if (e < rows * a) {
        const int r = e / a, j = e - r * a;
        const float x = pre[(size_t)r * ldp + j];
        const float pred = squash ? tanhf(x) : x;                       // (autoencoder / transition heads are linear)
        const float err = action[(size_t)r * lda + j] - pred;
        dpre[(size_t)r * ldd + j] = -sc * err * (squash ? 1.f - pred * pred : 1.f);
        sq = err * err;
    }
    sq = wsum(sq);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(64) icm_loss_finalize_kernel(const float* __restrict__ part, int nblk, int rows, int a,
                                                               float* __restrict__ metrics) {
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 64) s += (double)part[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) metrics[FBHIP_M_PHI_LOSS] = (float)(s / ((double)rows * (double)a));
}

// d phi += 2 (phi - next_phi) / (rows d);  d next_phi = -2 (phi - next_phi) / (rows d);  partial sums of the squared difference
__global__ void __launch_bounds__(256) lap_kernel(const float* __restrict__ phi, const float* __restrict__ nphi, int ld,
                                                  float* __restrict__ dphi, float* __restrict__ dnphi, float* __restrict__ part,
                                                  int rows, int d) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, row = blockIdx.x * 4 + wid;
    float sq = 0.f;
    if (row < rows) {
        const size_t base = (size_t)row * ld;
        const float sc = 2.f / ((float)rows * (float)d);
        for (int j = lane; j < d; j += 64) {
            const float df = phi[base + j] - nphi[base + j];
            dphi[base + j] += sc * df;
            dnphi[base + j] = -sc * df;
            sq += df * df;
        }
        sq = wsum(sq);
    }
    if (lane == 0) red[wid] = sq;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(64) lap_finalize_kernel(const float* __restrict__ part, int nblk, int rows, int d,
                                                          float* __restrict__ metrics) {
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 64) s += (double)part[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    // the pairwise kernel, run with zero F panels and ortho_coef 1, left orth_loss_offdiag + orth_loss_diag in ORTH_LOSS
    if (threadIdx.x == 0) metrics[FBHIP_M_PHI_LOSS] = (float)(s / ((double)rows * (double)d)) + metrics[FBHIP_M_ORTH_LOSS];
}

}  // namespace

hipError_t launch_sf_loss(const float* F1, const float* F2, const float* nF1, const float* nF2, const float* phi_next,
                          const float* z, int ld, const float* discount, int q_loss, float* dF1, float* dF2, float* metrics,
                          float* scratch, int rows, int d, hipStream_t s) {
    const int nblk = (rows + 3) / 4;
    hipLaunchKernelGGL(sf_loss_kernel, dim3(nblk), dim3(256), 0, s, F1, F2, nF1, nF2, phi_next, z, ld, discount, q_loss, dF1, dF2,
                       scratch, rows, d);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sf_loss_finalize_kernel, dim3(1), dim3(64), 0, s, scratch, nblk, rows, d, q_loss, metrics);
    return hipGetLastError();
}

hipError_t launch_icm_loss(const float* pre, int ldp, const float* action, int lda, float* dpre, int ldd, int rows, int a,
                           int squash, float* metrics, float* scratch, hipStream_t s) {
    const int nblk = (rows * a + 255) / 256;
    hipLaunchKernelGGL(icm_loss_kernel, dim3(nblk), dim3(256), 0, s, pre, ldp, action, lda, dpre, ldd, rows, a, squash, scratch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(icm_loss_finalize_kernel, dim3(1), dim3(64), 0, s, scratch, nblk, rows, a, metrics);
    return hipGetLastError();
}

__global__ void scale_metric_kernel(float* __restrict__ metrics, int src, int dst, float scale, int accumulate) {
    metrics[dst] = (accumulate ? metrics[dst] : 0.f) + scale * metrics[src];
}

hipError_t launch_scale_metric(float* metrics, int src, int dst, float scale, hipStream_t s, int accumulate) {
    hipLaunchKernelGGL(scale_metric_kernel, dim3(1), dim3(1), 0, s, metrics, src, dst, scale, accumulate);
    return hipGetLastError();
}

// one wavefront per row s of the [B, B] logit matrix
__global__ void __launch_bounds__(256) contrastive_rows_kernel(float* __restrict__ L, int ld, int B, float inv_d, float* __restrict__ part) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, s = blockIdx.x * 4 + (threadIdx.x >> 6);
    float loss = 0.f;
    if (s < B) {
        float* row = L + (size_t)s * ld;
        float mx = -3.0e38f;
        for (int t = lane; t < B; t += 64)
            if (t != s) mx = fmaxf(mx, row[t] * inv_d);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float se = 0.f;
        for (int t = lane; t < B; t += 64)
            if (t != s) se += expf(row[t] * inv_d - mx);
        se = wsum(se);
        const float lse = mx + logf(se);
        const float lss = row[s] * inv_d;
        const float gs = inv_d / (float)B;                                   // d mean / d l  times  d l / d L
        for (int t = lane; t < B; t += 64) row[t] = t == s ? -gs : gs * expf(row[t] * inv_d - lse);
        loss = lse - lss;
    }
    if (lane == 0) red[threadIdx.x >> 6] = loss;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(64) contrastive_finalize_kernel(const float* __restrict__ part, int nblk, int B, float* __restrict__ metrics) {
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 64) s += (double)part[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) metrics[FBHIP_M_PHI_LOSS] = (float)(s / (double)B);
}

hipError_t launch_contrastive_rows(float* L, int ld, int B, int d, float* metrics, float* scratch, hipStream_t s) {
    const int nblk = (B + 3) / 4;
    hipLaunchKernelGGL(contrastive_rows_kernel, dim3(nblk), dim3(256), 0, s, L, ld, B, 1.0f / (float)d, scratch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(contrastive_finalize_kernel, dim3(1), dim3(64), 0, s, scratch, nblk, B, metrics);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) fill_add_kernel(float* __restrict__ dst, const float* __restrict__ add, float fill, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = add != nullptr ? dst[i] + add[i] : fill;
}

hipError_t launch_fill_add(float* dst, const float* add, float fill, int64_t n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(fill_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dst, add, fill, n);
    return hipGetLastError();
}

hipError_t launch_lap(const float* phi, const float* next_phi, int ld, float* dphi, float* dnext_phi, float* metrics,
                      float* scratch, int rows, int d, hipStream_t s) {
    const int nblk = (rows + 3) / 4;
    hipLaunchKernelGGL(lap_kernel, dim3(nblk), dim3(256), 0, s, phi, next_phi, ld, dphi, dnext_phi, scratch, rows, d);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(lap_finalize_kernel, dim3(1), dim3(64), 0, s, scratch, nblk, rows, d, metrics);
    return hipGetLastError();
}

}  // namespace fbhip
