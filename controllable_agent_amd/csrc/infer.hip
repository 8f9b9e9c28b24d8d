// Batch-1 inference kernels for gfx950: what the online loop calls on EVERY environment step
//   FBDDPGAgent.act               fb_ddpg.py:258-281   (Actor forward, fb_modules.py:107-121, + TruncatedNormal sample)
//   FBDDPGAgent.compute_z_correl  fb_ddpg.py:283-289   (BackwardMap forward, fb_modules.py:223-230, + normalised dot)
// At one row an MLP is a chain of matrix-VECTOR products, bound by weight bytes (8.8 MB for the walker actor, all
// cache-resident) and by launch count, not by MFMA: every layer is one GEMV launch in which each wavefront owns one
// output neuron (coalesced float4 reads of its weight row, the input vector in LDS), the LayerNorm+tanh of the first
// layer is recomputed by every workgroup as a prologue of the SECOND layer's launch (1024 values -- cheaper than a
// launch), and the policy head, tanh, exploration noise (Philox) and the straight-through clamp are one workgroup.
// 4 launches per act(), 4 per compute_z_correl(); the C-ABI wraps them, the host<->device copies of the few hundred
// input/output bytes and nothing else into one hipGraph (api.hip).
#include "common.h"
#include "philox.h"

namespace fbhip {

namespace {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// sum over the 256 threads of a workgroup (result to every thread); ``red`` holds 4 floats
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wsum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

constexpr int GEMV_MAXK = 2048;

// y[n] = epi( W[n, :K] . f(x) + bias[n] ),  f = identity or tanh(LayerNorm_{n_ln}(x))  (the "ntanh" of fb_modules.py:49-50)
__global__ void __launch_bounds__(256) gemv_kernel(const GemvGroup g) {
    __shared__ __attribute__((aligned(16))) float xs[GEMV_MAXK];
    __shared__ float red[4];
    int pi = 0;
#pragma unroll
    for (int i = 1; i < GEMV_MAX_GROUP; ++i)
        if (i < g.n && (int)blockIdx.x >= g.p[i].block_start) pi = i;
    const GemvProblem& p = g.p[pi];
    const int K = p.K, tid = threadIdx.x;
    // ---- prologue: the input vector (optionally LayerNorm + tanh over its first n_ln entries; the rest reads as 0)
    float xv[GEMV_MAXK / 256];
#pragma unroll
    for (int j = 0; j < GEMV_MAXK / 256; ++j) {
        const int k = tid + 256 * j;
        xv[j] = k < K ? p.x[k] : 0.f;
    }
    if (p.n_ln > 0) {
        const int n = p.n_ln;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < GEMV_MAXK / 256; ++j) s += (tid + 256 * j < n) ? xv[j] : 0.f;
        const float mean = block_sum(s, red) / (float)n;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < GEMV_MAXK / 256; ++j) {
            const float dlt = (tid + 256 * j < n) ? xv[j] - mean : 0.f;
            q += dlt * dlt;
        }
        const float rstd = rsqrtf(block_sum(q, red) / (float)n + 1e-5f);       // biased variance, eps like nn.LayerNorm
#pragma unroll
        for (int j = 0; j < GEMV_MAXK / 256; ++j) {
            const int k = tid + 256 * j;
            xv[j] = k < n ? tanhf((xv[j] - mean) * rstd * p.ln_g[k] + p.ln_b[k]) : 0.f;
        }
    }
#pragma unroll
    for (int j = 0; j < GEMV_MAXK / 256; ++j) {
        const int k = tid + 256 * j;
        if (k < K) xs[k] = xv[j];
    }
    __syncthreads();
    // ---- one wavefront per output neuron
    const int lane = tid & 63, n = ((int)blockIdx.x - p.block_start) * 4 + (tid >> 6);
    if (n >= p.N) return;
    const float4* __restrict__ w4 = reinterpret_cast<const float4*>(p.W + (size_t)n * p.ldw);
    const float4* x4 = reinterpret_cast<const float4*>(xs);
    float acc = 0.f;
    for (int k4 = lane; k4 < K / 4; k4 += 64) {
        const float4 w = w4[k4], x = x4[k4];
        acc += w.x * x.x + w.y * x.y + w.z * x.z + w.w * x.w;
    }
    acc = wsum(acc);
    if (lane == 0) {
        float v = acc + p.bias[n];
        if (p.relu) v = fmaxf(v, 0.f);
        p.y[n] = v;
    }
}

// policy head + TruncatedNormal (fb_modules.py:117-121, utils.py:164-185), one workgroup:
//   mu = tanh(W4 p + b4);  eval: action = mu;  else action = clamp(mu + stddev * eps, +-(1 - 1e-6))   (sample(clip=None))
// eps: ``noise`` when given (parity tests), else Philox keyed by the agent's seed and a device-side act counter.
__global__ void __launch_bounds__(256) act_head_kernel(const float* __restrict__ x, const float* __restrict__ W, int ldw,
                                                       const float* __restrict__ bias, int a, int K, float stddev_arg,
                                                       const float* __restrict__ stddev_dev, int eval_mode,
                                                       const float* __restrict__ noise, unsigned k0,
                                                       unsigned k1, StepState* __restrict__ st, float* __restrict__ out,
                                                       const Squash sq, float* host_out, unsigned int* dseq) {
    __shared__ float pre[64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    // (the batch-1 graph reads the exploration stddev from its staged inputs: a schedule-driven stddev then replays ONE graph
    // instead of capturing a new one per value)
    const float stddev = stddev_dev != nullptr ? *stddev_dev : stddev_arg;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const int nout = sq.on ? 2 * a : a;            // DiagGaussianActor: [loc | raw log-std]
    for (int n = wid; n < nout; n += 4) {
        const float4* w4 = reinterpret_cast<const float4*>(W + (size_t)n * ldw);
        float acc = 0.f;
        for (int k4 = lane; k4 < K / 4; k4 += 64) {
            const float4 w = w4[k4], v = x4[k4];
            acc += w.x * v.x + w.y * v.y + w.z * v.z + w.w * v.w;
        }
        acc = wsum(acc);
        if (lane == 0) pre[n] = acc + bias[n];
    }
    __syncthreads();
    const int n = threadIdx.x;
    if (n < a) {
        const float mu = tanhf(pre[n]);
        float act = mu;
        if (!eval_mode) {
            float e;
            if (noise != nullptr) {
                e = noise[n];
            } else {
                const U4 r = philox4x32_10((unsigned)(n >> 1), STREAM_ACT, st->act_count, 0u, k0, k1);
                float n0, n1;
                box_muller(r.x, r.y, n0, n1);
                e = (n & 1) ? n1 : n0;
            }
            if (sq.on) {                           // SquashedNormal.sample(): tanh(loc + exp(log_std) eps)
                const float log_std = sq.lo + 0.5f * (sq.hi - sq.lo) * (tanhf(pre[a + n]) + 1.f);
                act = tanhf(pre[n] + expf(log_std) * e);
            } else {
                const float lo = (float)(-1.0 + 1e-6), hi = (float)(1.0 - 1e-6);
                act = fminf(fmaxf(mu + e * stddev, lo), hi);
            }
        }
        out[n] = act;
        if (host_out != nullptr) host_out[n] = act;          // pinned host memory: the caller spins on the number written below
    }
    if (host_out != nullptr) __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0 && !eval_mode && noise == nullptr) st->act_count += 1u;
    if (threadIdx.x == 0 && host_out != nullptr) {
        const unsigned int sq_ = *dseq + 1u;
        *dseq = sq_;
        __hip_atomic_store(reinterpret_cast<unsigned int*>(host_out + INFER_SEQ_SLOT), sq_, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// compute_z_correl's tail (fb_ddpg.py:286-289): b = sqrt(d) y / max(|y|_2, 1e-12) (BackwardMap's own projection), then
// BOTH vectors divided by their L1 norm (the reference's ``F.normalize(z, 1)``: the positional 1 is p) and dotted.
__global__ void __launch_bounds__(64) zcorrel_kernel(const float* __restrict__ y, const float* __restrict__ z, int d,
                                                     int project, float* __restrict__ out, float* host_out, unsigned int* dseq) {
    const int lane = threadIdx.x;
    float yv[4], zv[4], s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = lane + 64 * i;
        yv[i] = j < d ? y[j] : 0.f;
        zv[i] = j < d ? z[j] : 0.f;
        s2 += yv[i] * yv[i];
    }
    const float scale = project ? sqrtf((float)d) / fmaxf(sqrtf(wsum(s2)), 1e-12f) : 1.0f;   // norm_z == 0: B is unprojected
    float l1b = 0.f, l1z = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        yv[i] *= scale;
        l1b += fabsf(yv[i]);
        l1z += fabsf(zv[i]);
    }
    const float ib = 1.0f / fmaxf(wsum(l1b), 1e-12f), iz = 1.0f / fmaxf(wsum(l1z), 1e-12f);
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) dot += (yv[i] * ib) * (zv[i] * iz);
    dot = wsum(dot);
    if (lane == 0) {
        out[0] = dot;
        if (host_out != nullptr) {
            host_out[0] = dot;
            __threadfence_system();
            const unsigned int sq_ = *dseq + 1u;
            *dseq = sq_;
            __hip_atomic_store(reinterpret_cast<unsigned int*>(host_out + INFER_SEQ_SLOT), sq_, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

}  // namespace

hipError_t launch_gemv_group(GemvGroup g, hipStream_t s) {
    int start = 0;
    for (int i = 0; i < g.n; ++i) {
        GemvProblem& p = g.p[i];
        if (p.K > GEMV_MAXK || (p.K & 3) || (p.ldw & 3) || p.n_ln > p.K) return hipErrorInvalidValue;
        p.block_start = start;
        start += (p.N + 3) / 4;
    }
    if (start <= 0) return hipSuccess;
    hipLaunchKernelGGL(gemv_kernel, dim3(start), dim3(256), 0, s, g);
    return hipGetLastError();
}

hipError_t launch_act_head(const float* x, const float* W, int ldw, const float* bias, int a, int K, float stddev,
                           int eval_mode, const float* noise, uint64_t seed, uint32_t rank, StepState* st, float* out,
                           Squash sq, hipStream_t s, const float* stddev_dev, float* host_out, unsigned int* dseq) {
    if ((sq.on ? 2 * a : a) > 64 || a > INFER_SEQ_SLOT || (K & 3) || (ldw & 3)) return hipErrorInvalidValue;
    const unsigned k0 = (unsigned)(seed & 0xffffffffu), k1 = (unsigned)(seed >> 32) ^ (0x9E3779B9u * (rank + 1u));
    hipLaunchKernelGGL(act_head_kernel, dim3(1), dim3(256), 0, s, x, W, ldw, bias, a, K, stddev, stddev_dev, eval_mode, noise, k0, k1, st,
                       out, sq, dseq != nullptr ? host_out : nullptr, dseq);
    return hipGetLastError();
}

hipError_t launch_zcorrel(const float* y, const float* z, int d, int project, float* out, hipStream_t s, float* host_out,
                          unsigned int* dseq) {
    if (d > 256) return hipErrorInvalidValue;
    hipLaunchKernelGGL(zcorrel_kernel, dim3(1), dim3(64), 0, s, y, z, d, project, out, dseq != nullptr ? host_out : nullptr, dseq);
    return hipGetLastError();
}

}  // namespace fbhip
