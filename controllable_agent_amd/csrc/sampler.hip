// Device-resident replay sampler for gfx950: the index draw, row gather, z sampling and z mixing of
//   ReplayBuffer.sample        (in_memory_replay_buffer.py:139-190)
//   EpisodeBatch.to(device)    (replay_buffer.py:50-63)  -- eliminated: the storage already lives in HBM
//   FBDDPGAgent.sample_z       (fb_ddpg.py:224-232)
//   perm / mix of update()     (fb_ddpg.py:460-485)
// Storage is episode-major float32[n_episodes, T+1, dim] exactly like ReplayBuffer._storage, so obs and next_obs
// of a transition are two ADJACENT rows (one contiguous 2*o-float read).  The gather writes straight into the
// concatenated [obs|action], [obs|z], [next_obs|z] ... input panels of the first-layer GEMMs, so torch.cat
// (fb_modules.py:114,190-191) never happens.  Random numbers: Philox4x32-10, counter = (index, stream,
// update_count), key = (seed, rank) -- reproducible and independent of launch geometry.
#include "common.h"
#include <cassert>
#include "philox.h"

namespace fbhip {

namespace {


__global__ void __launch_bounds__(256) draw_kernel(ReplayView rv, SampleOut so, int B, int d, int a, unsigned k0,
                                                   unsigned k1, const StepState* __restrict__ st, float future, int norm_z) {
    const unsigned cnt = st->update_count;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < B) {
        const U4 r = philox4x32_10((unsigned)i, STREAM_INDEX, cnt, 0u, k0, k1);
        int ep, step;
        if (rv.fixed_length) {
            // np.random.randint(0, len(self)) then randint(0, eps_len) + 1   (in_memory_replay_buffer.py:147,155)
            ep = (int)(((unsigned long long)r.x * (unsigned)rv.n_episodes) >> 32);
            const int len = rv.episode_len[ep];
            step = 1 + (int)(((unsigned long long)r.y * (unsigned)len) >> 32);
        } else {
            // np.random.choice(p = len / sum len) then a uniform step == a uniform draw over all transitions
            // (in_memory_replay_buffer.py:149-155)
            const unsigned long long total = (unsigned long long)rv.cum_len[rv.n_episodes];
            const unsigned long long r64 = ((unsigned long long)r.x << 32) | r.y;
            const unsigned long long tix = __umul64hi(r64, total);
            int lo = 0, hi = rv.n_episodes;            // find ep with cum[ep] <= tix < cum[ep+1]
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if ((unsigned long long)rv.cum_len[mid] <= tix) lo = mid; else hi = mid;
            }
            ep = lo;
            step = 1 + (int)(tix - (unsigned long long)rv.cum_len[lo]);
        }
        so.ep_idx[i] = ep;
        so.step_idx[i] = step;
        if (future >= 0.f) {
            // hindsight replay: future_idx = clip(step_idx + Geometric(p = 1 - future), 0, eps_len)
            // (in_memory_replay_buffer.py:157-161); inverse CDF G = ceil(ln u / ln(1 - p)) >= 1.  Then the selection uniform
            // of fb_ddpg.py:490.
            const U4 f = philox4x32_10((unsigned)i, STREAM_FUTURE, cnt, 0u, k0, k1);
            int geo = 1;
            if (future > 0.f) {
                const float gq = ceilf(logf(u01(f.x)) / logf(future));
                geo = gq < 1.f ? 1 : (gq > 1.0e9f ? 1000000000 : (int)gq);
            }
            const int len = rv.episode_len[ep];
            const long long fi = (long long)step + geo;
            so.future_idx[i] = (int)(fi > len ? len : fi);
            so.future_uniform[i] = u01(f.y);
        }
        so.mix_uniform[i] = u01(philox4x32_10((unsigned)i, STREAM_MIX, cnt, 0u, k0, k1).x);   // fb_ddpg.py:471
    }
    // torch.randperm (fb_ddpg.py:467) = argsort of B random keys (Philox word, ties broken by index).  Workgroup b ranks
    // keys 32b .. 32b+31: it regenerates all B keys in LDS (Philox is counter-based), 8 lanes per key each count the
    // smaller keys of one residue class, and perm[rank(i)] = i.
    if (blockIdx.x * 32 < B) {
        extern __shared__ unsigned sk[];
        const int n64 = (B + 63) & ~63;                 // padded with sentinels that never count
        for (int j = threadIdx.x; j < n64; j += 256)
            sk[j] = j < B ? philox4x32_10((unsigned)j, STREAM_PERM, cnt, 0u, k0, k1).x : 0xffffffffu;
        __syncthreads();
        const int part = threadIdx.x & 7, ii = blockIdx.x * 32 + (threadIdx.x >> 3);
        const unsigned mine = sk[ii < B ? ii : 0];
        int r = 0;
        for (int t = 0; t < n64; t += 64) {
            unsigned h[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) h[u] = sk[t + 8 * u + part];
#pragma unroll
            for (int u = 0; u < 8; ++u) r += (h[u] < mine || (h[u] == mine && t + 8 * u + part < ii)) ? 1 : 0;
        }
        r += __shfl_xor(r, 1); r += __shfl_xor(r, 2); r += __shfl_xor(r, 4);
        if (ii < B && part == 0) so.perm[r] = ii;
    }
    // gaussians: 4 per Philox call
    const int nz = B * d, na = B * a;
    for (int q4 = i; 4 * q4 < nz; q4 += gridDim.x * 256) {
        const U4 r = philox4x32_10((unsigned)q4, STREAM_Z, cnt, 0u, k0, k1);
        float n[4];
        box_muller(r.x, r.y, n[0], n[1]);
        box_muller(r.z, r.w, n[2], n[3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) if (4 * q4 + j < nz) so.z_gauss[4 * q4 + j] = n[j];
        if (!norm_z) {                          // torch.rand of sample_z's norm_z == False branch (fb_ddpg.py:230)
            const U4 u = philox4x32_10((unsigned)q4, STREAM_ZU, cnt, 0u, k0, k1);
            const float uu[4] = {u01(u.x), u01(u.y), u01(u.z), u01(u.w)};
#pragma unroll
            for (int j = 0; j < 4; ++j) if (4 * q4 + j < nz) so.z_uniform[4 * q4 + j] = uu[j];
        }
    }
    for (int q4 = i; 4 * q4 < na; q4 += gridDim.x * 256) {
        const U4 r = philox4x32_10((unsigned)q4, STREAM_EPS_NEXT, cnt, 0u, k0, k1);
        const U4 t = philox4x32_10((unsigned)q4, STREAM_EPS_ACTOR, cnt, 0u, k0, k1);
        float n[4], m[4];
        box_muller(r.x, r.y, n[0], n[1]); box_muller(r.z, r.w, n[2], n[3]);
        box_muller(t.x, t.y, m[0], m[1]); box_muller(t.z, t.w, m[2], m[3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) if (4 * q4 + j < na) { so.eps_next[4 * q4 + j] = n[j]; so.eps_actor[4 * q4 + j] = m[j]; }
    }
}

__device__ __forceinline__ void copy_row(float* __restrict__ dst, const float* __restrict__ src, int n, int lane) {
    for (int j = lane; j < n; j += 64) dst[j] = src[j];
}

// one wavefront per sampled transition
__global__ void __launch_bounds__(256) gather_kernel(const GatherArgs g) {
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= g.B) return;
    const int e = g.ep_idx[i], s = g.step_idx[i];
#ifdef FBHIP_DEBUG       // make debug: every gathered row must lie inside its episode of the bound storage
    assert(e >= 0 && e < g.rv.n_episodes && s >= 1 && s < g.rv.t1 && s <= g.rv.episode_len[e]);
    // (the hindsight row is storage[ep, future_idx - 1]; the external-batch path keeps it as a third row behind a 2-row episode)
    assert(g.future_idx == nullptr || (g.future_idx[i] >= 1 && g.future_idx[i] <= g.rv.t1));
    assert(g.perm == nullptr || (g.perm[i] >= 0 && g.perm[i] < g.B));
#endif
    const size_t t = (size_t)e * g.rv.t1 + s;                         // row of the "next" step
    const float* obs = g.rv.observation + (t - 1) * g.o;              // observation[ep, step-1]
    const float* nobs = obs + g.o;                                    // observation[ep, step] (adjacent row)
    const float* act = g.rv.action + t * g.a;                         // action[ep, step]
    for (int j = lane; j < g.o; j += 64) {
        const float ov = obs[j], nv = nobs[j];
        g.Xoa[(size_t)i * g.ld_oa + j] = ov;
        g.Xoz[(size_t)i * g.ld_oz + j] = ov;
        g.Xopi[(size_t)i * g.ld_opi + j] = ov;
        g.Xo[(size_t)i * g.ld_o + j] = ov;
        g.Xnoz[(size_t)i * g.ld_noz + j] = nv;
        g.Xnoa[(size_t)i * g.ld_noa + j] = nv;
    }
    if (g.act_idx != nullptr) { if (lane == 0) g.act_idx[i] = act[0]; }
    else copy_row(g.Xoa + (size_t)i * g.ld_oa + g.aoff, act, g.a, lane);
    if (lane == 0) g.disc[i] = g.gamma * g.rv.discount[t];            // discount * storage['discount'] (:171)
    copy_row(g.next_goal + (size_t)i * g.ld_ng, g.use_goal ? g.rv.goal + t * g.g : nobs, g.g, lane);
    // backward_input[perm] (fb_ddpg.py:460-468): row i of the permuted panel is transition perm[i]
    const int pi = g.perm != nullptr ? g.perm[i] : i;
    const size_t tp = (size_t)g.ep_idx[pi] * g.rv.t1 + g.step_idx[pi] - 1;
    const float* bsrc = g.use_goal ? g.rv.goal + tp * g.g : g.rv.observation + tp * g.o;
    copy_row(g.bin + (size_t)i * g.ld_bin, bsrc, g.g, lane);
    if (g.pgoal != nullptr) {        // desired_goal = next_goal[perm] (sf.py:726-727)
        const int qi = g.pperm[i];
#ifdef FBHIP_DEBUG
        assert(qi >= 0 && qi < g.B);
#endif
        const size_t tq = (size_t)g.ep_idx[qi] * g.rv.t1 + g.step_idx[qi];
        copy_row(g.pgoal + (size_t)i * g.ld_pg, g.use_goal ? g.rv.goal + tq * g.g : g.rv.observation + tq * g.o, g.g, lane);
    }
    if (g.future_idx != nullptr) {   // future_goal / future_obs = storage[ep, future_idx - 1] (in_memory_replay_buffer.py:176-183)
        const size_t tf = (size_t)e * g.rv.t1 + g.future_idx[i] - 1;
        copy_row(g.fgoal + (size_t)i * g.ld_fg, g.use_goal ? g.rv.goal + tf * g.g : g.rv.observation + tf * g.o, g.g, lane);
    }
}

// z = mix ? sqrt(d) normalize(B(backward_input)) : sqrt(d) normalize(gauss), scattered into every panel that carries z.
//   sample_z (fb_ddpg.py:224-228) and the z-mix (fb_ddpg.py:470-485) in one pass.  ``ymix`` is the RAW output of the
//   BackwardMap mlp: the reference normalises it twice (inside BackwardMap.forward, fb_modules.py:229, and again at
//   fb_ddpg.py:483-484) and so does this kernel, with the same operation order.  Thread 0 also advances the RNG counter
//   for the next update (every draw of this update has been consumed by now: stream order).
__global__ void __launch_bounds__(256) mix_z_kernel(const float* __restrict__ gauss, int ldg,
                                                    const float* __restrict__ ymix, int ldy,
                                                    const float* __restrict__ mixu, float mix_ratio,
                                                    float* __restrict__ z, int ldz, float* __restrict__ Xoz, int ld_oz,
                                                    float* __restrict__ Xnoz, int ld_noz, int o, int B, int d,
                                                    StepState* __restrict__ st, const float* __restrict__ yfut,
                                                    const float* __restrict__ futu, float future_ratio,
                                                    const float* __restrict__ zunif, int mix_proj,
                                                    const ZPanels zx) {
    if (st != nullptr && blockIdx.x == 0 && threadIdx.x == 0) st->update_count += 1u;
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= B) return;
    // hindsight rows (fb_ddpg.py:487-491) override the mix: z = B(future_goal), projected ONCE (BackwardMap's own)
    const bool fut = (future_ratio > 0.f) && (futu[i] < future_ratio);
    const bool mix = !fut && (mix_ratio > 0.f) && (mixu[i] < mix_ratio);
    const float* src = fut ? yfut + (size_t)i * ldy : (mix ? ymix + (size_t)i * ldy : gauss + (size_t)i * ldg);
    constexpr int ME = 4;                         // d <= 256
    float v[ME];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < ME; ++k) {
        const int j = lane + 64 * k;
        v[k] = j < d ? src[j] : 0.f;
        s += v[k] * v[k];
    }
    const float sc = sqrtf((float)d);
    // mix_proj < 0 (IdentityMap backward nets, cfg.debug): hindsight rows are the RAW goal -- no BackwardMap projection exists
    const bool fut_raw = mix_proj < 0;
    if (mix_proj < 0) mix_proj = -mix_proj;
    if (fut && fut_raw) {
    } else if (zunif == nullptr) {
        // norm_z: gaussian and hindsight rows are projected once, mixed rows twice (BackwardMap's own + fb_ddpg.py:483-484);
        // with rand_weight the mixed row is a weighted sum of already projected rows and gets the :483 projection only
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        const float den = fmaxf(sqrtf(s), 1e-12f);
        float s2 = 0.f;
#pragma unroll
        for (int k = 0; k < ME; ++k) {
            v[k] = sc * (v[k] / den);
            s2 += v[k] * v[k];
        }
        if (mix && mix_proj > 1) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) s2 += __shfl_xor(s2, off);
            const float den2 = fmaxf(sqrtf(s2), 1e-12f);
#pragma unroll
            for (int k = 0; k < ME; ++k) v[k] = sc * (v[k] / den2);
        }
    } else if (!fut && !mix) {
        // norm_z == False: z = sqrt(d) * U * g/|g| (fb_ddpg.py:229-231); mixed / hindsight rows are the RAW BackwardMap output
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        const float den = fmaxf(sqrtf(s), 1e-12f);
#pragma unroll
        for (int k = 0; k < ME; ++k) {
            const int j = lane + 64 * k;
            v[k] = j < d ? sc * zunif[(size_t)i * d + j] * (v[k] / den) : 0.f;
        }
    }
#pragma unroll
    for (int k = 0; k < ME; ++k) {
        const int j = lane + 64 * k;
        if (j < d) {
            z[(size_t)i * ldz + j] = v[k];
            Xoz[(size_t)i * ld_oz + o + j] = v[k];
            Xnoz[(size_t)i * ld_noz + o + j] = v[k];
#pragma unroll
            for (int q = 0; q < 3; ++q)
                if (zx.p[q] != nullptr) zx.p[q][(size_t)i * zx.ld[q] + o + j] = v[k];
        }
    }
}

// rand_weight (fb_ddpg.py:477-480): one workgroup per row of the B x B weight matrix
__global__ void __launch_bounds__(256) rand_weight_kernel(float* __restrict__ W, float* __restrict__ u, int B, int generate,
                                                         unsigned k0, unsigned k1, const StepState* __restrict__ st) {
    __shared__ float red[4];
    const int i = blockIdx.x, tid = threadIdx.x;
    float* row = W + (size_t)i * B;
    const unsigned cnt = st->update_count;
    float ss = 0.f;
    for (int q4 = tid; 4 * q4 < B; q4 += 256) {
        float v[4];
        if (generate) {
            const U4 r = philox4x32_10((unsigned)(i * ((B + 3) / 4) + q4), STREAM_RW, cnt, 0u, k0, k1);
            v[0] = u01(r.x); v[1] = u01(r.y); v[2] = u01(r.z); v[3] = u01(r.w);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = 4 * q4 + j < B ? row[4 * q4 + j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * q4 + j < B) { ss += v[j] * v[j]; if (generate) row[4 * q4 + j] = v[j]; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    const float nrm = sqrtf(red[0] + red[1] + red[2] + red[3]);
    float ui;
    if (generate) {
        ui = u01(philox4x32_10((unsigned)i, STREAM_RWU, cnt, 0u, k0, k1).x);
        if (tid == 0) u[i] = ui;
    } else {
        ui = u[i];
    }
    const float den = fmaxf(nrm, 1e-12f);                       // F.normalize eps
    for (int j = tid; j < B; j += 256) row[j] = ui * (row[j] / den);
}

}  // namespace

hipError_t launch_rand_weight(float* W, float* u, int B, int generate, uint64_t seed, uint32_t rank, const StepState* st,
                              hipStream_t s) {
    const unsigned k0 = (unsigned)(seed & 0xffffffffu), k1 = (unsigned)(seed >> 32) ^ (0x9E3779B9u * (rank + 1u));
    hipLaunchKernelGGL(rand_weight_kernel, dim3(B), dim3(256), 0, s, W, u, B, generate, k0, k1, st);
    return hipGetLastError();
}

hipError_t launch_draw(const ReplayView& rv, const SampleOut& so, int B, int d, int a, uint64_t seed, uint32_t rank,
                       const StepState* st, float future, int norm_z, hipStream_t s) {
    if (B > 8192) return hipErrorInvalidValue;
    const unsigned k0 = (unsigned)(seed & 0xffffffffu), k1 = (unsigned)(seed >> 32) ^ (0x9E3779B9u * (rank + 1u));
    int blocks = (B * d / 4 + 255) / 256;
    if (blocks < (B + 31) / 32) blocks = (B + 31) / 32;
    hipLaunchKernelGGL(draw_kernel, dim3(blocks), dim3(256), (size_t)((B + 63) & ~63) * 4, s, rv, so, B, d, a, k0, k1, st, future, norm_z);
    return hipGetLastError();
}

hipError_t launch_gather(const GatherArgs& ga, hipStream_t s) {
    hipLaunchKernelGGL(gather_kernel, dim3((ga.B + 3) / 4), dim3(256), 0, s, ga);
    return hipGetLastError();
}

hipError_t launch_mix_z(const float* gauss, int ldg, const float* ymix, int ldy, const float* mix_uniform, float mix_ratio,
                        float* z, int ldz, float* Xoz, int ld_oz, float* Xnoz, int ld_noz, int o, int B, int d,
                        StepState* st, const float* yfut, const float* future_uniform, float future_ratio,
                        const float* z_uniform, int mix_projections, ZPanels extra, hipStream_t s) {
    if (d > 256) return hipErrorInvalidValue;
    if (future_ratio > 0.f && (!yfut || !future_uniform)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mix_z_kernel, dim3((B + 3) / 4), dim3(256), 0, s, gauss, ldg, ymix, ldy, mix_uniform, mix_ratio, z,
                       ldz, Xoz, ld_oz, Xnoz, ld_noz, o, B, d, st, yfut, future_uniform, future_ratio, z_uniform, mix_projections, extra);
    return hipGetLastError();
}

}  // namespace fbhip
