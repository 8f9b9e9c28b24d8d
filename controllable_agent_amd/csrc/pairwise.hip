// Pairwise forward-backward loss for gfx950: fb_ddpg.py:313-348 (target_M, M1, M2, off-diagonal contrastive
// FB loss, diagonal term, orthonormality loss) AND its gradient wrt F1, F2, B (fb_loss.backward(), :383) in one
// pass that never materialises a [B,B] matrix.  Closed-form gradients: SURVEY.md appendix C.
//
// The reference builds five [B,B] matrices and indexes them with a boolean off-diagonal mask
// (aten::nonzero + index_put_ backward = ~50 % of its CPU step).  Here the batch x batch grid is cut into
// 32x32 tiles; each tile is a handful of v_mfma_f32_32x32x2_f32 products over K = z_dim, the mask is the
// arithmetic predicate (s != t), and the tile's gradient contribution is contracted back immediately.
//
// Layout trick: a 32x32 MFMA accumulator holds tile T[r][c] with lane = column c and registers = rows r, which
// is exactly the A-operand layout of a product T^T . X contracting over the tile's ROWS -- so "sum over rows"
// needs no shuffle or LDS transpose.  Every workgroup therefore fixes a row-block I (32 batch rows) as the
// tile COLUMNS and walks blocks J as tile ROWS; everything it accumulates is indexed by I and stays in
// registers across the J loop:
//   role 1 (waves 0,1; I plays s, J plays t):  T = B[J] . F_i[I]^T  = M_i[s,t]^T  -> dF_i[I] += G_i^T-contract . B[J]
//   role 2 (waves 2,3; I plays t, J plays s):  P = F_i[J] . B[I]^T  = M_i[s,t]    -> dB[I]  += G_i  -contract . F_i[J]
//   cov    (waves 2/3 alternate):              C = B[J] . B[I]^T (symmetric)      -> dB[I]  += 2 c_o H-contract . B[J]
// I-side operands live in registers as MFMA B-fragments for the whole kernel; J-side tiles are staged in LDS and
// shared by the four waves.  Partial results over J-chunks go to a scratch buffer and are folded in a fixed
// order by pairwise_reduce_kernel (deterministic; no atomics).  Scalar sums are wave-shuffle reduced.
#include "common.h"
#include "fbhip.h"

namespace fbhip {

namespace {

constexpr int PW_SLOTS = 4;        // dF1, dF2, dB (wave 2), dB (wave 3)
constexpr int PW_SCAL = 12;        // scalar partials per workgroup

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

struct PwArgs {
    const float* F1; const float* F2; const float* Bm; const float* tF1; const float* tF2; const float* tB;
    const float* discount;
    int B, d, ld;
    float ortho2;              // 2 * ortho_coef
    int jpc;                   // J tiles per chunk
    int njt;                   // number of J tiles (ceil(B/32))
    int vec;                   // 16-byte aligned panels with ld % 4 == 0 -> float4 loads
    float* partial;            // [nchunks][PW_SLOTS][Bp][DP]
    float* scal;               // [nblocks][PW_SCAL]
    int Bp;
    int i_off;                 // block mode (global-batch data parallel): the workgroups' I rows are [i_off, i_off + 32 gridDim.x)
                               // of the B-row panels; outputs / partials are indexed by the LOCAL row (I - i_off)
};

// Stage the same 32-row block of SIX matrices (zero filled outside [0,B) x [0,d)) into LDS [32][LD] each.  load() only
// ISSUES the global loads (branch-free on the aligned path: out-of-range quads read a clamped, valid address), the
// zero-fill masks are applied in store() -- so the loads of J tile t+1 stay in flight under the MFMAs of tile t and the
// s_waitcnt lands right before the LDS writes of the next iteration, not behind every load.
template <int LD, bool VEC>
struct Stage6 {
    static constexpr int W = LD - 1;                 // multiple of 32
    static constexpr int QPT = 32 * (W / 4) / 256;   // float4 quads per thread per matrix (2 for W=64, 4 for W=128)
    float4 v[6][QPT > 0 ? QPT : 1];
    int row0_, B_, d_, ld_;

    __device__ __forceinline__ void load(const float* const (&X)[6], int ld, int row0, int B, int d, int tid) {
        row0_ = row0; B_ = B; d_ = d; ld_ = ld;
        if constexpr (VEC) {
#pragma unroll
            for (int i = 0; i < QPT; ++i) {
                const int q = tid + i * 256;
                const int r = q / (W / 4), n0 = 4 * (q % (W / 4));
                const bool ok = (row0 + r < B) && (n0 < ld);                 // ld % 4 == 0: the quad stays inside its row
                const size_t off = ok ? (size_t)(row0 + r) * ld + n0 : 0;
#pragma unroll
                for (int m = 0; m < 6; ++m) v[m][i] = *reinterpret_cast<const float4*>(X[m] + off);
            }
        } else {
#pragma unroll
            for (int m = 0; m < 6; ++m)
#pragma unroll
                for (int i = 0; i < QPT; ++i) {
                    const int q = tid + i * 256;
                    const int r = q / (W / 4), n0 = 4 * (q % (W / 4));
                    const int gr = row0 + r;
                    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (gr < B && n0 < d) {
                        const float* ptr = X[m] + (size_t)gr * ld + n0;
                        x.x = ptr[0];
                        if (n0 + 1 < d) x.y = ptr[1];
                        if (n0 + 2 < d) x.z = ptr[2];
                        if (n0 + 3 < d) x.w = ptr[3];
                    }
                    v[m][i] = x;
                }
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ lds_base, int tid) {
        // pin the first use of the loaded registers HERE: without it hipcc hoists the zero-fill selects up to the loads
        // (before the MFMAs of the previous tile) and waits for every load right where it was issued
#pragma unroll
        for (int m = 0; m < 6; ++m)
#pragma unroll
            for (int i = 0; i < QPT; ++i)
                asm volatile("" : "+v"(v[m][i].x), "+v"(v[m][i].y), "+v"(v[m][i].z), "+v"(v[m][i].w));
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const int q = tid + i * 256;
            const int r = q / (W / 4), n0 = 4 * (q % (W / 4));
            const bool row_ok = row0_ + r < B_;
            const bool k0 = row_ok && n0 < d_, k1 = row_ok && n0 + 1 < d_, k2 = row_ok && n0 + 2 < d_, k3 = row_ok && n0 + 3 < d_;
#pragma unroll
            for (int m = 0; m < 6; ++m) {
                float* dst = lds_base + m * 32 * LD + r * LD + n0;
                dst[0] = k0 ? v[m][i].x : 0.f; dst[1] = k1 ? v[m][i].y : 0.f;
                dst[2] = k2 ? v[m][i].z : 0.f; dst[3] = k3 ? v[m][i].w : 0.f;
            }
        }
    }
};

// acc(32x32) = rowsJ (LDS, [32][LD]) . fragI^T over KS k-steps
template <int KS, int LD>
__device__ __forceinline__ floatx16 tile_mm(const float* __restrict__ sJ, const float (&fragI)[KS], int l31, int h) {
    floatx16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const float* a = sJ + l31 * LD + h;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * ks], fragI[ks], acc, 0, 0, 0);
    return acc;
}

// out[nt] += G^T-contract . X[J]:  out[c][n] += sum_r G[r][c] * X[r][n]
template <int NT, int LD>
__device__ __forceinline__ void contract_rows(floatx16 (&out)[NT], const floatx16& G, const float* __restrict__ sX,
                                              int l31, int h) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const float* b = sX + acc_row(reg, h) * LD + l31;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) out[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(G[reg], b[32 * nt], out[nt], 0, 0, 0);
    }
}

template <int KS, int LD>
__device__ __forceinline__ void load_frag(float (&f)[KS], const float* __restrict__ sI, int l31, int h) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) f[ks] = sI[l31 * LD + 2 * ks + h];
}

// VEC: 16-byte aligned panels with ld % 4 == 0 (the product's workspace).  A separate instantiation, not a runtime
// branch: merging the two loaders' registers makes hipcc wait for the loads right after issuing them.
// NG wave groups of four waves each walk ALTERNATE J tiles of the workgroup's chunk concurrently (own LDS panels, own
// accumulators; group g > 0 hands its sums to group 0 through LDS at the end, in group order: deterministic).  With one group a
// SIMD holds a single wave whose every LDS round trip, barrier and staging phase leaves the matrix pipe idle (35 us for 14.5 us
// of MFMA issue at B = 1024, d = 50); two groups put two independent chains on every SIMD.  NG = 2 needs 2 x 6 panels in LDS:
// d <= 64.
template <int KS, bool VEC, int NG>
__global__ void __launch_bounds__(256 * NG) pairwise_kernel(const PwArgs p) {
    constexpr int NT = (2 * KS + 31) / 32;
    constexpr int W = (2 * KS > 32 * NT ? 2 * KS : 32 * NT);
    constexpr int LD = W + 1;                       // odd => conflict-free for both access patterns
    constexpr int GRP = 6 * 32 * LD + 32;           // floats per group: 6 x [32][LD] + gamma[32]
    extern __shared__ float lds_all[];              // NG x GRP
    const int grp = NG > 1 ? (int)(threadIdx.x >> 8) : 0;
    float* lds = lds_all + grp * GRP;
    float* sBm = lds;
    float* stB = sBm + 32 * LD;
    float* sF1 = stB + 32 * LD;
    float* sF2 = sF1 + 32 * LD;
    float* stF1 = sF2 + 32 * LD;
    float* stF2 = stF1 + 32 * LD;
    float* sGam = stF2 + 32 * LD;

    const int tid = threadIdx.x & 255, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int I0 = blockIdx.x * 32 + p.i_off, chunk = blockIdx.y;
    const int B = p.B, d = p.d;
    const float n_off = (float)B * (float)(B - 1);
    const float inv_noff = 1.0f / n_off, inv_b = 1.0f / (float)B;

    // LDS order of the six panels: Bm, tB, F1, F2, tF1, tF2
    const float* const srcs[6] = {p.Bm, p.tB, p.F1, p.F2, p.tF1, p.tF2};
    Stage6<LD, VEC> stg;

    // ---- I-side fragments (registers, whole kernel) ----------------------------------------------------
    stg.load(srcs, p.ld, I0, B, d, tid);
    stg.store(lds, tid);
    __syncthreads();
    float fa[KS], fb[KS], fc[KS];
    if (wid < 2) {                   // role 1: F_i[I], tF1[I], tF2[I]
        load_frag<KS, LD>(fa, wid == 0 ? sF1 : sF2, l31, h);
        load_frag<KS, LD>(fb, stF1, l31, h);
        load_frag<KS, LD>(fc, stF2, l31, h);
    } else {                         // role 2: B[I], tB[I]
        load_frag<KS, LD>(fa, sBm, l31, h);
        load_frag<KS, LD>(fb, stB, l31, h);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) fc[ks] = 0.f;
    }
    const int gcol = I0 + l31;                              // batch index of this lane's tile column
    const float gam_col = (gcol < B) ? p.discount[gcol] : 0.f;
    __syncthreads();

    floatx16 out[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) out[nt][i] = 0.f;
    float s_sq = 0.f, s_diag = 0.f, s_all = 0.f, s_tall = 0.f, s_csq = 0.f, s_cdiag = 0.f;

    const int jt_begin = chunk * p.jpc;
    const int jt_end = min(jt_begin + p.jpc, p.njt);
    // group g takes the tiles jt_begin + g, jt_begin + g + NG, ...; every group runs the same number of iterations (the
    // barriers are workgroup-wide), a group without a tile in the last one only keeps them company
#pragma unroll 1
    for (int jt0 = jt_begin; jt0 < jt_end; jt0 += NG) {
        const int jt = jt0 + grp;
        const bool live = jt < jt_end;
        const int J0 = min(jt, p.njt - 1) * 32;
        if (jt0 == jt_begin) stg.load(srcs, p.ld, J0, B, d, tid);          // later tiles were prefetched below
        stg.store(lds, tid);
        if (tid < 32) sGam[tid] = (J0 + tid < B) ? p.discount[J0 + tid] : 0.f;
        __syncthreads();
        if (jt0 + NG < jt_end) stg.load(srcs, p.ld, min(jt + NG, p.njt - 1) * 32, B, d, tid);      // next J tile in flight under the MFMAs
        if (live) {

        if (wid < 2) {
            // tile rows = t (J), cols = s (I):  T[r][c] = M_i[s = I0+c][t = J0+r]
            floatx16 tm = tile_mm<KS, LD>(stB, fb, l31, h);
            {
                const floatx16 t2 = tile_mm<KS, LD>(stB, fc, l31, h);
#pragma unroll
                for (int i = 0; i < 16; ++i) tm[i] = fminf(tm[i], t2[i]);
            }
            floatx16 G = tile_mm<KS, LD>(sBm, fa, l31, h);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int grow = J0 + acc_row(reg, h);
                const float m = G[reg], t = tm[reg];
                const float dlt = m - gam_col * t;                 // discount is indexed by s = column here
                const bool diag = (grow == gcol);
                s_all += m;
                s_tall += t;
                if (diag) {
                    s_diag += m;
                    G[reg] = (gcol < B) ? -inv_b : 0.f;
                } else {
                    s_sq += dlt * dlt;
                    G[reg] = dlt * inv_noff;
                }
            }
            contract_rows<NT, LD>(out, G, sBm, l31, h);            // dF_i[I] += G^T-contract . B[J]
        } else {
            // tile rows = s (J), cols = t (I):  P[r][c] = M_i[s = J0+r][t = I0+c]
            floatx16 tm = tile_mm<KS, LD>(stF1, fb, l31, h);
            {
                const floatx16 t2 = tile_mm<KS, LD>(stF2, fb, l31, h);
#pragma unroll
                for (int i = 0; i < 16; ++i) tm[i] = fminf(tm[i], t2[i]);
            }
            const float* sFi = (wid == 2) ? sF1 : sF2;
            floatx16 G = tile_mm<KS, LD>(sFi, fa, l31, h);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int r = acc_row(reg, h);
                const int grow = J0 + r;
                const float dlt = G[reg] - sGam[r] * tm[reg];      // discount is indexed by s = row here
                const bool diag = (grow == gcol);
                G[reg] = diag ? ((gcol < B) ? -inv_b : 0.f) : dlt * inv_noff;
            }
            contract_rows<NT, LD>(out, G, sFi, l31, h);            // dB[I] += G-contract . F_i[J]
            if ((jt & 1) == (wid & 1)) {
                // covariance tile C[r][c] = B[J0+r] . B[I0+c]  (fb_ddpg.py:344-348)
                floatx16 C = tile_mm<KS, LD>(sBm, fa, l31, h);
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int grow = J0 + acc_row(reg, h);
                    const float c = C[reg];
                    const bool diag = (grow == gcol);
                    if (diag) {
                        s_cdiag += c;
                        C[reg] = (gcol < B) ? -p.ortho2 * inv_b : 0.f;
                    } else {
                        s_csq += c * c;
                        C[reg] = p.ortho2 * c * inv_noff;
                    }
                }
                // L_orth = mean_offdiag C^2 - 2 mean_diag C;  dL/dC = Hm = 2C/N_off (off-diag), -2/B (diag);
                // C = B B^T  =>  dB = ortho * (Hm + Hm^T) B = ortho * 2 * Hm . B.  With ortho2 = 2*ortho the tile
                // coefficient is ortho2 * Hm = 2 * (ortho2*C/N_off) resp. 2 * (-ortho2/B):
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) C[reg] *= 2.0f;
                contract_rows<NT, LD>(out, C, sBm, l31, h);
            }
        }
        }
        __syncthreads();
    }

    if constexpr (NG > 1) {
        // groups 1 .. NG-1 hand their accumulators and scalar sums to group 0 (their own panels are free now), added in group order
        constexpr int PER = (16 * NT + 6) * 256;     // floats a group parks: [16 NT + 6][256 threads]
        static_assert(PER <= GRP, "the hand-off must fit a group's panels");
        if (grp > 0) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) lds[(nt * 16 + reg) * 256 + tid] = out[nt][reg];
            float* sc = lds + 16 * NT * 256;
            sc[0 * 256 + tid] = s_sq; sc[1 * 256 + tid] = s_diag; sc[2 * 256 + tid] = s_all;
            sc[3 * 256 + tid] = s_tall; sc[4 * 256 + tid] = s_csq; sc[5 * 256 + tid] = s_cdiag;
        }
        __syncthreads();
        if (grp > 0) return;
#pragma unroll
        for (int g2 = 1; g2 < NG; ++g2) {
            const float* src = lds_all + g2 * GRP;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) out[nt][reg] += src[(nt * 16 + reg) * 256 + tid];
            const float* sc = src + 16 * NT * 256;
            s_sq += sc[0 * 256 + tid]; s_diag += sc[1 * 256 + tid]; s_all += sc[2 * 256 + tid];
            s_tall += sc[3 * 256 + tid]; s_csq += sc[4 * 256 + tid]; s_cdiag += sc[5 * 256 + tid];
        }
    }

    // ---- write partial outputs ---------------------------------------------------------------------------
    constexpr int DP = 32 * NT;
    float* dst = p.partial + (((size_t)chunk * PW_SLOTS + wid) * p.Bp + (I0 - p.i_off)) * DP;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) dst[(size_t)acc_row(reg, h) * DP + nt * 32 + l31] = out[nt][reg];

    float* sc = p.scal + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * PW_SCAL;
    s_sq = wsum(s_sq); s_diag = wsum(s_diag); s_all = wsum(s_all); s_tall = wsum(s_tall);
    s_csq = wsum(s_csq); s_cdiag = wsum(s_cdiag);
    if (lane == 0) {
        if (wid == 0) { sc[0] = s_sq; sc[1] = s_diag; sc[2] = s_all; sc[3] = s_tall; }
        if (wid == 1) { sc[4] = s_sq; sc[5] = s_diag; }
        if (wid == 2) { sc[6] = s_csq; sc[7] = s_cdiag; }
        if (wid == 3) { sc[8] = s_csq; sc[9] = s_cdiag; }
    }
}

__global__ void __launch_bounds__(256) pairwise_reduce_kernel(const float* __restrict__ partial,
                                                              const float* __restrict__ scal, int nchunks,
                                                              int nblocks, int B, int Bp, int d, int DP, int ld,
                                                              int Bglobal /* normalisers of the metrics */,
                                                              float ortho_coef, float* __restrict__ dF1,
                                                              float* __restrict__ dF2, float* __restrict__ dB,
                                                              float* __restrict__ metrics, StepState* adv, int adv_which,
                                                              const float* __restrict__ y, const float* __restrict__ norms,
                                                              float* __restrict__ dy, float out_scale) {
    if (adv != nullptr && blockIdx.x == 0 && threadIdx.x == 255) step_advance_device(adv, adv_which);
    // one wavefront per row (d <= 128: two elements per lane); with ``y`` the backward of B = sqrt(d) y / |y|
    // (dy = (sqrt(d)/|y|)(dB - yhat (yhat . dB)), exactly l2norm_bwd_kernel) follows in the same registers
    {
        const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
        if (r < B) {
            float cc[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int n = lane + 64 * i;
                float a = 0.f, b = 0.f, c = 0.f;
                if (n < d) {
                    for (int ch = 0; ch < nchunks; ++ch) {
                        const float* base = partial + ((size_t)ch * PW_SLOTS * Bp + r) * DP + n;
                        a += base[0];
                        b += base[(size_t)Bp * DP];
                        c += base[(size_t)2 * Bp * DP] + base[(size_t)3 * Bp * DP];
                    }
                    a *= out_scale; b *= out_scale; c *= out_scale;
                    dF1[(size_t)r * ld + n] = a;
                    dF2[(size_t)r * ld + n] = b;
                    dB[(size_t)r * ld + n] = c;
                }
                cc[i] = c;
            }
            if (y != nullptr) {
                const float inv = 1.0f / fmaxf(norms[r], 1e-12f);
                float yh[2], dot = 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int n = lane + 64 * i;
                    yh[i] = n < d ? y[(size_t)r * ld + n] * inv : 0.f;
                    dot += yh[i] * cc[i];
                }
                dot = wsum(dot);
                const float scale = sqrtf((float)d);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int n = lane + 64 * i;
                    if (n < d) dy[(size_t)r * ld + n] = scale * inv * (cc[i] - yh[i] * dot);
                }
            }
        }
    }
    if (blockIdx.x == 0) {
        // workgroup 0: wave w folds scalar slots w, w+4, w+8 over all workgroups -- each lane a strided slice in
        // fp64, then a fixed xor-shuffle tree (deterministic)
        __shared__ double tot[12];
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
        for (int slot = wid; slot < 10; slot += 4) {
            double s = 0.0;
            for (int bk = lane; bk < nblocks; bk += 64) s += (double)scal[(size_t)bk * PW_SCAL + slot];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if (lane == 0) tot[slot] = s;
        }
        __syncthreads();
        if (threadIdx.x == 0 && metrics != nullptr) {
            const double noff = (double)Bglobal * (double)(Bglobal - 1), bb = (double)Bglobal;
            const double fb_off = 0.5 * (tot[0] + tot[4]) / noff;
            const double fb_diag = -(tot[1] + tot[5]) / bb;
            const double orth_off = (tot[6] + tot[8]) / noff;
            const double orth_diag = -2.0 * (tot[7] + tot[9]) / bb;
            metrics[FBHIP_M_FB_OFFDIAG] = (float)fb_off;
            metrics[FBHIP_M_FB_DIAG] = (float)fb_diag;
            metrics[FBHIP_M_ORTH_LOSS_OFFDIAG] = (float)orth_off;
            metrics[FBHIP_M_ORTH_LOSS_DIAG] = (float)orth_diag;
            metrics[FBHIP_M_ORTH_LOSS] = (float)(orth_off + orth_diag);
            metrics[FBHIP_M_FB_LOSS] = (float)(fb_off + fb_diag + (double)ortho_coef * (orth_off + orth_diag));
            metrics[FBHIP_M_M1] = (float)(tot[2] / (bb * bb));
            metrics[FBHIP_M_TARGET_M] = (float)(tot[3] / (bb * bb));
        }
    }
}

struct PwPlan { int ks, nt, ld, dp, njt, nchunks, jpc, Bp, nI, ng; size_t lds_bytes; };

// B: rows of the panels (the J extent); rows: the I extent this launch owns (== B for the square single-device loss)
PwPlan make_plan(int B, int d, int rows = -1) {
    if (rows < 0) rows = B;
    static const int opts[] = {4, 8, 16, 25, 32, 50, 64};
    PwPlan pl{};
    pl.ks = -1;
    for (int o : opts) if (2 * o >= d) { pl.ks = o; break; }
    if (pl.ks < 0) return pl;
    pl.nt = (2 * pl.ks + 31) / 32;
    const int w = (2 * pl.ks > 32 * pl.nt) ? 2 * pl.ks : 32 * pl.nt;
    pl.ld = w + 1;
    pl.dp = 32 * pl.nt;
    pl.njt = (B + 31) / 32;
    pl.nI = (rows + 31) / 32;
    pl.Bp = pl.nI * 32;
    // aim for >= 256 workgroups (one per CU): nI * nchunks
    int nchunks = (256 + pl.nI - 1) / pl.nI;
    if (nchunks > pl.njt) nchunks = pl.njt;
    if (nchunks < 1) nchunks = 1;
    pl.jpc = (pl.njt + nchunks - 1) / nchunks;
    pl.nchunks = (pl.njt + pl.jpc - 1) / pl.jpc;
    // two wave groups per workgroup when their panels fit the CU's LDS next to nothing else and a chunk has tiles for both
    pl.ng = (pl.ld <= 65 && pl.jpc >= 2) ? 2 : 1;
    pl.lds_bytes = (size_t)pl.ng * (6 * 32 * pl.ld + 32) * sizeof(float);
    return pl;
}

}  // namespace

// upper bound over every world size of the block mode: nchunks <= ceil(256 / nI) whatever the global row count is
size_t pairwise_scratch_floats(int B, int d) {
    const PwPlan pl = make_plan(B, d);
    if (pl.ks < 0) return 0;
    const size_t nchunks = (size_t)(256 + pl.nI - 1) / pl.nI;
    return nchunks * PW_SLOTS * pl.Bp * pl.dp + nchunks * pl.nI * PW_SCAL;
}

hipError_t pairwise_prepare(int B, int d) {
    const PwPlan pl = make_plan(B, d);
    if (pl.ks < 0) return hipErrorInvalidValue;
    // (both group counts: the block mode of the global-batch schedule plans its own chunking)
    const int bytes1 = (int)((size_t)(6 * 32 * pl.ld + 32) * sizeof(float)), bytes2 = 2 * bytes1;
#define PW_ATTR1(KS, VEC, NG_, BYTES)                                                                                  \
    if ((BYTES) > 48 * 1024) {                                                                                        \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pairwise_kernel<KS, VEC, NG_>),              \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (BYTES));                      \
        if (e != hipSuccess) return e;                                                                                \
    }
#define PW_ATTR(KS) { PW_ATTR1(KS, true, 1, bytes1) PW_ATTR1(KS, false, 1, bytes1) if (pl.ld <= 65) { PW_ATTR1(KS, true, 2, bytes2) PW_ATTR1(KS, false, 2, bytes2) } return hipSuccess; }
    switch (pl.ks) {
        case 4: PW_ATTR(4);
        case 8: PW_ATTR(8);
        case 16: PW_ATTR(16);
        case 25: PW_ATTR(25);
        case 32: PW_ATTR(32);
        case 50: PW_ATTR(50);
        case 64: PW_ATTR(64);
        default: return hipSuccess;
    }
#undef PW_ATTR1
#undef PW_ATTR
}

hipError_t launch_pairwise_fb(const float* F1, const float* F2, const float* Bm, const float* tF1, const float* tF2,
                              const float* tB, const float* discount, int B, int d, int ld, float ortho_coef,
                              float* dF1, float* dF2, float* dB, float* metrics, float* scratch, hipStream_t s,
                              StepState* adv, int adv_which, const float* y, const float* norms, float* dy, float out_scale) {
    return launch_pairwise_fb_block(F1, F2, Bm, tF1, tF2, tB, discount, B, d, ld, ortho_coef, 0, B, dF1, dF2, dB, metrics,
                                    scratch, s, adv, adv_which, y, norms, dy, out_scale);
}

// Rows [row_off, row_off + rows) of the loss on B-row panels: dF_i and dB of THOSE rows (each complete: the workgroups walk
// every J tile of the B rows), and this row block's share of the scalar sums with the B-row normalisers.
hipError_t launch_pairwise_fb_block(const float* F1, const float* F2, const float* Bm, const float* tF1, const float* tF2,
                                    const float* tB, const float* discount, int B, int d, int ld, float ortho_coef,
                                    int row_off, int rows, float* dF1, float* dF2, float* dB, float* metrics,
                                    float* scratch, hipStream_t s, StepState* adv, int adv_which, const float* y,
                                    const float* norms, float* dy, float out_scale) {
    const PwPlan pl = make_plan(B, d, rows);
    if (pl.ks < 0 || B < 2 || rows < 1 || row_off < 0 || row_off + rows > B) return hipErrorInvalidValue;
    if (rows != B && ((row_off & 31) || (rows & 31))) return hipErrorInvalidValue;      // whole 32-row blocks
    PwArgs a;
    a.F1 = F1; a.F2 = F2; a.Bm = Bm; a.tF1 = tF1; a.tF2 = tF2; a.tB = tB; a.discount = discount;
    a.B = B; a.d = d; a.ld = ld; a.ortho2 = 2.0f * ortho_coef; a.jpc = pl.jpc; a.njt = pl.njt;
    a.vec = ((ld & 3) == 0);
    for (const float* q : {F1, F2, Bm, tF1, tF2, tB}) if ((uintptr_t)q & 15) a.vec = 0;
    a.partial = scratch;
    a.scal = scratch + (size_t)pl.nchunks * PW_SLOTS * pl.Bp * pl.dp;
    a.Bp = pl.Bp;
    a.i_off = row_off;
    dim3 grid(pl.nI, pl.nchunks), block(256 * pl.ng);
    hipError_t e = hipSuccess;
#define PW_LAUNCH1(KS, NG_)                                                                                            \
    if (a.vec) hipLaunchKernelGGL((pairwise_kernel<KS, true, NG_>), grid, block, pl.lds_bytes, s, a);                  \
    else hipLaunchKernelGGL((pairwise_kernel<KS, false, NG_>), grid, block, pl.lds_bytes, s, a)
#define PW_LAUNCH(KS) if (pl.ng == 2) { PW_LAUNCH1(KS, 2); } else { PW_LAUNCH1(KS, 1); }
    switch (pl.ks) {
        case 4: PW_LAUNCH(4); break;
        case 8: PW_LAUNCH(8); break;
        case 16: PW_LAUNCH(16); break;
        case 25: PW_LAUNCH(25); break;
        case 32: PW_LAUNCH(32); break;
        case 50: PW_LAUNCH(50); break;
        case 64: PW_LAUNCH(64); break;
        default: return hipErrorInvalidValue;
    }
#undef PW_LAUNCH1
#undef PW_LAUNCH
    if (e != hipSuccess) return e;
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (d > 128 || (y != nullptr && (norms == nullptr || dy == nullptr))) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pairwise_reduce_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, a.partial, a.scal,
                       pl.nchunks, pl.nchunks * pl.nI, rows, pl.Bp, d, pl.dp, ld, B, ortho_coef, dF1, dF2, dB, metrics, adv,
                       adv_which, y, norms, dy, out_scale);
    return hipGetLastError();
}

}  // namespace fbhip
