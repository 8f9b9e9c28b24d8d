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
//   cov    (waves 2 and 3, half each):         C = B[J] . B[I]^T (symmetric)      -> dB[I]  += 2 c_o H-contract . B[J]
// I-side operands live in registers as MFMA B-fragments for the whole kernel; J-side tiles are staged in LDS and
// shared by the four waves.  Partial results over J-chunks go to a scratch buffer and are folded in a fixed
// order by pairwise_reduce_kernel (deterministic; no atomics).  Scalar sums are wave-shuffle reduced.
//
// Every wave computes ONE of the two target products of its role (wave i the one with target F_i) and the two waves of a role
// swap them through LDS (min(t1, t2) is what both need); the covariance tile is computed half of K by wave 2 and half by wave 3,
// swapped the same way, and each contracts one half of its rows -- every wave has the same work in every iteration (the
// workgroup-wide barrier per J tile makes the slowest wave of an ITERATION the pace, not the average).
//
// LDS panels are [32][LD], LD = W + 4: a lane reads its row's k values four at a time (ds_read_b128; row stride = 4 banks, so
// 8 lanes cover the 32 banks: conflict-free) and a quad feeds four MFMAs -- the k order of the contraction is permuted
// (lane half h takes k = 8 q + 4 h + m), identically on the register operand.
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

// Three panels of one 32-row block (zero filled outside [0,B) x [0,d)): global -> registers -> LDS [32][LD], one float4 quad per
// "op" (op = 3 i + m: quad i of this thread, matrix m).  load() only ISSUES the global load (branch-free on the aligned path:
// out-of-range quads read a clamped, valid address), the zero-fill masks are applied in store().  The kernel never stages a whole
// tile at once: the ops are handed out one or two at a time to the k-steps of a product (tile_mm's ``side``), so the loads, the
// selects and the LDS writes issue in the shadow of the matrix pipe (one wave per SIMD: nothing else would hide them).
template <int LD, bool VEC>
struct Stage3 {
    static constexpr int W = LD - 4;                 // multiple of 32
    static constexpr int QPT = 32 * (W / 4) / 256;   // float4 quads per thread per matrix (1, 2, 4 for W = 32, 64, 128)
    static constexpr int NOPS = 3 * QPT;
    float4 v[NOPS];

    __device__ __forceinline__ void load(int op, const float* __restrict__ X, int ld, int row0, int B, int d, int tid) {
        const int q = tid + (op / 3) * 256;
        const int r = q / (W / 4), n0 = 4 * (q % (W / 4));
        if constexpr (VEC) {
            const bool ok = (row0 + r < B) && (n0 < ld);                     // ld % 4 == 0: the quad stays inside its row
            const size_t off = ok ? (size_t)(row0 + r) * ld + n0 : 0;
            v[op] = *reinterpret_cast<const float4*>(X + off);
        } else {
            const int gr = row0 + r;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gr < B && n0 < d) {
                const float* ptr = X + (size_t)gr * ld + n0;
                x.x = ptr[0];
                if (n0 + 1 < d) x.y = ptr[1];
                if (n0 + 2 < d) x.z = ptr[2];
                if (n0 + 3 < d) x.w = ptr[3];
            }
            v[op] = x;
        }
    }
    __device__ __forceinline__ void store(int op, float* __restrict__ panel, int row0, int B, int d, int tid) const {
        const int q = tid + (op / 3) * 256;
        const int r = q / (W / 4), n0 = 4 * (q % (W / 4));
        const bool row_ok = row0 + r < B;
        const bool k0 = row_ok && n0 < d, k1 = row_ok && n0 + 1 < d, k2 = row_ok && n0 + 2 < d, k3 = row_ok && n0 + 3 < d;
        *reinterpret_cast<float4*>(panel + r * LD + n0) =
            make_float4(k0 ? v[op].x : 0.f, k1 ? v[op].y : 0.f, k2 ? v[op].z : 0.f, k3 ? v[op].w : 0.f);
    }
};

struct NoSide { __device__ __forceinline__ void operator()(int, int) const {} };

// acc(32x32) = rowsJ (LDS, [32][LD]) . fragI^T over KS k-steps, in the permuted k order of load_frag: step 4 q + m contracts
// k = 8 q + 4 h + m (one ds_read_b128 per four steps), the KS % 4 tail steps k = 8 (KS / 4) + 2 i + h.
// PART 0: all of K; 1: the first half of the quads; 2: the other quads and the tail (1 + 2 = 0 as sets).
// side(s, n): the caller's slot s of n for unrelated work (staging ops).
// A wave issues in order and a dependent (or merely next) MFMA waits ~60 cycles for the matrix pipe: whatever is to hide behind
// the MFMAs has to sit BETWEEN two of them, a few instructions per gap, and hipcc has to be kept from moving it (it otherwise emits
// read -> s_waitcnt lgkmcnt(0) -> 4 MFMAs, and collects the VALU work in front): one scheduling fence per MFMA, the read of quad
// q + 1 and one side slot right behind each MFMA of quad q.
template <int KS, int LD, int PART, class Side>
__device__ __forceinline__ floatx16 tile_mm(const float* __restrict__ sJ, const float (&fragI)[KS], int l31, int h, Side&& side) {
    constexpr int KQ = KS / 4, KT = KS % 4;
    constexpr int Q0 = PART == 2 ? KQ / 2 : 0, Q1 = PART == 1 ? KQ / 2 : KQ;
    constexpr bool TAIL = PART != 1 && KT > 0;
    constexpr int NS = 4 * (Q1 - Q0);
    floatx16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const float* a = sJ + l31 * LD + 4 * h;
    const float* t = sJ + l31 * LD + 8 * KQ + h;
    float4 v[2];
    float tl[KT > 0 ? KT : 1];
    if constexpr (Q1 > Q0) v[0] = *reinterpret_cast<const float4*>(a + 8 * Q0);
    else if constexpr (TAIL) {
#pragma unroll
        for (int i = 0; i < KT; ++i) tl[i] = t[2 * i];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = Q0; q < Q1; ++q) {
        const int c = (q - Q0) & 1;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[c].x, fragI[4 * q + 0], acc, 0, 0, 0);
        if (q + 1 < Q1) v[c ^ 1] = *reinterpret_cast<const float4*>(a + 8 * (q + 1));
        else if constexpr (TAIL) {
#pragma unroll
            for (int i = 0; i < KT; ++i) tl[i] = t[2 * i];
        }
        side(4 * (q - Q0) + 0, NS);
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[c].y, fragI[4 * q + 1], acc, 0, 0, 0);
        side(4 * (q - Q0) + 1, NS);
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[c].z, fragI[4 * q + 2], acc, 0, 0, 0);
        side(4 * (q - Q0) + 2, NS);
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[c].w, fragI[4 * q + 3], acc, 0, 0, 0);
        side(4 * (q - Q0) + 3, NS);
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (TAIL) {
#pragma unroll
        for (int i = 0; i < KT; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(tl[i], fragI[4 * KQ + i], acc, 0, 0, 0);
    }
    return acc;
}

// out[nt] += G^T-contract . X[J]:  out[c][n] += sum_r G[r][c] * X[r][n] over the accumulator registers [R0, R0 + NR) of the tile
// (two tile rows each).  gen(reg) PRODUCES G[reg] -- the loss epilogue of that register (mask, diagonal, scalar sums) -- and is
// called one register group (4 MFMAs) ahead, behind an MFMA of the running group, like the LDS reads of the next group's operands.
template <int NT, int LD, int R0, int NR, class Gen>
__device__ __forceinline__ void contract_gen(floatx16 (&out)[NT], Gen&& gen, const float* __restrict__ sX, int l31, int h) {
    static_assert(NT == 1 || NT == 2 || NT == 4, "column tiles");
    constexpr int GS = 4 / NT;                                        // registers per group: 4 MFMAs a group
    static_assert(NR % GS == 0, "register groups");
    constexpr int NGR = NR / GS;
    float a[2][GS], b[2][4];
    auto load = [&](int g, float (&dst)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < GS; ++i) {
            const float* src = sX + acc_row(R0 + g * GS + i, h) * LD + l31;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) dst[i * NT + nt] = src[32 * nt];
        }
    };
    load(0, b[0]);
#pragma unroll
    for (int i = 0; i < GS; ++i) a[0][i] = gen(R0 + i);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < NGR; ++g) {
        const int cur = g & 1, nxt = cur ^ 1;
#pragma unroll
        for (int idx = 0; idx < 4; ++idx) {
            out[idx % NT] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][idx / NT], b[cur][idx], out[idx % NT], 0, 0, 0);
            if (g + 1 < NGR) {
                if (idx == 0) load(g + 1, b[nxt]);
                if (idx < GS) a[nxt][idx] = gen(R0 + (g + 1) * GS + idx);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int KS, int LD>
__device__ __forceinline__ void load_frag(float (&f)[KS], const float* __restrict__ sI, int l31, int h) {
    constexpr int KQ = KS / 4, KT = KS % 4;
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(sI + l31 * LD + 8 * q + 4 * h);
        f[4 * q + 0] = v.x; f[4 * q + 1] = v.y; f[4 * q + 2] = v.z; f[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int i = 0; i < KT; ++i) f[4 * KQ + i] = sI[l31 * LD + 8 * KQ + 2 * i + h];
}

// a wave parks / fetches one 32x32 accumulator: [4][64 lanes] float4, lanes adjacent
__device__ __forceinline__ void put16(float* __restrict__ dst, const floatx16& v, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
        *reinterpret_cast<float4*>(dst + (i * 64 + lane) * 4) = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
}
__device__ __forceinline__ floatx16 get16(const float* __restrict__ src, int lane) {
    floatx16 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 u = *reinterpret_cast<const float4*>(src + (i * 64 + lane) * 4);
        v[4 * i] = u.x; v[4 * i + 1] = u.y; v[4 * i + 2] = u.z; v[4 * i + 3] = u.w;
    }
    return v;
}

constexpr int PW_XCH = 6 * 1024;                                           // floats: t x 2 waves, (u, C-half) x 2 waves
__host__ __device__ constexpr int pw_group_floats(int LD) { return 6 * 32 * LD + 32 + PW_XCH; }

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
    constexpr int LD = W + 4;                       // row stride = 4 banks: conflict-free for the b128 row reads and the b32 column reads
    constexpr int GRP = pw_group_floats(LD);        // floats per group: 6 x [32][LD] + gamma[32] + the swap area
    extern __shared__ __attribute__((aligned(16))) float lds_all[];        // NG x GRP
    const int grp = NG > 1 ? (int)(threadIdx.x >> 8) : 0;
    float* lds = lds_all + grp * GRP;
    float* sBm = lds;
    float* stB = sBm + 32 * LD;
    float* sF1 = stB + 32 * LD;
    float* sF2 = sF1 + 32 * LD;
    float* stF1 = sF2 + 32 * LD;
    float* stF2 = stF1 + 32 * LD;
    float* sGam = stF2 + 32 * LD;
    float* xch = sGam + 32;                         // [0, 2048): t of waves 0, 1; [2048, 6144): (u, C half) of waves 2, 3

    const int tid = threadIdx.x & 255, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int I0 = blockIdx.x * 32 + p.i_off, chunk = blockIdx.y;
    const int B = p.B, d = p.d;
    const float n_off = (float)B * (float)(B - 1);
    const float inv_noff = 1.0f / n_off, inv_b = 1.0f / (float)B;

    // the panels in two halves: T = {tB, tF1, tF2} (the target products), M = {Bm, F1, F2} (the M_i / covariance products and
    // every contraction)
    const float* const srcT[3] = {p.tB, p.tF1, p.tF2};
    const float* const srcM[3] = {p.Bm, p.F1, p.F2};
    float* const dstT[3] = {stB, stF1, stF2};
    float* const dstM[3] = {sBm, sF1, sF2};
    using Stage = Stage3<LD, VEC>;
    constexpr int NOPS = Stage::NOPS;
    Stage R;

    // ---- I-side fragments (registers, whole kernel) ----------------------------------------------------
#pragma unroll
    for (int op = 0; op < NOPS; ++op) R.load(op, srcT[op % 3], p.ld, I0, B, d, tid);
#pragma unroll
    for (int op = 0; op < NOPS; ++op) R.store(op, dstT[op % 3], I0, B, d, tid);
#pragma unroll
    for (int op = 0; op < NOPS; ++op) R.load(op, srcM[op % 3], p.ld, I0, B, d, tid);
#pragma unroll
    for (int op = 0; op < NOPS; ++op) R.store(op, dstM[op % 3], I0, B, d, tid);
    __syncthreads();
    float fa[KS], fb[KS];
    if (wid < 2) {                   // role 1, wave i: F_i[I], tF_i[I]
        load_frag<KS, LD>(fa, wid == 0 ? sF1 : sF2, l31, h);
        load_frag<KS, LD>(fb, wid == 0 ? stF1 : stF2, l31, h);
    } else {                         // role 2: B[I], tB[I]
        load_frag<KS, LD>(fa, sBm, l31, h);
        load_frag<KS, LD>(fb, stB, l31, h);
    }
    const int gcol = I0 + l31;                              // batch index of this lane's tile column
    const float gam_col = (gcol < B) ? p.discount[gcol] : 0.f;
    const float cdiag = (gcol < B) ? -inv_b : 0.f;          // d loss / d M on the diagonal
    __syncthreads();

    floatx16 out[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) out[nt][i] = 0.f;
    float s_sq = 0.f, s_diag = 0.f, s_all = 0.f, s_tall = 0.f, s_csq = 0.f, s_cdiag = 0.f;

    const int jt_begin = chunk * p.jpc;
    const int jt_end = min(jt_begin + p.jpc, p.njt);
    const float* sFi = (wid & 1) ? sF2 : sF1;               // role 2, wave 2 + i: F_i[J]
    float* my_x = wid < 2 ? xch + wid * 1024 : xch + 2048 + (wid - 2) * 2048;
    const float* their_x = wid < 2 ? xch + (wid ^ 1) * 1024 : xch + 2048 + ((wid - 2) ^ 1) * 2048;

    // Staging schedule.  The T panels of a tile are read by the target products only (phase A), the M panels by everything
    // after them (phase B): while phase A runs the M slots are free, while phase B runs the T slots are.  ONE register set R:
    //   phase A(tile): R (= M panels of this tile) -> M slots;  R <- T panels of the next tile
    //   phase B(tile): R (= T panels of the next tile) -> T slots;  R <- M panels of the next tile
    // each op riding on one k-step of a product.  Before the loop: T panels of the first tile in LDS, its M panels in R.
    {
        const int J0 = min(jt_begin + grp, p.njt - 1) * 32;
#pragma unroll
        for (int op = 0; op < NOPS; ++op) R.load(op, srcT[op % 3], p.ld, J0, B, d, tid);
#pragma unroll
        for (int op = 0; op < NOPS; ++op) R.store(op, dstT[op % 3], J0, B, d, tid);
#pragma unroll
        for (int op = 0; op < NOPS; ++op) R.load(op, srcM[op % 3], p.ld, J0, B, d, tid);
    }
    // group g takes the tiles jt_begin + g, jt_begin + g + NG, ...; every group runs the same number of iterations (the
    // barriers are workgroup-wide), a group without a tile in the last one only keeps them company
#pragma unroll 1
    for (int jt0 = jt_begin; jt0 < jt_end; jt0 += NG) {
        const int jt = jt0 + grp;
        const bool live = jt < jt_end;
        const int J0 = min(jt, p.njt - 1) * 32;
        const int Jn = min(jt + NG, p.njt - 1) * 32;        // (past the chunk's last tile: a clamped, unused tile)
        __syncthreads();                                    // T(jt) in LDS; the M slots and gamma are free
        floatx16 tm, G;
        if (live) {
            // ---- phase A: the wave's own target product (parked for the other wave of the role)
            if (tid < 32) sGam[tid] = (J0 + tid < B) ? p.discount[J0 + tid] : 0.f;
            auto side = [&](int sidx, int n) __attribute__((always_inline)) {      // slot sidx of n: its share of 2 NOPS sub-ops
#pragma unroll
                for (int u = sidx * (2 * NOPS) / n; u < (sidx + 1) * (2 * NOPS) / n; ++u) {
                    if (u & 1) R.load(u >> 1, srcT[(u >> 1) % 3], p.ld, Jn, B, d, tid);
                    else R.store(u >> 1, dstM[(u >> 1) % 3], J0, B, d, tid);
                }
            };
            // role 1: tile rows = t (J), cols = s (I):  T[r][c] = M_i[s = I0+c][t = J0+r]
            // role 2: tile rows = s (J), cols = t (I):  P[r][c] = M_i[s = J0+r][t = I0+c]
            tm = tile_mm<KS, LD, 0>(wid < 2 ? stB : ((wid & 1) ? stF2 : stF1), fb, l31, h, side);
            put16(my_x, tm, lane);
        }
        __syncthreads();                                    // M(jt), gamma and the target products in LDS; the T slots are free
        if (live) {
            // ---- phase B: (role 2) this wave's half of K of the covariance tile C[r][c] = B[J0+r] . B[I0+c]
            // (fb_ddpg.py:344-348), parked; then the M_i tile
            if (wid >= 2) {
                floatx16 Ch;
                if (wid == 2) Ch = tile_mm<KS, LD, 1>(sBm, fa, l31, h, NoSide());
                else Ch = tile_mm<KS, LD, 2>(sBm, fa, l31, h, NoSide());
                put16(my_x + 1024, Ch, lane);
            }
            auto side = [&](int sidx, int n) __attribute__((always_inline)) {
#pragma unroll
                for (int u = sidx * (2 * NOPS) / n; u < (sidx + 1) * (2 * NOPS) / n; ++u) {
                    if (u & 1) R.load(u >> 1, srcM[(u >> 1) % 3], p.ld, Jn, B, d, tid);
                    else R.store(u >> 1, dstT[(u >> 1) % 3], Jn, B, d, tid);
                }
            };
            G = tile_mm<KS, LD, 0>(wid < 2 ? sBm : sFi, fa, l31, h, side);
        }
        __syncthreads();                                    // covariance halves in LDS
        if (live) {
            const bool dtile = (J0 == I0);                  // the only tile of this column block that holds diagonal entries
            {
                const floatx16 t2 = get16(their_x, lane);
#pragma unroll
                for (int i = 0; i < 16; ++i) tm[i] = fminf(tm[i], t2[i]);
            }
            if (wid < 2) {
                auto gen = [&](int reg) __attribute__((always_inline)) -> float {
                    const float m = G[reg], t = tm[reg];
                    const float dlt = m - gam_col * t;                 // discount is indexed by s = column here
                    const bool diag = dtile && (J0 + acc_row(reg, h) == gcol);
                    s_all += m;
                    s_tall += t;
                    s_diag += diag ? m : 0.f;
                    s_sq += diag ? 0.f : dlt * dlt;
                    return diag ? cdiag : dlt * inv_noff;
                };
                contract_gen<NT, LD, 0, 16>(out, gen, sBm, l31, h);            // dF_i[I] += G^T-contract . B[J]
            } else {
                float gj[16];                                                  // discount of this lane's 16 tile rows (s = row here)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 u = *reinterpret_cast<const float4*>(sGam + 8 * i + 4 * h);
                    gj[4 * i] = u.x; gj[4 * i + 1] = u.y; gj[4 * i + 2] = u.z; gj[4 * i + 3] = u.w;
                }
                auto gen = [&](int reg) __attribute__((always_inline)) -> float {
                    const float dlt = G[reg] - gj[reg] * tm[reg];
                    const bool diag = dtile && (J0 + acc_row(reg, h) == gcol);
                    return diag ? cdiag : dlt * inv_noff;
                };
                contract_gen<NT, LD, 0, 16>(out, gen, sFi, l31, h);            // dB[I] += G-contract . F_i[J]
                // L_orth = mean_offdiag C^2 - 2 mean_diag C;  dL/dC = Hm = 2C/N_off (off-diag), -2/B (diag);
                // C = B B^T  =>  dB = ortho * (Hm + Hm^T) B = ortho * 2 * Hm . B.  With ortho2 = 2*ortho the tile
                // coefficient is ortho2 * Hm = 2 * (ortho2*C/N_off) resp. 2 * (-ortho2/B).  Wave 2 owns accumulator
                // registers 0..7 (scalar sums and contraction), wave 3 registers 8..15; C = half of wave 2 + half of wave 3.
                const float* cA = xch + 2048 + 1024, * cB = xch + 2048 + 2048 + 1024;
                float cc[8];
                auto fetch = [&](int f4) __attribute__((always_inline)) {      // accumulator registers 4 f4 .. 4 f4 + 7
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const float4 x = *reinterpret_cast<const float4*>(cA + ((f4 + i) * 64 + lane) * 4);
                        const float4 y = *reinterpret_cast<const float4*>(cB + ((f4 + i) * 64 + lane) * 4);
                        cc[4 * i] = x.x + y.x; cc[4 * i + 1] = x.y + y.y; cc[4 * i + 2] = x.z + y.z; cc[4 * i + 3] = x.w + y.w;
                    }
                };
                const float hdiag = 2.0f * p.ortho2 * cdiag, hoff = 2.0f * p.ortho2 * inv_noff;
                if (wid == 2) {
                    fetch(0);
                    auto genc = [&](int reg) __attribute__((always_inline)) -> float {
                        const float c = cc[reg];
                        const bool diag = dtile && (J0 + acc_row(reg, h) == gcol);
                        s_cdiag += diag ? c : 0.f;
                        s_csq += diag ? 0.f : c * c;
                        return diag ? hdiag : hoff * c;
                    };
                    contract_gen<NT, LD, 0, 8>(out, genc, sBm, l31, h);
                } else {
                    fetch(2);
                    auto genc = [&](int reg) __attribute__((always_inline)) -> float {
                        const float c = cc[reg - 8];
                        const bool diag = dtile && (J0 + acc_row(reg, h) == gcol);
                        s_cdiag += diag ? c : 0.f;
                        s_csq += diag ? 0.f : c * c;
                        return diag ? hdiag : hoff * c;
                    };
                    contract_gen<NT, LD, 8, 8>(out, genc, sBm, l31, h);
                }
            }
        }
    }
    __syncthreads();                                        // (the hand-off below reuses the panels)

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
    pl.ld = w + 4;
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
    pl.ng = (w <= 64 && pl.jpc >= 2) ? 2 : 1;
    pl.lds_bytes = (size_t)pl.ng * pw_group_floats(pl.ld) * sizeof(float);
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
    const int bytes1 = (int)((size_t)pw_group_floats(pl.ld) * sizeof(float)), bytes2 = 2 * bytes1;
#define PW_ATTR1(KS, VEC, NG_, BYTES)                                                                                  \
    if ((BYTES) > 48 * 1024) {                                                                                        \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pairwise_kernel<KS, VEC, NG_>),              \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (BYTES));                      \
        if (e != hipSuccess) return e;                                                                                \
    }
#define PW_ATTR(KS) { PW_ATTR1(KS, true, 1, bytes1) PW_ATTR1(KS, false, 1, bytes1) if (pl.ld <= 68) { PW_ATTR1(KS, true, 2, bytes2) PW_ATTR1(KS, false, 2, bytes2) } return hipSuccess; }
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
