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
//   cov    (product: role 1, half of K per wave; contraction: role 2, half of the rows per wave):
//                                                C = B[J] . B[I]^T (symmetric)      -> dB[I]  += 2 c_o H-contract . B[J]
// I-side operands live in registers as MFMA B-fragments for the whole kernel; J-side tiles are staged in LDS and
// shared by the four waves.  Partial results over J-chunks go to a scratch buffer and are folded in a fixed
// order by pairwise_reduce_kernel (deterministic; no atomics).  Scalar sums: fp64 wave sums, (hi, lo) float partials.
//
// Balance.  Every wave computes ONE of the two target products of its role (wave i the one with target F_i) and the two waves of
// a role swap them through LDS (min(t1, t2) is what both need).  The covariance PRODUCT is computed by the role-1 waves (half of K
// each, parked in LDS behind a flag), its CONTRACTION by the role-2 waves (half of the tile rows each) -- 2.5 products + 1
// contraction for role 1, 2 products + 1.5 contractions for role 2 (189 / 196 MFMAs per tile at d = 100) and two workgroup
// barriers per tile: a barrier makes the slowest wave of an ITERATION the pace, not the average.
//
// Vector-ALU work.  A wave's own v_* instructions do not overlap its MFMAs (tools/mfma_shadow.hip; LDS and global instructions
// do), and at d = 100 the registers allow one wave per SIMD: staging, epilogue and bookkeeping are arranged so that the tile loop
// holds almost none (see Stage3, mm_steps, contract_gen and the kernel body).
//
// LDS panels are [32][LD], LD = W + 4: a lane reads its row's k values four at a time (ds_read_b128; row stride = 4 banks, so
// 8 lanes cover the 32 banks: conflict-free) and a quad feeds four MFMAs -- the k order of the contraction is permuted
// (lane half h takes k = 8 q + 4 h + m), identically on the register operand.
#include "common.h"
#include "fbhip.h"
#include <type_traits>
#include <utility>

namespace fbhip {

namespace {

typedef float fx4 __attribute__((ext_vector_type(4)));

constexpr int PW_SLOTS = 4;        // dF1, dF2, dB (wave 2), dB (wave 3)
constexpr int PW_SCAL = 20;        // scalar partials per workgroup: 10 sums as (hi, lo) float pairs, lo at [10 + k]

// a wave's sum of per-lane fp32 partials, added in fp64 (the loss is a small difference of sums hundreds of times larger: the
// fp32 rounding of the partials is what the 2e-5 / 5e-6 tolerances of the free-running parity tests see first)
__device__ __forceinline__ double wsum_d(float v) {
    double s = (double)v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    return s;
}
__device__ __forceinline__ void put_hilo(float* sc, int k, double s) {
    const float hi = (float)s;
    sc[k] = hi;
    sc[10 + k] = (float)(s - (double)hi);
}
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
    int vec;                   // staging mode (Stage3): 0 scalar loads, 1 aligned panels with zero-fill masks, 2 aligned, whole, d == 2 KS
    float* partial;            // [nchunks][PW_SLOTS][Bp][DP]
    float* scal;               // [nblocks][PW_SCAL]
    int Bp;
    int i_off;                 // block mode (global-batch data parallel): the workgroups' I rows are [i_off, i_off + 32 gridDim.x)
                               // of the B-row panels; outputs / partials are indexed by the LOCAL row (I - i_off)
};

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{}) -- indices that select
// registers (array elements held in VGPRs) must be constants in the source, not after some unrolling pass
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// Three panels of one 32-row block: global -> registers -> LDS [32][LD], one float4 quad per "op" (op = 3 i + m: quad i of this
// thread, matrix m).  load() only ISSUES the global load, store() writes LDS.  The kernel never stages a whole tile at once: the
// ops are handed out one at a time to the k-steps of a product (mm_steps' ``side``).
// A wave's VALU instructions do NOT overlap its own MFMAs (tools/mfma_shadow.hip: +5..8 cycles per instruction between two
// MFMAs, while ds_read / ds_write / global_load are free), and with one wave per SIMD nothing else hides them, so MODE 2 -- the
// product's case: 16-byte aligned panels, ld % 4 == 0, B % 32 == 0, d == 2 KS -- stages with NO vector ALU work in the loop:
// per-thread global and LDS offsets are loop invariants, the tile base is scalar, and nothing is zero-filled: quads at or past
// ld land in the row's 4-float pad (never read), columns [d, W) of the panels only ever reach output columns >= d (dropped).
// MODE 1: aligned panels, zero-fill masks applied in store() (ragged B or d < 2 KS); MODE 0: scalar loads (unaligned panels).
template <int LD, int MODE>
struct Stage3 {
    static constexpr int W = LD - 4;                 // multiple of 32
    static constexpr int QPT = 32 * (W / 4) / 256;   // float4 quads per thread per matrix (1, 2, 4 for W = 32, 64, 128)
    static constexpr int NOPS = 3 * QPT;
    fx4 v[NOPS];                                     // (a native vector, not HIP's float4 struct: struct copies become memcpys between
                                                     //  address spaces that keep the whole stage in scratch memory)
    unsigned goff[QPT], loff[QPT];                   // MODE 2: float offsets inside the tile (global) / the panel (LDS)

    __device__ __forceinline__ void init(int ld, int tid) {
        if constexpr (MODE == 2) {
            static_for<QPT>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                const int q = tid + i * 256;
                const int r = q / (W / 4), n0 = 4 * (q % (W / 4));
                const bool ok = n0 < ld;
                goff[i] = (unsigned)(r * ld + (ok ? n0 : 0));
                loff[i] = (unsigned)(r * LD + (ok ? n0 : W));
            });
        }
    }
    __device__ __forceinline__ void load(int op, const float* __restrict__ X, int ld, int row0, int B, int d, int tid) {
        if constexpr (MODE == 2) {
            const float* tile = X + (size_t)row0 * ld;                           // (uniform: scalar registers)
            v[op] = *reinterpret_cast<const fx4*>(tile + goff[op / 3]);
        } else if constexpr (MODE == 1) {
            const int q = tid + (op / 3) * 256;
            const int r = q / (W / 4), n0 = 4 * (q % (W / 4));
            const bool ok = (row0 + r < B) && (n0 < ld);                     // ld % 4 == 0: the quad stays inside its row
            const size_t off = ok ? (size_t)(row0 + r) * ld + n0 : 0;
            v[op] = *reinterpret_cast<const fx4*>(X + off);
        } else {
            const int q = tid + (op / 3) * 256;
            const int r = q / (W / 4), n0 = 4 * (q % (W / 4));
            const int gr = row0 + r;
            fx4 x = {0.f, 0.f, 0.f, 0.f};
            if (gr < B && n0 < d) {
                const float* ptr = X + (size_t)gr * ld + n0;
                x[0] = ptr[0];
                if (n0 + 1 < d) x[1] = ptr[1];
                if (n0 + 2 < d) x[2] = ptr[2];
                if (n0 + 3 < d) x[3] = ptr[3];
            }
            v[op] = x;
        }
    }
    __device__ __forceinline__ void store(int op, float* __restrict__ panel, int row0, int B, int d, int tid) const {
        if constexpr (MODE == 2) {
            *reinterpret_cast<fx4*>(panel + loff[op / 3]) = v[op];
        } else {
            const int q = tid + (op / 3) * 256;
            const int r = q / (W / 4), n0 = 4 * (q % (W / 4));
            const bool row_ok = row0 + r < B;
            const bool k0 = row_ok && n0 < d, k1 = row_ok && n0 + 1 < d, k2 = row_ok && n0 + 2 < d, k3 = row_ok && n0 + 3 < d;
            const fx4 z = {k0 ? v[op][0] : 0.f, k1 ? v[op][1] : 0.f, k2 ? v[op][2] : 0.f, k3 ? v[op][3] : 0.f};
            *reinterpret_cast<fx4*>(panel + r * LD + n0) = z;
        }
    }
};

struct NoSide { template <class A, class B> __device__ __forceinline__ void operator()(A, B) const {} };

// acc(32x32) = rowsJ (LDS) . frag^T over 4 NQ + KT k-steps: quad j of the row (``aq`` + 8 j: one ds_read_b128) feeds the steps
// 4 j .. 4 j + 3, the KT tail steps read ``at`` + 2 i.  The k order of a product is permuted (load_frag: step 4 q + m contracts
// k = 8 q + 4 h + m, tail step i contracts k = 8 (KS / 4) + 2 i + h), identically on both operands.
// side(s, n): the caller's slot s of n for unrelated memory work (staging ops), one slot behind every MFMA.
// hipcc has to be kept from reordering (it otherwise emits read -> s_waitcnt lgkmcnt(0) -> 4 MFMAs into ONE register quad):
// one scheduling fence per MFMA, the read of quad j + 1 right behind the first MFMA of quad j.
template <int NQ, int KT, class Side>
__device__ __forceinline__ floatx16 mm_steps(const float* __restrict__ aq, const float* __restrict__ at,
                                             const float (&frag)[4 * NQ + KT], Side&& side) {
    using NS = std::integral_constant<int, 4 * NQ>;
    floatx16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float4 v[2];
    float tl[KT > 0 ? KT : 1];
    if constexpr (NQ > 0) v[0] = *reinterpret_cast<const float4*>(aq);
    else if constexpr (KT > 0) {
#pragma unroll
        for (int i = 0; i < KT; ++i) tl[i] = at[2 * i];
    }
    __builtin_amdgcn_sched_barrier(0);
    static_for<NQ>([&](auto qc) __attribute__((always_inline)) {
        constexpr int q = decltype(qc)::value, c = q & 1;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[c].x, frag[4 * q + 0], acc, 0, 0, 0);
        if constexpr (q + 1 < NQ) v[c ^ 1] = *reinterpret_cast<const float4*>(aq + 8 * (q + 1));
        else if constexpr (KT > 0) {
#pragma unroll
            for (int i = 0; i < KT; ++i) tl[i] = at[2 * i];
        }
        side(std::integral_constant<int, 4 * q + 0>{}, NS{});
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[c].y, frag[4 * q + 1], acc, 0, 0, 0);
        side(std::integral_constant<int, 4 * q + 1>{}, NS{});
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[c].z, frag[4 * q + 2], acc, 0, 0, 0);
        side(std::integral_constant<int, 4 * q + 2>{}, NS{});
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[c].w, frag[4 * q + 3], acc, 0, 0, 0);
        side(std::integral_constant<int, 4 * q + 3>{}, NS{});
        __builtin_amdgcn_sched_barrier(0);
    });
#pragma unroll
    for (int i = 0; i < KT; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(tl[i], frag[4 * NQ + i], acc, 0, 0, 0);
    return acc;
}
// all of K
template <int KS, int LD, class Side>
__device__ __forceinline__ floatx16 tile_mm(const float* __restrict__ sJ, const float (&fragI)[KS], int l31, int h, Side&& side) {
    return mm_steps<KS / 4, KS % 4>(sJ + l31 * LD + 4 * h, sJ + l31 * LD + 8 * (KS / 4) + h, fragI, side);
}

// out[nt] += G^T-contract . X[J]:  out[c][n] += sum_r G[r][c] * X[r][n] over the accumulator registers [R0, R0 + NR) of the tile
// (two tile rows each).  gen(reg) PRODUCES G[reg] -- the loss epilogue of that register (mask, diagonal, scalar sums) -- and is
// called one register group (4 MFMAs) ahead, behind an MFMA of the running group, like the LDS reads of the next group's operands.
template <int NT, int LD, int R0, int NR, class Gen>
__device__ __forceinline__ void contract_gen(floatx16 (&out)[NT], Gen&& gen, const float* __restrict__ sX, int l31, int h) {     // (sX: the panel at the first row of this call's registers)
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

constexpr int PW_XCH = 6 * 1024 + 4;                                       // floats: 6 parked tiles (below) + 2 ready flags
__host__ __device__ constexpr int pw_group_floats(int LD) { return 6 * 32 * LD + 32 + PW_XCH; }

// MODE: see Stage3 (separate instantiations, not a runtime branch).
// NG wave groups of four waves each walk ALTERNATE J tiles of the workgroup's chunk concurrently (own LDS panels, own
// accumulators; group g > 0 hands its sums to group 0 through LDS at the end, in group order: deterministic).  With one group a
// SIMD holds a single wave whose every VALU instruction, LDS round trip and barrier leaves the matrix pipe idle; two groups put
// two independent chains on every SIMD.  NG = 2 needs 2 x 6 panels in LDS and <= 256 registers a wave: d <= 64.
template <int KS, int MODE, int NG>
__global__ void __launch_bounds__(256 * NG) pairwise_kernel(const PwArgs p) {
    constexpr int NT = (2 * KS + 31) / 32;
    constexpr int W = (2 * KS > 32 * NT ? 2 * KS : 32 * NT);
    constexpr int LD = W + 4;                       // row stride = 4 banks: conflict-free for the b128 row reads and the b32 column reads
    constexpr int GRP = pw_group_floats(LD);        // floats per group: 6 x [32][LD] + gamma[32] + the swap area
    constexpr int KQ = KS / 4, KT = KS % 4;
    constexpr int NQH = KQ - KQ / 2;                // quads of the larger half of K (the covariance product is split over K)
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
    float* xch = sGam + 32;                         // parked 32x32 tiles, 1024 floats each: t of wave 0, 1; u of wave 2, 3; the
                                                    // covariance tile's K-halves of wave 0, 1; then the two "half is parked" flags
    int* cflag = reinterpret_cast<int*>(xch + 6 * 1024);

    const int tid = threadIdx.x & 255, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int I0 = blockIdx.x * 32 + p.i_off, chunk = blockIdx.y;
    const int B = p.B, d = p.d;

    // the panels in two halves: T = {tB, tF1, tF2} (the target products), M = {Bm, F1, F2} (the M_i / covariance products and
    // every contraction)
    const float* const srcT[3] = {p.tB, p.tF1, p.tF2};
    const float* const srcM[3] = {p.Bm, p.F1, p.F2};
    float* const dstT[3] = {stB, stF1, stF2};
    float* const dstM[3] = {sBm, sF1, sF2};
    using Stage = Stage3<LD, MODE>;
    constexpr int NOPS = Stage::NOPS;
    Stage R;
    R.init(p.ld, tid);

    // ---- I-side fragments (registers, whole kernel) ----------------------------------------------------
    // All four loads of the prologue -- both panel halves of the I block and of the first J tile -- are issued up front (the
    // registers are free until the fragments exist): one global round trip instead of four.
    const int jt_begin = chunk * p.jpc;
    const int jt_end = min(jt_begin + p.jpc, p.njt);
    // Tile order.  Exactly one tile of a column block holds diagonal entries (J0 == I0), and a contraction that can handle them
    // costs the others vector-ALU work they do not need; a second instantiation chosen per tile makes hipcc keep ``out`` in two
    // places and copy 64 registers per tile.  So the chunk's tiles are walked with the diagonal one (if the chunk has it) LAST,
    // and the last iteration of every wave is peeled: the loop body knows there is no diagonal, the peeled copy checks.
    const int ntl = jt_end - jt_begin;                      // tiles of this chunk (>= 1)
    const int jdiag = I0 >> 5;
    const bool has_diag = jdiag >= jt_begin && jdiag < jt_end;
    auto tile_at = [&](int k) __attribute__((always_inline)) -> int {          // k-th tile of the walk, k clamped to the chunk
        k = min(k, ntl - 1);
        if (!has_diag) return jt_begin + k;
        const int t = jt_begin + k;
        return k == ntl - 1 ? jdiag : (t >= jdiag ? t + 1 : t);
    };
    Stage RT0;                                              // T panels of the first J tile (R: its M panels)
    RT0.init(p.ld, tid);
    {
        Stage RA, RB;
        RA.init(p.ld, tid);
        RB.init(p.ld, tid);
        const int J0 = tile_at(grp) * 32;
        static_for<NOPS>([&](auto oc) __attribute__((always_inline)) { constexpr int op = decltype(oc)::value; RA.load(op, srcT[op % 3], p.ld, I0, B, d, tid); });
        static_for<NOPS>([&](auto oc) __attribute__((always_inline)) { constexpr int op = decltype(oc)::value; RB.load(op, srcM[op % 3], p.ld, I0, B, d, tid); });
        static_for<NOPS>([&](auto oc) __attribute__((always_inline)) { constexpr int op = decltype(oc)::value; RT0.load(op, srcT[op % 3], p.ld, J0, B, d, tid); });
        static_for<NOPS>([&](auto oc) __attribute__((always_inline)) { constexpr int op = decltype(oc)::value; R.load(op, srcM[op % 3], p.ld, J0, B, d, tid); });
        static_for<NOPS>([&](auto oc) __attribute__((always_inline)) { constexpr int op = decltype(oc)::value; RA.store(op, dstT[op % 3], I0, B, d, tid); });
        static_for<NOPS>([&](auto oc) __attribute__((always_inline)) { constexpr int op = decltype(oc)::value; RB.store(op, dstM[op % 3], I0, B, d, tid); });
    }
    if (tid < 2) cflag[tid] = 0;
    __syncthreads();
    float fa[KS], fb[KS];
    float fh[4 * NQH + KT];          // role 1: this wave's half of K of B[I] (wave 0: the first KQ / 2 quads; wave 1: the others + the tail)
    const int hq0 = (wid & 1) ? KQ / 2 : 0;
    if (wid < 2) {                   // role 1, wave i: F_i[I], tF_i[I], half i of B[I]
        load_frag<KS, LD>(fa, wid == 0 ? sF1 : sF2, l31, h);
        load_frag<KS, LD>(fb, wid == 0 ? stF1 : stF2, l31, h);
        const int nq = (wid & 1) ? NQH : KQ / 2;
#pragma unroll
        for (int j = 0; j < NQH; ++j) {
            const float4 v = *reinterpret_cast<const float4*>(sBm + l31 * LD + 8 * (hq0 + min(j, max(nq - 1, 0))) + 4 * h);
            const bool on = j < nq;
            fh[4 * j] = on ? v.x : 0.f; fh[4 * j + 1] = on ? v.y : 0.f; fh[4 * j + 2] = on ? v.z : 0.f; fh[4 * j + 3] = on ? v.w : 0.f;
        }
#pragma unroll
        for (int i = 0; i < KT; ++i) fh[4 * NQH + i] = (wid & 1) ? sBm[l31 * LD + 8 * KQ + 2 * i + h] : 0.f;
    } else {                         // role 2: B[I], tB[I]
        load_frag<KS, LD>(fa, sBm, l31, h);
        load_frag<KS, LD>(fb, stB, l31, h);
#pragma unroll
        for (int i = 0; i < 4 * NQH + KT; ++i) fh[i] = 0.f;
    }
    const int gcol = I0 + l31;                              // batch index of this lane's tile column
    const float gam_col = (gcol < B) ? p.discount[gcol] : 0.f;
    // the contraction coefficients are kept WITHOUT their common factor 1 / N_off (applied once at the end): on the diagonal
    // d loss / d M = -1/B = cdiag / N_off
    const float cdiag = (gcol < B) ? -(float)(B - 1) : 0.f;
    const float o2 = 2.0f * p.ortho2;
    __syncthreads();

    floatx16 out[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) out[nt][i] = 0.f;
    float s_sq = 0.f, s_diag = 0.f, s_all = 0.f, s_tall = 0.f, s_csq = 0.f, s_cdiag = 0.f;

    const float* sFi = (wid & 1) ? sF2 : sF1;               // role 2, wave 2 + i: F_i[J]
    float* my_x = xch + wid * 1024;                         // this wave's target product, read by the other wave of the role
    const float* their_x = xch + (wid ^ 1) * 1024;
    float* my_c = xch + 4096 + (wid & 1) * 1024;            // role 1: this wave's K-half of the covariance tile

    const int niter = (ntl + NG - 1) / NG;                  // group g takes the positions g, g + NG, ...; every group runs niter
                                                            // iterations (the barriers are workgroup-wide), a group without a
                                                            // tile in the last one only keeps the others company

    // Staging schedule.  The T panels of a tile are read by the target products only (phase A), the M panels by everything
    // after them (phase B): while phase A runs the M slots are free, while phase B runs the T slots are.  ONE register set R:
    //   phase A(tile): R (= M panels of this tile) -> M slots;  R <- T panels of the next tile
    //   phase B(tile): R (= T panels of the next tile) -> T slots;  R <- M panels of the next tile
    // each op riding behind one MFMA of a product.  Before the loop: T panels of the first tile in LDS, its M panels in R.
    static_for<NOPS>([&](auto oc) __attribute__((always_inline)) { constexpr int op = decltype(oc)::value; RT0.store(op, dstT[op % 3], tile_at(grp) * 32, B, d, tid); });

    // One loop per ROLE (the waves of a role run identical code: what differs between them is pointers and offsets), so the
    // accumulators of a role are updated in straight-line code.  Every barrier below is executed by all waves of the workgroup
    // the same number of times (the two roles have the same barrier sequence).
    auto run = [&](auto role_tag) __attribute__((always_inline)) {
        constexpr int ROLE = decltype(role_tag)::value;
        auto body = [&](int it, auto diag_tag) __attribute__((always_inline)) {
            constexpr bool DIAG = decltype(diag_tag)::value;
            const int k = it * NG + grp;
            const bool live = k < ntl;
            const int J0 = tile_at(k) * 32;
            const int Jn = tile_at(k + NG) * 32;                // (past the chunk's last tile: a clamped, unused tile)
            __syncthreads();                                    // T(tile) in LDS; the M slots and gamma are free
            floatx16 tm, G;
            if (live) {
                // ---- phase A: the wave's own target product (parked for the other wave of the role)
                if (tid < 32) sGam[tid] = (J0 + tid < B) ? p.discount[J0 + tid] : 0.f;
                auto side = [&](auto sc, auto nc) __attribute__((always_inline)) {     // slot sc of nc: its share of 2 NOPS sub-ops
                    constexpr int sidx = decltype(sc)::value, n = decltype(nc)::value;
                    constexpr int u0 = sidx * (2 * NOPS) / n, u1 = (sidx + 1) * (2 * NOPS) / n;
                    static_for<u1 - u0>([&](auto kc) __attribute__((always_inline)) {
                        constexpr int u = u0 + decltype(kc)::value, op = u >> 1;
                        if constexpr (u & 1) R.load(op, srcT[op % 3], p.ld, Jn, B, d, tid);
                        else R.store(op, dstM[op % 3], J0, B, d, tid);
                    });
                };
                // role 1: tile rows = t (J), cols = s (I):  T[r][c] = M_i[s = I0+c][t = J0+r]
                // role 2: tile rows = s (J), cols = t (I):  P[r][c] = M_i[s = J0+r][t = I0+c]
                tm = tile_mm<KS, LD>(ROLE == 1 ? stB : ((wid & 1) ? stF2 : stF1), fb, l31, h, side);
                put16(my_x, tm, lane);
            }
            __syncthreads();                                    // M(tile), gamma and the target products in LDS; the T slots are free
            if (live) {
                // ---- phase B.  Role 1 first computes its half of K of the covariance tile C[r][c] = B[J0+r] . B[I0+c]
                // (fb_ddpg.py:344-348) for role 2, parks it and raises its flag (LDS operations of a wave complete in order: the
                // flag is visible after the tile); role 2 needs it only at the end of the phase, ~100 MFMAs later, and checks
                // the flags there -- no third workgroup barrier, and the two roles carry 189 / 196 MFMAs per tile at d = 100.
                if constexpr (ROLE == 1) {
                    const floatx16 Ch = mm_steps<NQH, KT>(sBm + l31 * LD + 4 * h + 8 * hq0, sBm + l31 * LD + 8 * KQ + h, fh, NoSide());
                    put16(my_c, Ch, lane);
                    __hip_atomic_store(cflag + (wid & 1), it + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                auto side = [&](auto sc, auto nc) __attribute__((always_inline)) {
                    constexpr int sidx = decltype(sc)::value, n = decltype(nc)::value;
                    constexpr int u0 = sidx * (2 * NOPS) / n, u1 = (sidx + 1) * (2 * NOPS) / n;
                    static_for<u1 - u0>([&](auto kc) __attribute__((always_inline)) {
                        constexpr int u = u0 + decltype(kc)::value, op = u >> 1;
                        if constexpr (u & 1) R.load(op, srcM[op % 3], p.ld, Jn, B, d, tid);
                        else R.store(op, dstT[op % 3], Jn, B, d, tid);
                    });
                };
                G = tile_mm<KS, LD>(ROLE == 1 ? sBm : sFi, fa, l31, h, side);
                const bool dtile = DIAG && (J0 == I0);
                const floatx16 t2 = get16(their_x, lane);
                if constexpr (ROLE == 1) {
                    // discount is indexed by s = column here
                    auto gen = [&](int reg) __attribute__((always_inline)) -> float {
                        const float m = G[reg], t = fminf(tm[reg], t2[reg]);
                        const float dlt = m - gam_col * t;
                        s_all += m;
                        s_tall += t;
                        s_sq += dlt * dlt;
                        return dlt;
                    };
                    auto gen_d = [&](int reg) __attribute__((always_inline)) -> float {
                        const float m = G[reg], t = fminf(tm[reg], t2[reg]);
                        const float dlt = m - gam_col * t;
                        const bool diag = (J0 + acc_row(reg, h) == gcol);
                        s_all += m;
                        s_tall += t;
                        s_diag += diag ? m : 0.f;
                        s_sq += diag ? 0.f : dlt * dlt;
                        return diag ? cdiag : dlt;
                    };
                    // dF_i[I] += G^T-contract . B[J]
                    if (!dtile) contract_gen<NT, LD, 0, 16>(out, gen, sBm, l31, h);
                    else contract_gen<NT, LD, 0, 16>(out, gen_d, sBm, l31, h);
                } else {
                    float gj[16];                                              // discount of this lane's 16 tile rows (s = row here)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float4 u = *reinterpret_cast<const float4*>(sGam + 8 * i + 4 * h);
                        gj[4 * i] = u.x; gj[4 * i + 1] = u.y; gj[4 * i + 2] = u.z; gj[4 * i + 3] = u.w;
                    }
                    auto gen = [&](int reg) __attribute__((always_inline)) -> float {
                        return G[reg] - gj[reg] * fminf(tm[reg], t2[reg]);
                    };
                    auto gen_d = [&](int reg) __attribute__((always_inline)) -> float {
                        const float dlt = G[reg] - gj[reg] * fminf(tm[reg], t2[reg]);
                        return (J0 + acc_row(reg, h) == gcol) ? cdiag : dlt;
                    };
                    // dB[I] += G-contract . F_i[J]
                    if (!dtile) contract_gen<NT, LD, 0, 16>(out, gen, sFi, l31, h);
                    else contract_gen<NT, LD, 0, 16>(out, gen_d, sFi, l31, h);
                    // L_orth = mean_offdiag C^2 - 2 mean_diag C;  dL/dC = Hm = 2C/N_off (off-diag), -2/B (diag);
                    // C = B B^T  =>  dB = ortho * (Hm + Hm^T) B = ortho * 2 * Hm . B.  With ortho2 = 2*ortho the tile
                    // coefficient is ortho2 * Hm = 2 * (ortho2*C/N_off) resp. 2 * (-ortho2/B).  Wave 2 owns accumulator
                    // registers 0..7 = tile rows 0-3, 8-11 (+ 4 h) -- scalar sums and contraction --, wave 3 registers 8..15 =
                    // the same rows + 16; C = half of wave 0 + half of wave 1.
                    while (__hip_atomic_load(cflag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= it ||
                           __hip_atomic_load(cflag + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= it) {}
                    const int half = wid & 1;
                    const float* cA = xch + 4096 + half * 512, * cB = xch + 5120 + half * 512;
                    float cc[8];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const float4 x = *reinterpret_cast<const float4*>(cA + (i * 64 + lane) * 4);
                        const float4 y = *reinterpret_cast<const float4*>(cB + (i * 64 + lane) * 4);
                        cc[4 * i] = x.x + y.x; cc[4 * i + 1] = x.y + y.y; cc[4 * i + 2] = x.z + y.z; cc[4 * i + 3] = x.w + y.w;
                    }
                    auto genc = [&](int reg) __attribute__((always_inline)) -> float {
                        const float c = cc[reg];
                        s_csq += c * c;
                        return o2 * c;
                    };
                    auto genc_d = [&](int reg) __attribute__((always_inline)) -> float {
                        const float c = cc[reg];
                        const bool diag = (J0 + 16 * half + acc_row(reg, h) == gcol);
                        s_cdiag += diag ? c : 0.f;
                        s_csq += diag ? 0.f : c * c;
                        return diag ? o2 * cdiag : o2 * c;
                    };
                    if (!dtile) contract_gen<NT, LD, 0, 8>(out, genc, sBm + 16 * half * LD, l31, h);
                    else contract_gen<NT, LD, 0, 8>(out, genc_d, sBm + 16 * half * LD, l31, h);
                }
            }
        };
#pragma unroll 1
        for (int it = 0; it < niter - 1; ++it) body(it, std::false_type{});
        body(niter - 1, std::true_type{});
    };
    if (wid < 2) run(std::integral_constant<int, 1>{});
    else run(std::integral_constant<int, 2>{});
    // (``out`` still lacks the factor 1 / N_off: pairwise_reduce_kernel applies it -- no vector-ALU use of the accumulators here,
    //  or hipcc carries them through the loop in scattered VGPRs and copies 64 registers into MFMA tuples per tile)
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
    if (wid < 2) {
        const double a = wsum_d(s_sq), b = wsum_d(s_diag), c = wsum_d(s_all), e = wsum_d(s_tall);
        if (lane == 0) {
            put_hilo(sc, 4 * wid, a);
            put_hilo(sc, 4 * wid + 1, b);
            if (wid == 0) { put_hilo(sc, 2, c); put_hilo(sc, 3, e); }
        }
    } else {
        const double a = wsum_d(s_csq), b = wsum_d(s_cdiag);
        if (lane == 0) { put_hilo(sc, 6 + 2 * (wid - 2), a); put_hilo(sc, 7 + 2 * (wid - 2), b); }
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
    // the LAST workgroup has no rows: it folds the scalar partials (and advances the step state) beside the row work of the others
    const bool scalar_wg = blockIdx.x == gridDim.x - 1;
    if (adv != nullptr && scalar_wg && threadIdx.x == 255) step_advance_device(adv, adv_which);
    // one wavefront per row (d <= 128: two elements per lane); with ``y`` the backward of B = sqrt(d) y / |y|
    // (dy = (sqrt(d)/|y|)(dB - yhat (yhat . dB)), exactly l2norm_bwd_kernel) follows in the same registers
    if (!scalar_wg) {
        const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
        if (r < B) {
            float cc[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int n = lane + 64 * i;
                float a = 0.f, b = 0.f, c = 0.f;
                if (n < d) {
                    // eight chunks' partials in flight at once (clamped index, masked add; same summation order): written as a
                    // plain loop hipcc waits for every chunk's loads before it issues the next chunk's -- 2 x nchunks round trips
                    for (int c0 = 0; c0 < nchunks; c0 += 8) {
                        float xa[8], xb[8], xc[8], xd[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int ch = min(c0 + u, nchunks - 1);
                            const float* base = partial + ((size_t)ch * PW_SLOTS * Bp + r) * DP + n;
                            xa[u] = base[0];
                            xb[u] = base[(size_t)Bp * DP];
                            xc[u] = base[(size_t)2 * Bp * DP];
                            xd[u] = base[(size_t)3 * Bp * DP];
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const bool on = c0 + u < nchunks;
                            a += on ? xa[u] : 0.f;
                            b += on ? xb[u] : 0.f;
                            c += on ? xc[u] + xd[u] : 0.f;
                        }
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
    if (scalar_wg) {
        // wave w folds scalar slots w, w+4, w+8 over all workgroups -- each lane a strided slice in
        // fp64, then a fixed xor-shuffle tree (deterministic)
        __shared__ double tot[12];
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
        for (int slot = wid; slot < 10; slot += 4) {
            double s = 0.0;
            for (int bk = lane; bk < nblocks; bk += 64)
                s += (double)scal[(size_t)bk * PW_SCAL + slot] + (double)scal[(size_t)bk * PW_SCAL + 10 + slot];
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
#define PW_ATTR(KS) { PW_ATTR1(KS, 0, 1, bytes1) PW_ATTR1(KS, 1, 1, bytes1) PW_ATTR1(KS, 2, 1, bytes1) if (pl.ld <= 68) { PW_ATTR1(KS, 0, 2, bytes2) PW_ATTR1(KS, 1, 2, bytes2) PW_ATTR1(KS, 2, 2, bytes2) } return hipSuccess; }
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
    if (a.vec && (B & 31) == 0 && d == 2 * pl.ks) a.vec = 2;
    a.partial = scratch;
    a.scal = scratch + (size_t)pl.nchunks * PW_SLOTS * pl.Bp * pl.dp;
    a.Bp = pl.Bp;
    a.i_off = row_off;
    dim3 grid(pl.nI, pl.nchunks), block(256 * pl.ng);
    hipError_t e = hipSuccess;
#define PW_LAUNCH1(KS, NG_)                                                                                            \
    if (a.vec == 2) hipLaunchKernelGGL((pairwise_kernel<KS, 2, NG_>), grid, block, pl.lds_bytes, s, a);                \
    else if (a.vec == 1) hipLaunchKernelGGL((pairwise_kernel<KS, 1, NG_>), grid, block, pl.lds_bytes, s, a);           \
    else hipLaunchKernelGGL((pairwise_kernel<KS, 0, NG_>), grid, block, pl.lds_bytes, s, a)
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
    hipLaunchKernelGGL(pairwise_reduce_kernel, dim3((rows + 3) / 4 + 1), dim3(256), 0, s, a.partial, a.scal,
                       pl.nchunks, pl.nchunks * pl.nI, rows, pl.Bp, d, pl.dp, ld, B, ortho_coef, dF1, dF2, dB, metrics, adv,
                       adv_which, y, norms, dy, out_scale / ((float)B * (float)(B - 1)));
    return hipGetLastError();
}

}  // namespace fbhip
