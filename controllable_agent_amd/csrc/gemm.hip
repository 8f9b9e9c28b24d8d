// Grouped fp32 GEMM for gfx950 on v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD).
//
// Covers every dense contraction of the FB-DDPG step:
//   forward   Y[M,N]  = X[M,K]  . W[N,K]^T (+bias, relu)      a_kcontig=1 b_kcontig=1   (nn.Linear, fb_modules.py:60-78)
//   dgrad     dX[M,K] = dY[M,N] . W[N,K]   (* relu mask)       a_kcontig=1 b_kcontig=0
//   wgrad     dW[N,K] = dY[M,N]^T . X[M,K] (+ bias grad)       a_kcontig=0 b_kcontig=0
// One launch runs up to MAX_GROUP independent problems (e.g. the two trunks of a ForwardMap, the F1/F2
// heads), so small layers share the 256 CUs instead of serialising.
//
// Structure: 256 threads = 4 waves; each wave owns ONE 32x32 accumulator (16 VGPRs) -- the 32x32x2 f32 MFMA
// has issue interval == dependent latency == 64 cycles, so a single accumulator chain per SIMD already runs the
// matrix pipe back-to-back.  Waves are arranged WM x WN x WK: WK > 1 splits the K range inside the workgroup
// (partial tiles are reduced through LDS) so skinny outputs (N = z_dim, N = action_dim, N = feature_dim) still
// put four waves on every CU.  Operand tiles are staged global -> registers -> LDS as [k][row] (row stride
// odd => conflict-free ds_read_b32 for both operands), double-buffered with one barrier per K step; the global
// loads of step t+1 are issued before the MFMAs of step t.
#include "common.h"

#include <cstdlib>
#include <mutex>
#include <type_traits>

namespace fbhip {

template <int ROWS, int BKT, int Q>
__device__ __forceinline__ void load_tile(float4 (&v)[Q], const float* __restrict__ src, int ld, int kcontig,
                                          int vec, int row0, int nrows, int k0, int K, int tid) {
    if (kcontig) {
        constexpr int QK = BKT / 4;
#pragma unroll
        for (int i = 0; i < Q; ++i) {
            const int q = tid + i * 256;
            const int r = q / QK, kq = q % QK;
            const int gr = row0 + r, gk = k0 + 4 * kq;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gr < nrows && gk < K) {
                const float* ptr = src + (size_t)gr * ld + gk;
                if (vec && gk + 3 < K) {
                    x = *reinterpret_cast<const float4*>(ptr);
                } else {
                    x.x = ptr[0];
                    if (gk + 1 < K) x.y = ptr[1];
                    if (gk + 2 < K) x.z = ptr[2];
                    if (gk + 3 < K) x.w = ptr[3];
                }
            }
            v[i] = x;
        }
    } else {
        constexpr int QR = ROWS / 4;
#pragma unroll
        for (int i = 0; i < Q; ++i) {
            const int q = tid + i * 256;
            const int k = q / QR, rq = q % QR;
            const int gk = k0 + k, gr = row0 + 4 * rq;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gk < K && gr < nrows) {
                const float* ptr = src + (size_t)gk * ld + gr;
                if (vec && gr + 3 < nrows) {
                    x = *reinterpret_cast<const float4*>(ptr);
                } else {
                    x.x = ptr[0];
                    if (gr + 1 < nrows) x.y = ptr[1];
                    if (gr + 2 < nrows) x.z = ptr[2];
                    if (gr + 3 < nrows) x.w = ptr[3];
                }
            }
            v[i] = x;
        }
    }
}

// LDS image is [k][row] for both operands.  A K-contiguous source quad (4 consecutive k of one row) is written
// transposed (stride LD), a K-strided quad (4 consecutive rows of one k) is written contiguously (stride 1).
__device__ __forceinline__ void store_quad(float* __restrict__ d, int step, const float4& v) {
    d[0] = v.x;
    d[step] = v.y;
    d[2 * step] = v.z;
    d[3 * step] = v.w;
}

// accumulator tile -> C (or the split-K partial slab) with the fused epilogue; acc row = (r&3) + 8(r>>2) + 4h, col = l31
__device__ __forceinline__ void gemm_epilogue(const GemmProblem& p, const floatx16& acc, float csum, int slice, int row0,
                                              int col0, int wm, int wn, int tn, int l31, int h) {
    const int M = p.M, N = p.N;
    if (p.kslices > 1) {
        // raw partial tile (+ partial column sums behind the tiles); splitk_reduce_kernel applies the epilogue
        float* part = p.partial + (size_t)slice * M * N;
        if (p.colsum != nullptr && tn == 0 && wn == 0) {
            const float tot = csum + __shfl_xor(csum, 32);
            const int r = row0 + wm * 32 + l31;
            if (h == 0 && r < M) p.partial[(size_t)p.kslices * M * N + (size_t)slice * M + r] = tot;
        }
        const int col = col0 + wn * 32 + l31;
        if (col >= N) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < M) part[(size_t)row * N + col] = acc[r];
        }
        return;
    }

    if (p.colsum != nullptr && tn == 0 && wn == 0) {
        const float tot = csum + __shfl_xor(csum, 32);
        const int r = row0 + wm * 32 + l31;
        if (h == 0 && r < M) p.colsum[r] = tot;
    }

    const int col = col0 + wn * 32 + l31;
    if (col >= N) return;
    const int epi = p.epi;
    const float bias = (epi == EPI_BIAS || epi == EPI_BIAS_RELU) ? p.bias[col] : 0.f;
    float* __restrict__ C = p.C;
    // the sixteen aux values of this lane FIRST (clamped rows), then the stores: read inside the store loop every row waits for
    // its own load AND for the previous row's store (vmcnt counts both) -- sixteen dependent round trips per tile
    float ax[16];
    if (epi == EPI_MASK_RELU || epi == EPI_TANH_BWD) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            ax[r] = p.aux[(size_t)min(row, M - 1) * p.ldaux + col];
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) ax[r] = 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < M) {
            float v = acc[r];
            if (epi == EPI_BIAS) {
                v += bias;
            } else if (epi == EPI_BIAS_RELU) {
                v = fmaxf(v + bias, 0.f);
            } else if (epi == EPI_MASK_RELU) {
                v = ax[r] > 0.f ? v : 0.f;
            } else if (epi == EPI_TANH_BWD) {
                v = v * (1.f - ax[r] * ax[r]);
            }
            C[(size_t)row * p.ldc + col] = v;
        }
    }
}

// -DFBHIP_TRACE (tools/gemm_trace.hip only): cycle stamps of one consumer and one producer wave per workgroup, kept in LDS
// behind the stages and dumped at the end -- where a chunk's time goes (MFMA span, barrier waits, load landing)
#ifdef FBHIP_TRACE
constexpr int TRACE_IT = 40, TRACE_SLOTS = 4, TRACE_WGS = 2048;
__device__ unsigned long long g_trace[TRACE_WGS * 2 * TRACE_IT * TRACE_SLOTS];
__device__ unsigned g_trace_hw[TRACE_WGS];
#define FBHIP_TR(role, it, slot)                                                                                      \
    do {                                                                                                              \
        if (lane == 0 && (it) < TRACE_IT)                                                                             \
            reinterpret_cast<unsigned long long*>(smem + 2 * STAGE)[((role) * TRACE_IT + (it)) * TRACE_SLOTS + (slot)] = \
                __builtin_readcyclecounter();                                                                         \
    } while (0)
#else
#define FBHIP_TR(role, it, slot) do {} while (0)
#endif

// Wave-specialised workgroup of 8 waves: waves 0-3 are CONSUMERS (one 32x32 accumulator each, arranged WM x WN x WK;
// they only read MFMA fragments from LDS and issue the dependent MFMA chain), waves 4-7 are PRODUCERS (global ->
// registers -> LDS staging of the next K chunks, prefetch distance 2).  One s_barrier per K chunk couples them.  With
// symmetric waves the co-resident workgroups fall into lockstep and the matrix pipe idles through every
// store/barrier/refill phase (PMC: 44-52 % MFMA busy); here the pipe-owning waves have nothing else to do.
// (A deeper producer prefetch -- two chunks in flight behind the landing one -- was built and measured in round 1: ISA as intended,
// 9 % slower on the step: the staging path is not latency-bound.  Removed in round 2; DESIGN.md section 9 keeps the record.)
// CSUM: some problem of the launch wants the column sums of its A operand (bias gradients).  A template parameter, and the
// consumers' LDS addresses are loop invariants of a loop unrolled over the two stages: a wave's own vector-ALU instructions do not
// overlap its MFMAs (tools/mfma_shadow.hip: +5..8 cycles each), and the consumer chunk used to carry 16 adds for the sums and 14
// address adds per 16 MFMAs.
template <int WM, int WN, int WK, int BK, bool CSUM>
__global__ void __launch_bounds__(512) gemm_kernel(const GemmGroup g) {
    constexpr int BM = 32 * WM, BN = 32 * WN, BKT = BK * WK;
    constexpr int LDA_S = BM + 1, LDB_S = BN + 1;
    constexpr int STAGE = BKT * (LDA_S + LDB_S);
    constexpr int QA = BM * BKT / 1024, QB = BN * BKT / 1024;
    constexpr int NF = BK / 2;                         // MFMA k-steps per chunk
    static_assert(WM * WN * WK == 4, "four consumer waves per workgroup");
    static_assert((BM * BKT) % 1024 == 0 && (BN * BKT) % 1024 == 0, "tile must split into float4 per thread");
    static_assert((WK - 1) * WM * WN * 17 * 64 <= 2 * STAGE, "split-K reduction scratch must fit");
    extern __shared__ __attribute__((aligned(16))) float smem[];      // 2 * STAGE floats

    // problem lookup by launch position, then an XCD-aware bijective remap INSIDE the problem: workgroup b runs on
    // XCD b % 8; each XCD gets a contiguous run of the problem's tiles (neighbours share an A row-panel / adjacent
    // B panels in that XCD's L2) while every problem of a heterogeneous group still spreads over all 8 XCDs.
    const int orig = blockIdx.x;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < MAX_GROUP; ++i)
        if (i < g.n && orig >= g.p[i].tile_start) pi = i;
    const GemmProblem& p = g.p[pi];
    const int nwg = p.tiles_m * p.tiles_n * p.kslices;
    const int jb = orig - p.tile_start;
    const int xcd = jb & 7, qd = nwg >> 3, rm = nwg & 7;
    const int t = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (jb >> 3);
    const int tiles_mn = p.tiles_m * p.tiles_n;
    const int slice = t / tiles_mn, tt = t % tiles_mn;        // slice-major: neighbours share operand panels
    int tn = tt % p.tiles_n, tm = tt / p.tiles_n;
    if (p.xgn > 0) {
        // 2-D ownership: XCD x owns block (x / xgn, x % xgn) of an xgm x xgn grid of tile blocks, so its L2 holds 1 / xgm of the A
        // rows and 1 / xgn of the B rows instead of a band of A and ALL of B (the fabric fetch of a 1024 x 2048 output: 2.1x less)
        const int bn = p.tiles_n / p.xgn, bm = p.tiles_m / p.xgm, j = jb >> 3;
        tm = (xcd / p.xgn) * bm + j / bn;
        tn = (xcd % p.xgn) * bn + j % bn;
    }
    const int M = p.M, N = p.N;
    const int kb = slice * p.kper * BKT;                       // this workgroup's K range [kb, K)
    const int K = min(p.K, kb + p.kper * BKT);
    const int row0 = tm * BM, col0 = tn * BN;
    const int nt = (K - kb + BKT - 1) / BKT;                  // chunks of this slice

    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;

    if (wid >= 4) {
        // =========================================================================================== PRODUCERS
        const int tid = threadIdx.x - 256;
        const float* __restrict__ A = p.A;
        const float* __restrict__ Bp = p.B;
        const int akc = p.a_kcontig, bkc = p.b_kcontig;
        // per-thread staging geometry: global element offset at chunk 0, LDS float offset, LDS write stride
        size_t ga[QA], gb[QB];
        int sa[QA], sb[QB];
#pragma unroll
        for (int i = 0; i < QA; ++i) {
            const int q = tid + i * 256;
            // rows past the matrix edge read a CLAMPED (valid) address: whatever lands in those LDS rows only reaches
            // output rows / columns the epilogue drops, so ragged tiles of aligned operands keep the branch-free loader
            if (akc) {
                const int r = q / (BKT / 4), kq = q % (BKT / 4);
                ga[i] = (size_t)min(row0 + r, M - 1) * p.lda + kb + 4 * kq;
                sa[i] = (4 * kq) * LDA_S + r;
            } else {
                const int k = q / (BM / 4), rq = q % (BM / 4);
                ga[i] = (size_t)(kb + k) * p.lda + min(row0 + 4 * rq, p.lda - 4);
                sa[i] = k * LDA_S + 4 * rq;
            }
        }
#pragma unroll
        for (int i = 0; i < QB; ++i) {
            const int q = tid + i * 256;
            if (bkc) {
                const int r = q / (BKT / 4), kq = q % (BKT / 4);
                gb[i] = (size_t)min(col0 + r, N - 1) * p.ldb + kb + 4 * kq;
                sb[i] = (4 * kq) * LDB_S + r;
            } else {
                const int k = q / (BN / 4), rq = q % (BN / 4);
                gb[i] = (size_t)(kb + k) * p.ldb + min(col0 + 4 * rq, p.ldb - 4);
                sb[i] = k * LDB_S + 4 * rq;
            }
        }
        const int step_a = akc ? LDA_S : 1, step_b = bkc ? LDB_S : 1;
        const size_t adv_a = akc ? (size_t)BKT : (size_t)BKT * p.lda;
        const size_t adv_b = bkc ? (size_t)BKT : (size_t)BKT * p.ldb;
        // 16-byte-aligned operands (lda, ldb multiples of 4, >= 4) take the branch-free loader for every full K chunk --
        // edge tiles included (clamped rows above); unaligned views and the K tail go through the predicated loader
        // (same register image)
        const bool interior = p.a_vec && p.b_vec && p.lda >= 4 && p.ldb >= 4;
        const int nfast = interior ? (K - kb) / BKT : 0;

        auto load_fast = [&](int ch, float4 (&xa)[QA], float4 (&xb)[QB]) __attribute__((always_inline)) {
            const size_t oa = (size_t)ch * adv_a, ob = (size_t)ch * adv_b;
#pragma unroll
            for (int i = 0; i < QA; ++i) xa[i] = *reinterpret_cast<const float4*>(A + ga[i] + oa);
#pragma unroll
            for (int i = 0; i < QB; ++i) xb[i] = *reinterpret_cast<const float4*>(Bp + gb[i] + ob);
        };
        auto load_slow = [&](int ch, float4 (&xa)[QA], float4 (&xb)[QB]) __attribute__((always_inline)) {
            load_tile<BM, BKT, QA>(xa, A, p.lda, akc, p.a_vec, row0, M, kb + ch * BKT, K, tid);
            load_tile<BN, BKT, QB>(xb, Bp, p.ldb, bkc, p.b_vec, col0, N, kb + ch * BKT, K, tid);
        };
        auto store_chunk = [&](int stage, const float4 (&xa)[QA], const float4 (&xb)[QB]) __attribute__((always_inline)) {
            float* dA = smem + stage * STAGE;
#pragma unroll
            for (int i = 0; i < QA; ++i) store_quad(dA + sa[i], step_a, xa[i]);
#pragma unroll
            for (int i = 0; i < QB; ++i) store_quad(dA + BKT * LDA_S + sb[i], step_b, xb[i]);
        };
        // producer step ``it``: request chunk it+2 into (la_, lb_), land chunk it+1 (in flight in (sa_, sb_)) in LDS
        // stage (it+1)&1, then meet the consumers at the barrier.  LOAD/STORE are compile-time so the steady state is
        // straight-line code and hipcc emits a COUNTED s_waitcnt vmcnt(n) that keeps the newest chunk in flight.
        auto p_step = [&](int it, auto do_load, auto do_store, float4 (&la_)[QA], float4 (&lb_)[QB],
                          const float4 (&sa_)[QA], const float4 (&sb_)[QB]) __attribute__((always_inline)) {
            if (wid == 4) FBHIP_TR(1, it, 0);
            if constexpr (decltype(do_load)::value) load_fast(it + 2, la_, lb_);
            if (wid == 4) FBHIP_TR(1, it, 1);
            if constexpr (decltype(do_store)::value) store_chunk((it + 1) & 1, sa_, sb_);
            if (wid == 4) FBHIP_TR(1, it, 2);
            __syncthreads();
            if (wid == 4) FBHIP_TR(1, it, 3);
        };
        constexpr std::true_type T{};
        constexpr std::false_type F{};
        float4 ra0[QA], rb0[QB], ra1[QA], rb1[QB];
        if (nfast > 0) {
            const int nf = nfast;
            load_fast(0, ra0, rb0);
            if (nf > 1) load_fast(1, ra1, rb1);
            else {
#pragma unroll
                for (int i = 0; i < QA; ++i) ra1[i] = ra0[i];
#pragma unroll
                for (int i = 0; i < QB; ++i) rb1[i] = rb0[i];
            }
            store_chunk(0, ra0, rb0);
            __syncthreads();                                          // barrier #0: chunk 0 visible
            int it = 0;
            for (; it + 3 < nf; it += 2) {
                p_step(it, T, T, ra0, rb0, ra1, rb1);
                p_step(it + 1, T, T, ra1, rb1, ra0, rb0);
            }
            const int rem = nf - it;                                    // 1, 2 or 3 fast steps left
            const bool tail = nt > nf;                                  // ragged K tail chunk (index nf)
            if (rem == 3) {
                p_step(it, T, T, ra0, rb0, ra1, rb1);
                p_step(it + 1, F, T, ra1, rb1, ra0, rb0);
                if (tail) { load_slow(nf, ra1, rb1); p_step(it + 2, F, T, ra0, rb0, ra1, rb1); }
                else p_step(it + 2, F, F, ra0, rb0, ra1, rb1);
            } else if (rem == 2) {
                p_step(it, F, T, ra0, rb0, ra1, rb1);
                if (tail) { load_slow(nf, ra0, rb0); p_step(it + 1, F, T, ra1, rb1, ra0, rb0); }
                else p_step(it + 1, F, F, ra1, rb1, ra0, rb0);
            } else {
                if (tail) { load_slow(nf, ra1, rb1); p_step(it, F, T, ra0, rb0, ra1, rb1); }
                else p_step(it, F, F, ra0, rb0, ra1, rb1);
            }
            if (tail) __syncthreads();                                  // the consumers' barrier after chunk nf
        } else {
            load_slow(0, ra0, rb0);
            store_chunk(0, ra0, rb0);
            __syncthreads();
            for (int it = 0; it < nt; ++it) {
                if (it + 1 < nt) {
                    load_slow(it + 1, ra0, rb0);
                    store_chunk((it + 1) & 1, ra0, rb0);
                }
                __syncthreads();
            }
        }
        if constexpr (WK > 1) __syncthreads();                          // stay for the consumers' reduction barrier
#ifdef FBHIP_TRACE
        __syncthreads();
#endif
        return;
    }

    // ============================================================================================== CONSUMERS
    const int l31 = lane & 31, h = lane >> 5;
    const int wk = wid / (WM * WN), wrem = wid % (WM * WN);
    const int wm = wrem / WN, wn = wrem % WN;
    const int frag_a = (wk * BK + h) * LDA_S + wm * 32 + l31;
    const int frag_b = BKT * LDA_S + (wk * BK + h) * LDB_S + wn * 32 + l31;

    floatx16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float csum = 0.f;

    __syncthreads();                                                    // barrier #0
    auto chunk = [&](int it, const float* __restrict__ fa_, const float* __restrict__ fb_) __attribute__((always_inline)) {
        if (wid == 0) FBHIP_TR(0, it, 0);
        float av[NF], bv[NF];
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            av[j] = fa_[2 * j * LDA_S];
            bv[j] = fb_[2 * j * LDB_S];
        }
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc, 0, 0, 0);
            if constexpr (CSUM) csum += av[j];
        }
        // Pin the issue order: fragment reads run ahead of the MFMA pair that consumes them, so each s_waitcnt only
        // covers reads issued >= 128 MFMA-cycles earlier (ds_read_b32 pairs are merged into ds_read2_b32).
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);              // DS reads for MFMA pairs 0, 1
#pragma unroll
        for (int j = 0; j < NF / 2 - 2; ++j) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);          // MFMA pair j
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);          // DS reads for pair j + 2
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);              // last two MFMA pairs
        if (wid == 0) FBHIP_TR(0, it, 1);
        __syncthreads();
        if (wid == 0) FBHIP_TR(0, it, 2);
    };
    {
        // (accumulating the sums only in the waves whose epilogue stores them -- a second copy of the loop behind a uniform
        //  branch -- measured 1168 against 1183 update-steps/s: not done)
        const float* a0 = smem + frag_a, * b0 = smem + frag_b, * a1 = smem + STAGE + frag_a, * b1 = smem + STAGE + frag_b;
        int it = 0;
        for (; it + 1 < nt; it += 2) {
            chunk(it, a0, b0);
            chunk(it + 1, a1, b1);
        }
        if (it < nt) chunk(it, a0, b0);
    }
#ifdef FBHIP_TRACE
    __syncthreads();
    if (orig < TRACE_WGS) {
        const unsigned long long* tl = reinterpret_cast<const unsigned long long*>(smem + 2 * STAGE);
        for (int i = threadIdx.x; i < 2 * TRACE_IT * TRACE_SLOTS; i += 256) g_trace[(size_t)orig * 2 * TRACE_IT * TRACE_SLOTS + i] = tl[i];
        if (threadIdx.x == 0) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            g_trace_hw[orig] = (hw & 0xffffff) | (xcc << 24);
        }
    }
#endif

    if constexpr (WK > 1) {
        // reduce the WK partial tiles (and the partial column sums) through LDS; wave group wk == 0 finishes
        float* red = smem;
        if (wk > 0) {
            float* dst = red + ((wk - 1) * (WM * WN) + wrem) * (17 * 64) + lane;
#pragma unroll
            for (int i = 0; i < 16; ++i) dst[i * 64] = acc[i];
            dst[16 * 64] = csum;
        }
        __syncthreads();
        if (wk > 0) return;
#pragma unroll
        for (int j = 0; j < WK - 1; ++j) {
            const float* srcp = red + (j * (WM * WN) + wrem) * (17 * 64) + lane;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] += srcp[i * 64];
            csum += srcp[16 * 64];
        }
    }

    gemm_epilogue(p, acc, csum, slice, row0, col0, wm, wn, tn, l31, h);
}

// ---------------------------------------------------------------------------------------------------------------
// LDS-DMA kernels: (64 TM) x 64 x 32 tiles, TM = 1 or 2 accumulators per consumer wave.
//
// Knock-out measurements on the register-staged kernel above and on this one (tools/gemm_diag.py, 4096^3, MI355X):
// without any MFMA the staging pipeline alone (global -> LDS of every chunk + fragment reads + barriers) needs 625 us
// (registers) / 893 us (DMA) against 874 us of pure matrix-pipe time, and it does not get faster when every chunk reads
// the same (cache-resident) addresses: 64x64 tiles are bound by the per-CU global->LDS path (~16 B/clk/CU), not by MFMA
// issue, LDS bandwidth, barriers or L2 misses.  The lever is bytes per FLOP: the 128x64 tile moves 25 % fewer, with
// two INDEPENDENT accumulator chains per wave so one wave per SIMD keeps the matrix pipe busy.
//
// Producer waves move global -> LDS with global_load_lds_dwordx4 (1 KiB per wave-instruction, no VGPR round trip, no
// ds_write pass).  The DMA writes lane-linear (M0 base + 16 B * lane), so the LDS image is chosen by the SOURCE
// address of each lane:
//   k-contiguous operand (X[M,K], W[N,K]):  piece = 8 rows x 32 k = 8 FULL 128-byte lines (fragment-shaped
//       16-row x 64-byte pieces request every line twice and halve the TA/L2 rate): lane l fetches row l%8 and the
//       16-byte quad (l/8) ^ (piece & 1) of it, so granule position pos of row r holds quad pos ^ ((r>>3) & 1).  A
//       fragment is ONE ds_read_b128 per lane = 4 consecutive k of its row; with the XOR the 16 rows of a hardware
//       b128 lane group hit 16 distinct granule slots (conflict-free, 256 B/clk).  Linear destination, swizzled SOURCE,
//       same involution on the read.
//   k-strided operand (dY^T, X^T of wgrad): piece = 4 k x 64 rows: lane l fetches k l/16, rows 4(l%16)..+3
//       -> [k][row] rows of 64 floats, fragments by ds_read_b32 (consecutive lanes = consecutive dwords).
// MFMA (t, m), t = 0..3, m = 0..3 of a chunk contracts k = 8t + 4h + m (h = lane >> 5) for both operands.
// A ring of S chunk buffers with ONE barrier per chunk: at barrier i the producers have waited (counted vmcnt) for
// chunk i+1 and the consumers have retired their fragment reads of chunk i, whose buffer then takes chunk i + S; the
// consumers read the fragments of chunk i+1 while the MFMAs of chunk i run (two register sets), so the MFMA chains
// restart right after each barrier.  Requires K % 32 == 0 per slice and 16-byte aligned operands; ragged M / N tiles
// read clamped (valid) rows whose results the epilogue drops.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

__device__ __forceinline__ void glds16(const float* src, float* lds_dst) {
    __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)lds_dst, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until at most ``chunks`` chunks (P DMAs each) are still in flight
template <int P, int MAXC>
__device__ __forceinline__ void wait_chunks(int chunks) {
    static_assert(MAXC >= 1 && MAXC <= 3 && P * MAXC < 64, "vmcnt is a 6-bit immediate");
    if (chunks >= 3 && MAXC >= 3) wait_vmcnt<P * (MAXC >= 3 ? 3 : 0)>();
    else if (chunks >= 2 && MAXC >= 2) wait_vmcnt<P * (MAXC >= 2 ? 2 : 0)>();
    else if (chunks >= 1) wait_vmcnt<P>();
    else wait_vmcnt<0>();
}

template <int TM> struct DmaGeom {
    static constexpr int BM = 64 * TM, BN = 64, BKT = 32;
    static constexpr int S = 3;                                // ring depth (chunks)
    static constexpr int NPW = 4;                              // producer waves (a DMA costs its wave 60-180 issue cycles)
    static constexpr int STAGE = (BM + BN) * BKT;              // floats per chunk buffer: A tile then B tile
    static constexpr int BOFF = BM * BKT;
    static constexpr int PIECES = (BM + BN) / 8;               // 1 KiB DMA pieces per chunk
    static constexpr int P = PIECES / NPW;                     // per producer wave
    static constexpr size_t LDS_BYTES = (size_t)S * STAGE * sizeof(float);
};

// fragment offsets (floats) inside an operand tile of R rows
template <bool KC, int R>
__device__ __forceinline__ int dma_frag_off(int row, int h) {
    return KC ? ((row >> 3) * 256 + (row & 7) * 4 + ((h ^ ((row >> 3) & 1)) * 32))
              : ((h * (R / 64) + (row >> 6)) * 256 + (row & 63));
}
template <bool KC, int R>
__device__ __forceinline__ void dma_read_frag(const float* __restrict__ op, int off, float (&f)[4][4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if constexpr (KC) {
            const float4 v = *reinterpret_cast<const float4*>(op + off + t * 64);
            f[t][0] = v.x; f[t][1] = v.y; f[t][2] = v.z; f[t][3] = v.w;
        } else {
#pragma unroll
            for (int m = 0; m < 4; ++m) f[t][m] = op[off + t * 512 * (R / 64) + m * 64];
        }
    }
}

template <int TM, bool AKC, bool BKC>
__device__ __forceinline__ void dma_consume(const float* __restrict__ smem, int nt, int lane, int wm, int wn,
                                            floatx16 (&acc)[TM], float (&csum)[TM]) {
    using G = DmaGeom<TM>;
    const int l31 = lane & 31, h = lane >> 5;
    int offa[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) offa[i] = dma_frag_off<AKC, G::BM>(wm * 32 * TM + 32 * i + l31, h);
    const int offb = G::BOFF + dma_frag_off<BKC, G::BN>(wn * 32 + l31, h);
    float a0[TM][4][4], b0[4][4];
    auto read = [&](int c, float (&a)[TM][4][4], float (&b)[4][4]) __attribute__((always_inline)) {
        const float* st = smem + (c % G::S) * G::STAGE;
#pragma unroll
        for (int i = 0; i < TM; ++i) dma_read_frag<AKC, G::BM>(st, offa[i], a[i]);
        dma_read_frag<BKC, G::BN>(st, offb, b);
    };
    auto mfma = [&](const float (&a)[TM][4][4], const float (&b)[4][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t][m], b[t][m], acc[i], 0, 0, 0);
                    csum[i] += a[i][t][m];
                }
    };
    __syncthreads();                                            // barrier "init": chunk 0 has landed
    // two independent accumulator chains per wave and a second workgroup on the CU cover the fragment-read latency;
    // one register set keeps the kernel at two workgroups per CU
    for (int it = 0; it < nt; ++it) {
        read(it, a0, b0);
        mfma(a0, b0);
        __syncthreads();                                        // barrier it: reads of chunk it retired
    }
}

template <int TM>
__global__ void __launch_bounds__(256 + 64 * DmaGeom<TM>::NPW) gemm_dma_kernel(const GemmGroup g) {
    using G = DmaGeom<TM>;
    constexpr int BM = G::BM, BN = G::BN, BKT = G::BKT;
    extern __shared__ __attribute__((aligned(16))) float smem[];      // S * STAGE floats

    const int orig = blockIdx.x;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < MAX_GROUP; ++i)
        if (i < g.n && orig >= g.p[i].tile_start) pi = i;
    const GemmProblem& p = g.p[pi];
    const int nwg = p.tiles_m * p.tiles_n * p.kslices;
    const int jb = orig - p.tile_start;
    const int xcd = jb & 7, qd = nwg >> 3, rm = nwg & 7;
    const int t = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (jb >> 3);
    const int tiles_mn = p.tiles_m * p.tiles_n;
    const int slice = t / tiles_mn, tt = t % tiles_mn;
    int tn = tt % p.tiles_n, tm = tt / p.tiles_n;
    if (p.xgn > 0) {                                            // 2-D XCD ownership, see gemm_kernel
        const int bn = p.tiles_n / p.xgn, bm = p.tiles_m / p.xgm, j = jb >> 3;
        tm = (xcd / p.xgn) * bm + j / bn;
        tn = (xcd % p.xgn) * bn + j % bn;
    }
    const int M = p.M, N = p.N;
    const int kb = slice * p.kper * BKT;
    const int K = min(p.K, kb + p.kper * BKT);
    const int row0 = tm * BM, col0 = tn * BN;
    const int nt = (K - kb) / BKT;

    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;

    if (wid >= 4) {
        // =========================================================================================== PRODUCERS
        const int w = wid - 4;                               // pieces w*P .. w*P + P-1 (A pieces first, then B)
        const float* src[G::P];
        size_t adv[G::P];
#pragma unroll
        for (int q = 0; q < G::P; ++q) {
            const int pc = w * G::P + q;
            const bool isA = pc < BM / 8;
            const int po = isA ? pc : pc - BM / 8;           // piece index inside its operand tile
            const float* base = isA ? p.A : p.B;
            const int ld = isA ? p.lda : p.ldb, kc = isA ? p.a_kcontig : p.b_kcontig;
            const int r0 = isA ? row0 : col0, nr = isA ? M : N, rpk = isA ? BM / 64 : BN / 64;
            if (kc) {
                const int r = min(r0 + po * 8 + (lane & 7), nr - 1);
                src[q] = base + (size_t)r * ld + kb + (((lane >> 3) ^ (po & 1)) * 4);
                adv[q] = (size_t)BKT;
            } else {
                const int r = min(r0 + (po % rpk) * 64 + 4 * (lane & 15), (nr - 1) & ~3);
                src[q] = base + (size_t)(kb + (po / rpk) * 4 + (lane >> 4)) * ld + r;
                adv[q] = (size_t)BKT * ld;
            }
        }
        float* const dst = smem + (w * G::P) * 256;           // pieces are 256 floats, A tile then B tile
        auto issue = [&](int c) __attribute__((always_inline)) {
            float* d = dst + (c % G::S) * G::STAGE;
#pragma unroll
            for (int q = 0; q < G::P; ++q) glds16(src[q] + (size_t)c * adv[q], d + q * 256);
        };
        for (int c = 0; c < G::S && c < nt; ++c) issue(c);
        // barrier "init": chunk 0 landed (up to S - 1 later chunks stay in flight)
        wait_chunks<G::P, G::S - 1>(nt - 1);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        for (int it = 0; it < nt; ++it) {
            // barrier it: chunk it+1 must have landed; chunks it+2 .. it+S-1 (those already issued) stay in flight
            wait_chunks<G::P, G::S - 2>(nt - 2 - it);
            __builtin_amdgcn_s_barrier();                       // raw: a fence would drain the DMAs
            asm volatile("" ::: "memory");
            // the consumers retired their reads of chunk ``it`` before this barrier: its buffer takes chunk it + S
            if (it + G::S < nt) issue(it + G::S);
        }
        return;
    }

    // ============================================================================================== CONSUMERS
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;
    floatx16 acc[TM];
    float csum[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        csum[i] = 0.f;
    }
    const int mode = p.a_kcontig * 2 + p.b_kcontig;
    if (mode == 3) dma_consume<TM, true, true>(smem, nt, lane, wm, wn, acc, csum);
    else if (mode == 2) dma_consume<TM, true, false>(smem, nt, lane, wm, wn, acc, csum);
    else if (mode == 0) dma_consume<TM, false, false>(smem, nt, lane, wm, wn, acc, csum);
    else dma_consume<TM, false, true>(smem, nt, lane, wm, wn, acc, csum);

#pragma unroll
    for (int i = 0; i < TM; ++i)
        gemm_epilogue(p, acc[i], csum[i], slice, row0 + wm * 32 * TM + 32 * i, col0, 0, wn, tn, l31, h);
}

// C = epi(sum_slices partial[s]) for every split-K problem of a group; one thread per output element (+ the column
// sums appended behind the M*N elements of each problem)
// out[j] = sum_b partials[b][j], j < 2n: exactly ln_colreduce_kernel (rowops.hip), addressed by a linear job block index
__device__ __forceinline__ void colreduce_job(const ColReduceJobs& cr, int b) {
    __shared__ float red[8][32];
    int pi = 0;
#pragma unroll
    for (int i = 1; i < CR_MAX; ++i)
        if (i < cr.count && b >= cr.block_start[i]) pi = i;
    const int bx = b - cr.block_start[pi];
    const int n = cr.n[pi], nb = (cr.rows[pi] + LN_BWD_ROWS_PER_BLOCK - 1) / LN_BWD_ROWS_PER_BLOCK;
    const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int j = bx * 32 + c;
    float s = 0.f;
    if (j < 2 * n)
        for (int q0 = rg; q0 < nb; q0 += 64) {             // eight partial rows in flight at once (same summation order)
            float x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = cr.partials[pi][(size_t)min(q0 + 8 * u, nb - 1) * 2 * n + j];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (q0 + 8 * u < nb) ? x[u] : 0.f;
        }
    red[rg][c] = s;
    __syncthreads();
    if (rg == 0 && j < 2 * n) {
        float t = red[0][c];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += red[k][c];
        if (j < n) cr.dgamma[pi][j] = t; else cr.dbeta[pi][j - n] = t;
    }
}

__global__ void __launch_bounds__(256) splitk_reduce_kernel(const GemmGroup g, const ColReduceJobs cr, const int red_blocks) {
    if ((int)blockIdx.x >= red_blocks) { colreduce_job(cr, (int)blockIdx.x - red_blocks); return; }
    const int e = blockIdx.x * 256 + threadIdx.x;
    int pi = -1;
#pragma unroll
    for (int i = 0; i < MAX_GROUP; ++i)
        if (i < g.n && g.p[i].kslices > 1 && e >= g.p[i].red_start) pi = i;
    if (pi < 0) return;
    const GemmProblem& p = g.p[pi];
    const int M = p.M, N = p.N, ks = p.kslices;
    int idx = e - p.red_start;
    const int mn = M * N;
    if (idx < mn) {
        const int row = idx / N, col = idx % N;
        // eight slices in flight at once (clamped index, masked add; same summation order): as a plain loop every slice costs a
        // round trip of its own
        // (the epilogue's operand is requested with the slices, not behind them)
        const int epi = p.epi;
        float eo = 0.f;
        if (epi == EPI_BIAS || epi == EPI_BIAS_RELU) eo = p.bias[col];
        else if (epi == EPI_MASK_RELU || epi == EPI_TANH_BWD) eo = p.aux[(size_t)row * p.ldaux + col];
        float v = 0.f;
        for (int s0 = 0; s0 < ks; s0 += 8) {
            float x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = p.partial[(size_t)min(s0 + u, ks - 1) * mn + idx];
#pragma unroll
            for (int u = 0; u < 8; ++u) v += (s0 + u < ks) ? x[u] : 0.f;
        }
        if (epi == EPI_BIAS) {
            v += eo;
        } else if (epi == EPI_BIAS_RELU) {
            v = fmaxf(v + eo, 0.f);
        } else if (epi == EPI_MASK_RELU) {
            v = eo > 0.f ? v : 0.f;
        } else if (epi == EPI_TANH_BWD) {
            v = v * (1.f - eo * eo);
        }
        p.C[(size_t)row * p.ldc + col] = v;
    } else if (p.colsum != nullptr && idx < mn + M) {
        const int r = idx - mn;
        float v = 0.f;
        for (int s0 = 0; s0 < ks; s0 += 8) {
            float x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = p.partial[(size_t)ks * mn + (size_t)min(s0 + u, ks - 1) * M + r];
#pragma unroll
            for (int u = 0; u < 8; ++u) v += (s0 + u < ks) ? x[u] : 0.f;
        }
        p.colsum[r] = v;
    }
}

hipError_t launch_splitk_reduce(const GemmGroup& g, int total_elems, hipStream_t stream, const ColReduceJobs* extra) {
    const int red_blocks = total_elems > 0 ? (total_elems + 255) / 256 : 0;
    ColReduceJobs cr{};
    if (extra != nullptr) cr = *extra;
    const int cr_blocks = cr.count > 0 ? cr.block_start[cr.count] : 0;
    if (red_blocks + cr_blocks <= 0) return hipSuccess;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(red_blocks + cr_blocks), dim3(256), 0, stream, g, cr, red_blocks);
    return hipGetLastError();
}

static const int kCfgWM[CFG_COUNT] = {2, 2, 1, 1, 4, 4};        // CFG_DMA128: 128 x 64 (LDS-DMA kernel)
static const int kCfgWN[CFG_COUNT] = {2, 1, 2, 1, 1, 2};

// <WM, WN, WK, BK> per configuration.  (BK = 64 for the 64x64 configuration -- half the barriers per K -- was measured at the
// end of round 1: 151 VGPRs = one workgroup per CU unless forced to 128 with amdgpu_waves_per_eu(4) (16 dwords of scratch);
// forced, 1024x2048x1024 alone gains ~2 % (42.7-43.8 vs 43.6-44.5 us) but the step loses 7 % (1015 vs 1093 updates/s): the
// thin-K launches and the split-K bookkeeping pay for the deeper chunk.)
#define FBHIP_CFGS(X) X(CFG_2x2x1, 2, 2, 1, 32) X(CFG_2x1x2, 2, 1, 2, 32) X(CFG_1x2x2, 1, 2, 2, 32) X(CFG_1x1x4, 1, 1, 4, 16) X(CFG_4x1x1, 4, 1, 1, 32)

template <int WM, int WN, int WK, int BK>
constexpr size_t gemm_lds_bytes() {
    size_t b = (size_t)2 * (BK * WK) * (32 * WM + 1 + 32 * WN + 1) * sizeof(float);
#ifdef FBHIP_TRACE
    b += 2 * TRACE_IT * TRACE_SLOTS * sizeof(unsigned long long);
#endif
    return b;
}

int gemm_cfg_bkt(int cfg) {
    switch (cfg) {
#define X(id, wm, wn, wk, bk) case id: return bk * wk;
        FBHIP_CFGS(X)
#undef X
        default: return 32;
    }
}

int gemm_cfg_bm(int cfg) { return 32 * kCfgWM[cfg]; }
int gemm_cfg_bn(int cfg) { return 32 * kCfgWN[cfg]; }

// once per DEVICE (hipFuncSetAttribute applies to the current device's copy of the function): raise the dynamic-LDS limit of
// every instantiation (must not happen inside a stream capture)
hipError_t gemm_init() {
    static std::mutex mu;
    static bool done_on[64] = {};
    int dev = 0;
    hipError_t de = hipGetDevice(&dev);
    if (de != hipSuccess) return de;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lk(mu);
    bool& done = done_on[dev];
    if (done) return hipSuccess;
#define X(id, wm, wn, wk, bk)                                                                                       \
    {                                                                                                                \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<wm, wn, wk, bk, false>),       \
                                           hipFuncAttributeMaxDynamicSharedMemorySize,                               \
                                           (int)gemm_lds_bytes<wm, wn, wk, bk>());                                   \
        if (e != hipSuccess) return e;                                                                               \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<wm, wn, wk, bk, true>),                   \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_lds_bytes<wm, wn, wk, bk>());  \
        if (e != hipSuccess) return e;                                                                               \
    }
    FBHIP_CFGS(X)
#undef X
    {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_dma_kernel<2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)DmaGeom<2>::LDS_BYTES);
        if (e != hipSuccess) return e;
    }
    done = true;
    return hipSuccess;
}

// (The 64x64 LDS-DMA variant tied the register-staged kernel on large GEMMs and lost on the step's launches -- 870 vs 882
// updates/s in round 1 -- and was removed in round 2; the 128x64 one stays for large generic GEMMs, fbhip_gemm.)
// the LDS-DMA kernels need whole 32-deep chunks and 16-byte aligned operands
bool gemm_problem_dma_ok(const GemmProblem& p) {
    const bool a_vec = (((uintptr_t)p.A & 15) == 0 && (p.lda & 3) == 0), b_vec = (((uintptr_t)p.B & 15) == 0 && (p.ldb & 3) == 0);
    return a_vec && b_vec && p.K > 0 && (p.K % 32) == 0 && (p.a_kcontig || p.M >= 4) && (p.b_kcontig || p.N >= 4);
}
static bool gemm_group_dma_ok(const GemmGroup& g) {
    for (int i = 0; i < g.n; ++i)
        if (!gemm_problem_dma_ok(g.p[i])) return false;
    return true;
}

void gemm_problem_finalize(GemmProblem& p, int cfg) {
    if (p.kslices < 1) { p.kslices = 1; }
    if (p.kslices == 1) p.kper = (p.K + gemm_cfg_bkt(cfg) - 1) / gemm_cfg_bkt(cfg);
    const int BM = 32 * kCfgWM[cfg], BN = 32 * kCfgWN[cfg];
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    p.a_vec = (((uintptr_t)p.A & 15) == 0 && (p.lda & 3) == 0) ? 1 : 0;
    p.b_vec = (((uintptr_t)p.B & 15) == 0 && (p.ldb & 3) == 0) ? 1 : 0;
    // Which tiles of a problem go through which XCD's L2 (workgroup b runs on XCD b % 8: observed, used for speed only, never for
    // correctness).  Default: bands of tile rows (each XCD a contiguous run of the row-major tile list).  FBHIP_GEMM_XCD2D=1: a grid of
    // xgm x xgn tile BLOCKS, the one that minimises what an XCD's L2 has to hold (M / xgm rows of A + N / xgn rows of B) -- measured in
    // round 5 (profiles/r05m_*, r05o_*, r05p_*): FETCH_SIZE 753 -> 545 MB raw per update (-28 %), and the step 0.6 % SLOWER (walker
    // 1210.4 -> 1203.2, quadruped 613.2 -> 608.8; kernel table: 64x32 tiles -3.6 %, 64x64 +0.7 %, 32x32 +8 %): the re-fetch
    // through eight L2s is served by the Infinity Cache and is not what bounds these kernels, so the bands stay the default.
    p.xgm = p.xgn = 0;
    const char* xe = getenv("FBHIP_GEMM_XCD2D");                  // (host side, when a launch is built: read every time so that tests can switch it)
    if (xe != nullptr && xe[0] == '1' && p.kslices == 1 && cfg != CFG_1x1x4) {
        long best = -1;
        for (int gm = 1; gm <= 8; gm *= 2) {
            const int gn = 8 / gm;
            if (p.tiles_m % gm || p.tiles_n % gn) continue;
            const long cost = (long)p.M / gm + (long)p.N / gn;
            if (best < 0 || cost < best) { best = cost; p.xgm = gm; p.xgn = gn; }
        }
    }
}

int pick_gemm_cfg(int M, int N, int K) {
    const long tiles32 = (long)((M + 31) / 32) * ((N + 31) / 32);
    if (K <= 64) return (N <= 32) ? CFG_4x1x1 : CFG_2x2x1;      // nothing to split
    if (tiles32 >= 768) return CFG_2x2x1;
    if (tiles32 >= 320) return (N >= M) ? CFG_1x2x2 : CFG_2x1x2;
    return CFG_1x1x4;
}

hipError_t launch_gemm_group(const GemmGroup& g, int cfg, hipStream_t stream) {
    if (g.total_tiles <= 0) return hipSuccess;
    dim3 grid(g.total_tiles), block(512);
    if (cfg == CFG_DMA128) {
        if (!gemm_group_dma_ok(g)) return hipErrorInvalidValue;
        hipLaunchKernelGGL(gemm_dma_kernel<2>, grid, dim3(256 + 64 * DmaGeom<2>::NPW), DmaGeom<2>::LDS_BYTES, stream, g);
        return hipGetLastError();
    }
    bool csum = false;
    for (int i = 0; i < g.n; ++i) csum = csum || g.p[i].colsum != nullptr;
    switch (cfg) {
#define X(id, wm, wn, wk, bk)                                                                                       \
    case id:                                                                                                         \
        if (csum) hipLaunchKernelGGL((gemm_kernel<wm, wn, wk, bk, true>), grid, block, (gemm_lds_bytes<wm, wn, wk, bk>()), stream, g); \
        else hipLaunchKernelGGL((gemm_kernel<wm, wn, wk, bk, false>), grid, block, (gemm_lds_bytes<wm, wn, wk, bk>()), stream, g); \
        break;
        FBHIP_CFGS(X)
#undef X
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace fbhip
