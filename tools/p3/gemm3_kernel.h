// Grouped GEMM on P3 operands (p3.h): fp32-accurate products from six v_mfma_f32_32x32x16_bf16 per 32x32x16 block.
//
//   C[M,N] = epi( sum_k A(m,k) B(n,k) )        A, B given as P3 images, C written as fp32 AND (optionally) as a P3 image
//
// so that a chain of layers never converts: every producer (this kernel's epilogue, the LayerNorm kernels, the optimiser pass)
// emits the planes the next GEMM stages.  Same three operand orientations as gemm.hip (a_kcontig / b_kcontig), same epilogues.
//
// Workgroup = 8 waves, wave-specialised like gemm_dma_kernel: waves 0-3 CONSUME (2 x 2, each TM x TN accumulators of 32x32 =
// 64 TM x 64 TN tile per workgroup, K chunks of 32), waves 4-7 PRODUCE with global_load_lds_dwordx4 (1 KiB per instruction,
// no VGPR round trip) into a ring of S chunk buffers; one raw s_barrier per chunk, counted vmcnt.  A chunk buffer holds, per
// operand, R = 128 (or 64) blocks of 192 bytes:
//   k-contiguous operand (rows of the tile = rows of the matrix, the chunk = one block per row): block of tile row r at
//     192 r; inside each 64-byte plane the four 16-byte granules are stored at position q ^ ((r >> 2) & 3) -- the DMA writes
//     lane-linear, so the permutation is applied to the SOURCE address -- which makes the 16-lane groups of ds_read_b128
//     (16 distinct rows, one granule each) hit 16 distinct bank slots.  A fragment = 8 consecutive k of one row = ONE b128.
//   k-strided operand (rows of the tile = 32-column blocks of 32 consecutive matrix rows k): block (k, nb) at slot
//     (k >> 2) 4 NB + 4 nb + (k & 3): four consecutive k of a column block sit in consecutive slots, i.e. at four distinct
//     64-byte bank windows (192 j mod 256), which is what ds_read_b64_tr_b16 wants: a 16-lane group fetches a [4 k][16 col]
//     patch (one 8-byte quarter row per lane) and hands lane c the 4 k of column c.  A fragment = two transposing reads.
// MFMA operand layout (32x32x16 bf16): lane l holds row/col (l & 31), k = 8 (l >> 5) .. + 7; accumulator as in gemm.hip.
//
// Consumer software pipeline (one workgroup per CU: 144 KiB of LDS at TM = TN = 2): the two k-steps of a chunk use two
// fragment register sets; the chunk barrier sits BETWEEN them -- reads of (chunk, step 1) retire, barrier, reads of
// (chunk + 1, step 0) are issued and the MFMAs of (chunk, step 1) run under them -- so the matrix pipe never waits for an
// LDS round trip at a chunk boundary.
//
// Epilogue through LDS (the ring is free by then): the accumulators (lane = column) are written to a wave-private fp32
// patch and read back row-major, 8 consecutive columns per lane, so that C goes out as full 128-byte lines and its P3 image
// as 16-byte stores (the bf16 split costs ~5.5 VALU per element once per produced element, not once per use).
#pragma once
#include "p3_problem.h"
#include "p3.h"

#include <type_traits>

namespace fbhip {

// -DG3_KNOCK (tools/gemm3_probe.hip only): a runtime mask that removes one pipeline stage at a time -- 1 no DMA issue, 2 no MFMA,
// 4 no fragment reads -- to see what a chunk's time is made of
#ifdef G3_KNOCK
__device__ int g3_knock_mask;
#define G3_KNOCKED(bit) (g3_knock & (bit))       // ``g3_knock``: the mask read ONCE at kernel entry and passed down (a load in the
#define G3_KARG , int g3_knock                   // loops would drain the DMAs in flight / the LDS reads: lgkmcnt counts scalar loads too)
#define G3_KPASS , g3_knock
#else
#define G3_KNOCKED(bit) false
#define G3_KARG
#define G3_KPASS
#endif

typedef __bf16 g3_bf16x8 __attribute__((ext_vector_type(8)));
typedef short g3_short4 __attribute__((ext_vector_type(4)));
typedef short g3_short8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* g3_lds_ptr;
typedef const __attribute__((address_space(1))) void* g3_glb_ptr;
typedef __attribute__((address_space(3))) g3_short4* g3_lds_s4;

__device__ __forceinline__ void g3_glds16(const char* src, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((g3_glb_ptr)src, (g3_lds_ptr)lds_dst, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void g3_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until at most ``chunks`` chunks (P DMAs each) are still in flight
template <int P, int MAXC>
__device__ __forceinline__ void g3_wait_chunks(int chunks) {
    static_assert(MAXC >= 0 && MAXC <= 3 && P * MAXC < 64, "vmcnt is a 6-bit immediate");
    if (MAXC >= 3 && chunks >= 3) g3_wait_vmcnt<P * (MAXC >= 3 ? 3 : 0)>();
    else if (MAXC >= 2 && chunks >= 2) g3_wait_vmcnt<P * (MAXC >= 2 ? 2 : 0)>();
    else if (MAXC >= 1 && chunks >= 1) g3_wait_vmcnt<P * (MAXC >= 1 ? 1 : 0)>();
    else g3_wait_vmcnt<0>();
}

template <int TM, int TN, int S_>
struct G3Geom {
    static constexpr int BM = 64 * TM, BN = 64 * TN, BK = 32, S = S_;
    static constexpr int A_BYTES = BM * P3_BLOCK_BYTES, B_BYTES = BN * P3_BLOCK_BYTES, STAGE = A_BYTES + B_BYTES;
    static constexpr int PA = A_BYTES / 1024, PB = B_BYTES / 1024, PIECES = PA + PB;
    static constexpr int NPW = 4;
    static constexpr int P = PIECES / NPW;
    static constexpr int EPI_LD = 36;                                  // floats per row of the epilogue patch (32 + pad)
    static constexpr size_t EPI_BYTES = (size_t)4 * TM * TN * 32 * EPI_LD * 4;   // TM x TN 32-row patches per consumer wave
    static constexpr size_t LDS_BYTES = (size_t)S * STAGE > EPI_BYTES ? (size_t)S * STAGE : EPI_BYTES;
    static_assert(PIECES % NPW == 0, "pieces must split evenly over the producer waves");
};

// one operand fragment (8 bf16 of one row/col) from a chunk buffer
template <bool KC>
__device__ __forceinline__ g3_bf16x8 g3_read_frag(const char* __restrict__ lds, int off /* lane part */, int imm /* compile-time part */,
                                                   int tstride /* k-strided: bytes between the two transposing reads */) {
    if constexpr (KC) {
        return *reinterpret_cast<const g3_bf16x8*>(lds + off + imm);
    } else {
        const g3_short4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((g3_lds_s4)(lds + off + imm));
        const g3_short4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((g3_lds_s4)(lds + off + imm + tstride));
        g3_short8 v;
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
        v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        return __builtin_bit_cast(g3_bf16x8, v);
    }
}

template <int TM, int TN>
struct G3Frags { g3_bf16x8 a[TM][3]; g3_bf16x8 b[TN][3]; };

// fragment reads of k-step ``s`` of one chunk buffer
template <int TM, int TN, int S, bool AKC, bool BKC>
__device__ __forceinline__ void g3_read_step(const char* __restrict__ st, int offA0, int offA1, int offB0, int offB1, int s,
                                             G3Frags<TM, TN>& f G3_KARG) {
    using G = G3Geom<TM, TN, S>;
    constexpr int NBA = G::BM / 32, NBB = G::BN / 32;
    // k-contiguous: lane offsets differ per k-step (swizzle), immediates walk (i, plane); k-strided: one lane offset,
    // immediates walk (s, i, plane) and the second transposing read sits one slot group (4 NB blocks) further
    const int oa = AKC ? (s ? offA1 : offA0) : offA0, ob = BKC ? (s ? offB1 : offB0) : offB0;
    if (G3_KNOCKED(4)) return;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int imm = AKC ? (i * 32 * P3_BLOCK_BYTES + 64 * p) : (s * 4 * 4 * NBA * P3_BLOCK_BYTES + i * 4 * P3_BLOCK_BYTES + 64 * p);
            f.a[i][p] = g3_read_frag<AKC>(st, oa, imm, 4 * NBA * P3_BLOCK_BYTES);
        }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int imm = G::A_BYTES + (BKC ? (j * 32 * P3_BLOCK_BYTES + 64 * p) : (s * 4 * 4 * NBB * P3_BLOCK_BYTES + j * 4 * P3_BLOCK_BYTES + 64 * p));
            f.b[j][p] = g3_read_frag<BKC>(st, ob, imm, 4 * NBB * P3_BLOCK_BYTES);
        }
}

// the six products of one k-step, smallest terms first; ``ones`` != 0 also accumulates the row sums of A (bias gradients)
template <int TM, int TN, bool CSUM>
__device__ __forceinline__ void g3_mfma_step(const G3Frags<TM, TN>& f, floatx16 (&acc)[TM][TN], floatx16 (&accs)[TM] G3_KARG) {
    // plane pairs (a, b): (0,2) (2,0) (1,1) (0,1) (1,0) (0,0)
    constexpr int PA_[6] = {0, 2, 1, 0, 1, 0}, PB_[6] = {2, 0, 1, 1, 0, 0};
#ifdef G3_KNOCK
    if (G3_KNOCKED(2)) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(f.a[i][p]));
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(f.b[j][p]));
        }
        return;
    }
#endif
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][PA_[t]], f.b[j][PB_[t]], acc[i][j], 0, 0, 0);
    if constexpr (CSUM) {
        g3_bf16x8 ones;
#pragma unroll
        for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
#pragma unroll
        for (int p = 2; p >= 0; --p)
#pragma unroll
            for (int i = 0; i < TM; ++i) accs[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][p], ones, accs[i], 0, 0, 0);
    }
}

template <int TM, int TN, int S, bool AKC, bool BKC, bool CSUM>
__device__ __forceinline__ void g3_consume(const char* __restrict__ smem, int nt, int lane, int wm, int wn,
                                           floatx16 (&acc)[TM][TN], floatx16 (&accs)[TM] G3_KARG) {
    using G = G3Geom<TM, TN, S>;
    constexpr int NBA = G::BM / 32, NBB = G::BN / 32;
    const int l31 = lane & 31, h = lane >> 5;
    int offA0, offA1, offB0, offB1;
    {
        const int sw = (l31 >> 2) & 3;
        const int gidx = lane >> 4, nhalf = gidx & 1, hh = gidx >> 1, j4 = (lane & 15) >> 2, ci = lane & 3;
        if constexpr (AKC) {
            const int r = wm * 32 * TM + l31;
            offA0 = P3_BLOCK_BYTES * r + 16 * ((0 + h) ^ sw);
            offA1 = P3_BLOCK_BYTES * r + 16 * ((2 + h) ^ sw);
        } else {
            offA0 = P3_BLOCK_BYTES * (2 * hh * 4 * NBA + 4 * (wm * TM) + j4) + 32 * nhalf + 8 * ci;
            offA1 = offA0;
        }
        if constexpr (BKC) {
            const int r = wn * 32 * TN + l31;
            offB0 = P3_BLOCK_BYTES * r + 16 * ((0 + h) ^ sw);
            offB1 = P3_BLOCK_BYTES * r + 16 * ((2 + h) ^ sw);
        } else {
            offB0 = P3_BLOCK_BYTES * (2 * hh * 4 * NBB + 4 * (wn * TN) + j4) + 32 * nhalf + 8 * ci;
            offB1 = offB0;
        }
    }
    G3Frags<TM, TN> f0, f1;
    __builtin_amdgcn_s_barrier();                                   // barrier "init": chunk 0 has landed
    asm volatile("" ::: "memory");
    g3_read_step<TM, TN, S, AKC, BKC>(smem, offA0, offA1, offB0, offB1, 0, f0 G3_KPASS);
    for (int it = 0; it < nt; ++it) {
        const char* st = smem + (size_t)(it % S) * G::STAGE;
        g3_read_step<TM, TN, S, AKC, BKC>(st, offA0, offA1, offB0, offB1, 1, f1 G3_KPASS);
        g3_mfma_step<TM, TN, CSUM>(f0, acc, accs G3_KPASS);
        __builtin_amdgcn_sched_barrier(0);                          // (hipcc would sink the MFMAs below the barrier: they are not memory operations)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // reads of (it, step 1) retired: the buffer may be refilled
        __builtin_amdgcn_s_barrier();                               // barrier it: chunk it + 1 has landed
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (it + 1 < nt) {
            const char* sn = smem + (size_t)((it + 1) % S) * G::STAGE;
            g3_read_step<TM, TN, S, AKC, BKC>(sn, offA0, offA1, offB0, offB1, 0, f0 G3_KPASS);
        }
        g3_mfma_step<TM, TN, CSUM>(f1, acc, accs G3_KPASS);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// C (fp32) + its P3 image for the TM x TN accumulator blocks of one wave, through wave-private LDS patches.  Three phases, in
// this order: (1) every accumulator to its patch; (2) EVERY global load of the epilogue (bias of the lane's columns, the aux
// panel of relu' / tanh'); (3) patch reads, arithmetic and stores with no load in between -- on gfx9 stores count on vmcnt
// like loads, so a load issued after a store makes its s_waitcnt wait for that store to complete: with loads interleaved the
// passes serialised on store latency (20 us per 128x128 tile in the first probe, as long as the tile's whole K loop).
template <int TM, int TN, int EPI_LD, int EPI /* compile-time epilogue; -1: raw split-K partial */>
__device__ __forceinline__ void g3_store_tile(const Gemm3Problem& p, const floatx16 (&acc)[TM][TN], float* __restrict__ patches, int slice,
                                              int row0, int col0, int lane) {
    const int l31 = lane & 31, h = lane >> 5;
    const int M = p.M, N = p.N;
    constexpr int PATCH = 32 * EPI_LD;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) patches[(i * TN + j) * PATCH + ((r & 3) + 8 * (r >> 2) + 4 * h) * EPI_LD + l31] = acc[i][j][r];
    const int g = lane & 3, rr = lane >> 2;                         // 8 columns 8 g .. 8 g + 7 of rows rr and rr + 16 of each block
    constexpr int epi = EPI;
    float bias[TN][8];
    float aux[TM][TN][2][8];
    if (epi == EPI_BIAS || epi == EPI_BIAS_RELU) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) bias[j][e] = p.bias[min(col0 + 32 * j + 8 * g + e, N - 1)];
    } else if (epi == EPI_MASK_RELU || epi == EPI_TANH_BWD) {
        const bool vec = (p.ldaux & 3) == 0 && ((uintptr_t)p.aux & 15) == 0;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    const int row = min(row0 + 32 * i + rr + 16 * pass, M - 1), col = col0 + 32 * j + 8 * g;
                    const float* ax = p.aux + (size_t)row * p.ldaux;
                    if (vec && col + 7 < N) {
                        const float4 a0 = *reinterpret_cast<const float4*>(ax + col), a1 = *reinterpret_cast<const float4*>(ax + col + 4);
                        aux[i][j][pass][0] = a0.x; aux[i][j][pass][1] = a0.y; aux[i][j][pass][2] = a0.z; aux[i][j][pass][3] = a0.w;
                        aux[i][j][pass][4] = a1.x; aux[i][j][pass][5] = a1.y; aux[i][j][pass][6] = a1.z; aux[i][j][pass][7] = a1.w;
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) aux[i][j][pass][e] = ax[min(col + e, N - 1)];
                    }
                }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // loads landed, patches written: nothing but stores from here on
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int lr = rr + 16 * pass, row = row0 + 32 * i + lr, col = col0 + 32 * j + 8 * g;
                const float* pt = patches + (i * TN + j) * PATCH + lr * EPI_LD + 8 * g;
                const float4 v0 = *reinterpret_cast<const float4*>(pt), v1 = *reinterpret_cast<const float4*>(pt + 4);
                float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                if (row >= M || col >= N) continue;
                if constexpr (EPI < 0) {                            // raw partial tile; splitk_reduce applies the epilogue
                    float* part = p.partial + (size_t)slice * M * N + (size_t)row * N + col;
                    if (col + 7 < N && (N & 3) == 0) {
                        *reinterpret_cast<float4*>(part) = make_float4(v[0], v[1], v[2], v[3]);
                        *reinterpret_cast<float4*>(part + 4) = make_float4(v[4], v[5], v[6], v[7]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) if (col + e < N) part[e] = v[e];
                    }
                    continue;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (epi == EPI_BIAS) v[e] += bias[j][e];
                    else if (epi == EPI_BIAS_RELU) v[e] = fmaxf(v[e] + bias[j][e], 0.f);
                    else if (epi == EPI_MASK_RELU) v[e] = aux[i][j][pass][e] > 0.f ? v[e] : 0.f;
                    else if (epi == EPI_TANH_BWD) v[e] = v[e] * (1.f - aux[i][j][pass][e] * aux[i][j][pass][e]);
                }
                float* c = p.C + (size_t)row * p.ldc + col;
                if (col + 7 < N && (p.ldc & 3) == 0) {
                    *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<float4*>(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (col + e < N) c[e] = v[e];
                }
                if (p.C3 != nullptr) {                              // whole 8-column groups inside the padded row: ldc % 32 == 0
                    unsigned hw[4], mw[4], lw[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const P3Triple t0 = p3_split(col + 2 * e < N ? v[2 * e] : 0.f), t1 = p3_split(col + 2 * e + 1 < N ? v[2 * e + 1] : 0.f);
                        hw[e] = (unsigned)t0.h | ((unsigned)t1.h << 16);
                        mw[e] = (unsigned)t0.m | ((unsigned)t1.m << 16);
                        lw[e] = (unsigned)t0.l | ((unsigned)t1.l << 16);
                    }
                    char* c3 = p.C3 + p3_offset((size_t)row, (size_t)col, (size_t)p.ldc);
                    *reinterpret_cast<uint4*>(c3) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
                    *reinterpret_cast<uint4*>(c3 + 64) = make_uint4(mw[0], mw[1], mw[2], mw[3]);
                    *reinterpret_cast<uint4*>(c3 + 128) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
                }
            }
}

// ---- producers ------------------------------------------------------------------------------------------------------------
// One operand tile of R rows per chunk = R blocks of 192 bytes in the chunk buffer at ``lds_op`` (layouts: file header).  Two
// producer waves share an operand, half each (pw = 0 / 1).  Both variants keep the same cadence: one "init" barrier once chunk 0
// is in LDS, then at barrier ``it`` chunk it + 1 is in LDS.

// P3 image -> LDS by DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, ring of S chunks, counted vmcnt)
template <int R, int S, int STAGE>
__device__ __forceinline__ void g3_produce_dma(const char* __restrict__ base, int ld, bool kc, int r0, int nr, int kb, int nt, int pw,
                                               int lane, char* __restrict__ lds_op G3_KARG) {
    constexpr int P = R * P3_BLOCK_BYTES / 1024 / 2;           // pieces per wave and chunk
    constexpr int NB = R / 32;
    const char* src[P];
    size_t adv[P];
    const size_t rowblocks = (size_t)(ld / P3_BLOCK);
#pragma unroll
    for (int q = 0; q < P; ++q) {
        const int slot = (pw * P + q) * 64 + lane, bs = slot / 12, gq = slot % 12;
        if (kc) {
            const int r = bs, pl = gq >> 2, q4 = (gq & 3) ^ ((r >> 2) & 3);
            const int grow = min(r0 + r, nr - 1);
            src[q] = base + ((size_t)grow * rowblocks + kb / P3_BLOCK) * P3_BLOCK_BYTES + 64 * pl + 16 * q4;
            adv[q] = (size_t)P3_BLOCK_BYTES;
        } else {
            const int kk = 4 * (bs / (4 * NB)) + (bs & 3), nb = (bs >> 2) % NB;
            const int cb = min(r0 / P3_BLOCK + nb, (nr - 1) / P3_BLOCK);
            src[q] = base + ((size_t)(kb + kk) * rowblocks + cb) * P3_BLOCK_BYTES + 16 * gq;
            adv[q] = (size_t)32 * rowblocks * P3_BLOCK_BYTES;
        }
    }
    char* const dst = lds_op + (size_t)(pw * P) * 1024;
    auto issue = [&](int c) __attribute__((always_inline)) {
        char* d = dst + (size_t)(c % S) * STAGE;
#pragma unroll
        for (int q = 0; q < P; ++q)
            if (!G3_KNOCKED(1)) g3_glds16(src[q] + (size_t)c * adv[q], d + q * 1024);
    };
    for (int c = 0; c < S && c < nt; ++c) issue(c);
    g3_wait_chunks<P, S - 1>(nt - 1);                          // barrier "init": chunk 0 landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    for (int it = 0; it < nt; ++it) {
        g3_wait_chunks<P, S - 2>(nt - 2 - it);                 // chunk it + 1 landed; later ones stay in flight
        __builtin_amdgcn_s_barrier();                           // raw: a fence would drain the DMAs
        asm volatile("" ::: "memory");
        if (it + S < nt) issue(it + S);                         // the consumers retired their reads of chunk ``it`` before this barrier
    }
}

// fp32 -> registers -> three bf16 planes -> LDS: the operand is an activation / gradient panel that exists in fp32 only.  Per chunk
// and wave Q float4 loads of whole 128-byte lines, ~22 VALU per float4 for the split, 3 ds_write_b64 per float4; the loads of
// chunk it + 2 are in flight while chunk it + 1 is split and written (straight-line code: hipcc emits counted vmcnt).
template <int R, int S, int STAGE>
__device__ __forceinline__ void g3_produce_f32(const float* __restrict__ base, int ld, bool kc, int r0, int nr, int kb, int nt, int pw,
                                               int lane, char* __restrict__ lds_op) {
    constexpr int NB = R / 32;
    constexpr int Q = R / 16;                                   // float4 per lane and chunk (half a tile per wave)
    constexpr int LPR = R / 4, RPI = 64 / LPR;                  // k-strided: lanes per k-row, k-rows per instruction
    const float* src[Q];
    int dsto[Q];
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        if (kc) {
            const int r = pw * (R / 2) + 8 * i + (lane >> 3), kq = lane & 7;
            src[i] = base + (size_t)min(r0 + r, nr - 1) * ld + kb + 4 * kq;
            dsto[i] = P3_BLOCK_BYTES * r + 16 * ((kq >> 1) ^ ((r >> 2) & 3)) + 8 * (kq & 1);
        } else {
            const int k = pw * 16 + i * RPI + lane / LPR, n = 4 * (lane % LPR);
            src[i] = base + (size_t)(kb + k) * ld + min(r0 + n, (nr - 1) & ~3);
            dsto[i] = P3_BLOCK_BYTES * ((k >> 2) * 4 * NB + 4 * (n >> 5) + (k & 3)) + 2 * (n & 31);
        }
    }
    const size_t adv = kc ? (size_t)32 : (size_t)32 * ld;
    auto load = [&](int c, float4 (&x)[Q]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < Q; ++i) x[i] = *reinterpret_cast<const float4*>(src[i] + (size_t)c * adv);
    };
    auto store = [&](int c, const float4 (&x)[Q]) __attribute__((always_inline)) {
        char* d = lds_op + (size_t)(c % S) * STAGE;
#pragma unroll
        for (int i = 0; i < Q; ++i) {
            const P3Triple a = p3_split(x[i].x), b = p3_split(x[i].y), cc = p3_split(x[i].z), e = p3_split(x[i].w);
            *reinterpret_cast<uint2*>(d + dsto[i]) = make_uint2((unsigned)a.h | ((unsigned)b.h << 16), (unsigned)cc.h | ((unsigned)e.h << 16));
            *reinterpret_cast<uint2*>(d + dsto[i] + 64) = make_uint2((unsigned)a.m | ((unsigned)b.m << 16), (unsigned)cc.m | ((unsigned)e.m << 16));
            *reinterpret_cast<uint2*>(d + dsto[i] + 128) = make_uint2((unsigned)a.l | ((unsigned)b.l << 16), (unsigned)cc.l | ((unsigned)e.l << 16));
        }
    };
    // step ``it``: request chunk it + 2 into ``la``, land chunk it + 1 (in flight in ``sa``) in its buffer, meet at barrier ``it``
    auto p_step = [&](int it, auto do_load, auto do_store, float4 (&la)[Q], const float4 (&sa)[Q]) __attribute__((always_inline)) {
        if constexpr (decltype(do_load)::value) load(it + 2, la);
        if constexpr (decltype(do_store)::value) store(it + 1, sa);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    constexpr std::true_type T{};
    constexpr std::false_type F{};
    float4 x0[Q], x1[Q];
    load(0, x0);
    if (nt > 1) load(1, x1);
    else {
#pragma unroll
        for (int i = 0; i < Q; ++i) x1[i] = x0[i];
    }
    store(0, x0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                               // barrier "init": chunk 0 visible
    asm volatile("" ::: "memory");
    int it = 0;
    for (; it + 3 < nt; it += 2) {
        p_step(it, T, T, x0, x1);
        p_step(it + 1, T, T, x1, x0);
    }
    const int rem = nt - it;                                    // 1, 2 or 3 steps left
    if (rem == 3) {
        p_step(it, T, T, x0, x1);
        p_step(it + 1, F, T, x1, x0);
        p_step(it + 2, F, F, x0, x1);
    } else if (rem == 2) {
        p_step(it, F, T, x0, x1);
        p_step(it + 1, F, F, x1, x0);
    } else {
        p_step(it, F, F, x0, x1);
    }
}

template <int TM, int TN, int S>
__global__ void __launch_bounds__(512) gemm3_kernel(const Gemm3Group g) {
    using G = G3Geom<TM, TN, S>;
    constexpr int BM = G::BM, BN = G::BN, BK = G::BK;
    extern __shared__ __attribute__((aligned(16))) char smem3[];
#ifdef G3_KNOCK
    const int g3_knock = g3_knock_mask;
#endif

    const int orig = blockIdx.x;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < MAX_GROUP; ++i)
        if (i < g.n && orig >= g.p[i].tile_start) pi = i;
    const Gemm3Problem& p = g.p[pi];
    const int nwg = p.tiles_m * p.tiles_n * p.kslices;
    const int jb = orig - p.tile_start;
    const int xcd = jb & 7, qd = nwg >> 3, rm = nwg & 7;
    const int t = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (jb >> 3);
    const int tiles_mn = p.tiles_m * p.tiles_n;
    const int slice = t / tiles_mn, tt = t % tiles_mn;
    const int tn = tt % p.tiles_n, tm = tt / p.tiles_n;
    const int M = p.M, N = p.N;
    const int kb = slice * p.kper * BK;
    const int K = min(p.K, kb + p.kper * BK);
    const int row0 = tm * BM, col0 = tn * BN;
    const int nt = (K - kb) / BK;

    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;

    if (wid >= 4) {
        // =========================================================================================== PRODUCERS
        // waves 4, 5 stage operand A, waves 6, 7 operand B (half a tile each); per operand either the LDS-DMA path (a P3 image
        // exists: parameters) or the register path (fp32 activations / gradient panels: load, split, ds_write)
        const int w = wid - 4, pw = w & 1;
        if (w < 2) {
            if (p.A3 != nullptr) g3_produce_dma<BM, S, G::STAGE>(p.A3, p.lda, p.a_kcontig != 0, row0, M, kb, nt, pw, lane, smem3 G3_KPASS);
            else g3_produce_f32<BM, S, G::STAGE>(p.A, p.lda, p.a_kcontig != 0, row0, M, kb, nt, pw, lane, smem3);
        } else {
            if (p.B3 != nullptr) g3_produce_dma<BN, S, G::STAGE>(p.B3, p.ldb, p.b_kcontig != 0, col0, N, kb, nt, pw, lane, smem3 + G::A_BYTES G3_KPASS);
            else g3_produce_f32<BN, S, G::STAGE>(p.B, p.ldb, p.b_kcontig != 0, col0, N, kb, nt, pw, lane, smem3 + G::A_BYTES);
        }
        return;
    }

    // ============================================================================================== CONSUMERS
    const int wm = wid >> 1, wn = wid & 1;
    floatx16 acc[TM][TN];
    floatx16 accs[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) accs[i][e] = 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    }
    const bool csum = p.colsum != nullptr && tn == 0 && wn == 0;
    const int mode = p.a_kcontig * 2 + p.b_kcontig;
    if (mode == 3) g3_consume<TM, TN, S, true, true, false>(smem3, nt, lane, wm, wn, acc, accs G3_KPASS);
    else if (mode == 2) g3_consume<TM, TN, S, true, false, false>(smem3, nt, lane, wm, wn, acc, accs G3_KPASS);
    else if (mode == 0) {
        if (csum) g3_consume<TM, TN, S, false, false, true>(smem3, nt, lane, wm, wn, acc, accs G3_KPASS);
        else g3_consume<TM, TN, S, false, false, false>(smem3, nt, lane, wm, wn, acc, accs G3_KPASS);
    } else g3_consume<TM, TN, S, false, true, false>(smem3, nt, lane, wm, wn, acc, accs G3_KPASS);

    // every wave's last fragment reads retired before its last barrier; the producers are done: the ring is free
    float* patches = reinterpret_cast<float*>(smem3) + wid * TM * TN * 32 * G::EPI_LD;
    if (csum) {                                                      // column sums of the A operand (bias gradient of a wgrad)
        const int l31 = lane & 31, h = lane >> 5;
        if (l31 == 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + wm * 32 * TM + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (row < M) {
                        if (p.kslices > 1) p.partial[(size_t)p.kslices * M * N + (size_t)slice * M + row] = accs[i][r];
                        else p.colsum[row] = accs[i][r];
                    }
                }
        }
    }
    const int r0w = row0 + wm * 32 * TM, c0w = col0 + wn * 32 * TN;
    if (p.kslices > 1) g3_store_tile<TM, TN, G::EPI_LD, -1>(p, acc, patches, slice, r0w, c0w, lane);
    else switch (p.epi) {
        case EPI_BIAS: g3_store_tile<TM, TN, G::EPI_LD, EPI_BIAS>(p, acc, patches, slice, r0w, c0w, lane); break;
        case EPI_BIAS_RELU: g3_store_tile<TM, TN, G::EPI_LD, EPI_BIAS_RELU>(p, acc, patches, slice, r0w, c0w, lane); break;
        case EPI_MASK_RELU: g3_store_tile<TM, TN, G::EPI_LD, EPI_MASK_RELU>(p, acc, patches, slice, r0w, c0w, lane); break;
        case EPI_TANH_BWD: g3_store_tile<TM, TN, G::EPI_LD, EPI_TANH_BWD>(p, acc, patches, slice, r0w, c0w, lane); break;
        default: g3_store_tile<TM, TN, G::EPI_LD, EPI_NONE>(p, acc, patches, slice, r0w, c0w, lane); break;
    }
}

}  // namespace fbhip
