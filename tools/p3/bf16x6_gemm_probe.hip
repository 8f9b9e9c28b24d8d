// Feasibility probe (not part of the product; DESIGN.md section 9, lever 4): an fp32-accurate GEMM on the bf16 matrix pipe.
//   x = x_hi + x_mid + x_lo (three bf16 pieces, 24 mantissa bits);  a.b ~ hi.hi + hi.mid + mid.hi + hi.lo + lo.hi + mid.mid
// Six v_mfma_f32_32x32x16_bf16 (8 passes each, k = 16) replace eight v_mfma_f32_32x32x2_f32 (16 passes each): 192 matrix-pipe
// cycles per 32x32x16 block instead of 512.  The price is bytes: 6 B per operand element instead of 4 in HBM / L2 / LDS.
// This program measures what a plain kernel gets out of that trade on the update step's shapes, with the operands ALREADY
// split into planes (in a product the pieces would be written by the producing epilogues and by the Adam pass), plus the cost
// of a standalone split pass, and checks the result against fp64.
//   C[M,N] = A[M,K] . B[N,K]^T   (both k-contiguous: the NT layout of a Linear forward)
// Kernel: 4 waves (2 x 2), wave tile 32 WT x 32 WT, workgroup tile 64 WT square, K chunks of 16, two LDS stages,
// planes staged as [row][16 k] bf16 with a 48-byte row stride (ds_read_b128 conflict-free), one barrier per chunk.
//   hipcc --offload-arch=gfx950 -O3 tools/bf16x6_gemm_probe.hip -o tools/scratch/bf16x6_gemm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void split3(const float* __restrict__ x, __bf16* __restrict__ p, size_t n) {       // p: [3][n]
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = x[i];
        const __bf16 h = (__bf16)v;
        const float r1 = v - (float)h;
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;
        p[i] = h; p[n + i] = m; p[2 * n + i] = (__bf16)r2;
    }
}

// BK k per chunk (16 or 32), LDS row stride BK + 8 bf16 (48 / 80 bytes: ds_read_b128 conflict-free); DB: two LDS stages and one
// barrier per chunk, else one stage, the next chunk waits in registers, two barriers per chunk
template <int WT, int BK, bool DB>
__global__ void __launch_bounds__(256) gemm_bf16x6(const __bf16* __restrict__ Ap, const __bf16* __restrict__ Bp,
                                                    float* __restrict__ C, int M, int N, int K, int kper) {
    constexpr int ROWS = 64 * WT;                         // tile rows of either operand
    constexpr int RS = BK + 8, UPR = BK / 8;              // row stride (bf16), 16-byte units per row
    constexpr int PLANE = ROWS * RS;                      // bf16 per staged plane
    constexpr int STAGE = 6 * PLANE;                      // A hi/mid/lo, B hi/mid/lo
    constexpr int U = 6 * ROWS * UPR / 256;               // 16-byte units per thread per chunk
    extern __shared__ __attribute__((aligned(16))) __bf16 smem[];      // 2 * STAGE
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, kh = lane >> 5;
    const int tiles_n = N / ROWS;
    const int row0 = (blockIdx.x / tiles_n) * ROWS, col0 = (blockIdx.x % tiles_n) * ROWS;
    const size_t planeA = (size_t)M * K, planeB = (size_t)N * K;

    // staging geometry of this thread's U units
    const __bf16* src[U];
    int dst[U];
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int q = tid + 256 * i;
        const int po = q / (ROWS * UPR), r = (q % (ROWS * UPR)) / UPR, half = q % UPR;
        const int pl = po % 3;
        src[i] = (po < 3 ? Ap + pl * planeA + (size_t)(row0 + r) * K + half * 8 : Bp + pl * planeB + (size_t)(col0 + r) * K + half * 8) +
                 (size_t)blockIdx.y * kper;              // grid-level K slices: slice y contracts k in [y kper, (y + 1) kper)
        dst[i] = po * PLANE + r * RS + half * 8;
    }
    floatx16 acc[WT][WT];
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    uint4 regs[U];
    const int nchunks = kper / BK;
    C += (size_t)blockIdx.y * M * N;                      // raw partial tiles, folded by reduce_slices
#pragma unroll
    for (int i = 0; i < U; ++i) regs[i] = *reinterpret_cast<const uint4*>(src[i]);
#pragma unroll
    for (int i = 0; i < U; ++i) *reinterpret_cast<uint4*>(smem + dst[i]) = regs[i];
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const __bf16* st = smem + (DB ? (c & 1) * STAGE : 0);
        const int cn = c + 1 < nchunks ? c + 1 : c;              // (branch-free: the last iteration re-loads its own chunk)
#pragma unroll
        for (int i = 0; i < U; ++i) regs[i] = *reinterpret_cast<const uint4*>(src[i] + (size_t)cn * BK);
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
        bf16x8 a[3][WT], b[3][WT];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < WT; ++i) {
                a[p][i] = *reinterpret_cast<const bf16x8*>(st + p * PLANE + (wm * 32 * WT + i * 32 + l31) * RS + ks * 16 + kh * 8);
                b[p][i] = *reinterpret_cast<const bf16x8*>(st + (3 + p) * PLANE + (wn * 32 * WT + i * 32 + l31) * RS + ks * 16 + kh * 8);
            }
        // small terms first: mid.mid, hi.lo, lo.hi, hi.mid, mid.hi, hi.hi
#define TERM(pa, pb)                                                                                              \
        _Pragma("unroll") for (int i = 0; i < WT; ++i)                                                            \
            _Pragma("unroll") for (int j = 0; j < WT; ++j)                                                        \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa][i], b[pb][j], acc[i][j], 0, 0, 0);
        TERM(1, 1) TERM(0, 2) TERM(2, 0) TERM(0, 1) TERM(1, 0) TERM(0, 0)
#undef TERM
        }
        __bf16* nx = smem + (DB ? ((c + 1) & 1) * STAGE : 0);
        if (!DB) __syncthreads();                                // every wave is done reading the only stage
#pragma unroll
        for (int i = 0; i < U; ++i) *reinterpret_cast<uint4*>(nx + dst[i]) = regs[i];
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = row0 + wm * 32 * WT + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
                const int cc = col0 + wn * 32 * WT + j * 32 + l31;
                C[(size_t)r * N + cc] = acc[i][j][e];
            }
}

// The same product from fp32 operands, split INSIDE the kernel while staging (64x64 tiles, K chunks of 32, two LDS stages):
// x_hi = x & 0xffff0000, r = x - x_hi, x_mid = r & 0xffff0000, x_lo = r - x_mid (exact: 8 + 8 + 8 mantissa bits), packed in pairs.
__device__ __forceinline__ void split_store(const float4 v, __bf16* dst_hi, int plane_stride) {
    unsigned hi[4], mid[4], lo[4];
    const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned xb = __float_as_uint(x[i]);
        hi[i] = xb & 0xffff0000u;
        const float r1 = x[i] - __uint_as_float(hi[i]);
        mid[i] = __float_as_uint(r1) & 0xffff0000u;
        lo[i] = __float_as_uint(r1 - __uint_as_float(mid[i]));
    }
    uint2 ph, pm, pl;
    ph.x = (hi[0] >> 16) | hi[1]; ph.y = (hi[2] >> 16) | hi[3];
    pm.x = (mid[0] >> 16) | mid[1]; pm.y = (mid[2] >> 16) | mid[3];
    pl.x = (lo[0] >> 16) | (lo[1] & 0xffff0000u); pl.y = (lo[2] >> 16) | (lo[3] & 0xffff0000u);
    *reinterpret_cast<uint2*>(dst_hi) = ph;
    *reinterpret_cast<uint2*>(dst_hi + plane_stride) = pm;
    *reinterpret_cast<uint2*>(dst_hi + 2 * plane_stride) = pl;
}

__global__ void __launch_bounds__(256) gemm_bf16x6_insplit(const float* __restrict__ A, const float* __restrict__ B,
                                                           float* __restrict__ C, int M, int N, int K) {
    constexpr int ROWS = 64, BK = 32, RS = BK + 8, PLANE = ROWS * RS, STAGE = 6 * PLANE;
    extern __shared__ __attribute__((aligned(16))) __bf16 smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, kh = lane >> 5;
    const int tiles_n = N / ROWS;
    const int row0 = (blockIdx.x / tiles_n) * ROWS, col0 = (blockIdx.x % tiles_n) * ROWS;
    // 4 float4 per thread per chunk: A rows tid/8 and 32 + tid/8, B the same; quad tid % 8
    const int r = tid >> 3, q = tid & 7;
    const float* srcA0 = A + (size_t)(row0 + r) * K + 4 * q;
    const float* srcA1 = A + (size_t)(row0 + 32 + r) * K + 4 * q;
    const float* srcB0 = B + (size_t)(col0 + r) * K + 4 * q;
    const float* srcB1 = B + (size_t)(col0 + 32 + r) * K + 4 * q;
    const int dA0 = r * RS + 4 * q, dA1 = (32 + r) * RS + 4 * q, dB0 = 3 * PLANE + dA0, dB1 = 3 * PLANE + dA1;
    floatx16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    float4 ra0 = *reinterpret_cast<const float4*>(srcA0), ra1 = *reinterpret_cast<const float4*>(srcA1);
    float4 rb0 = *reinterpret_cast<const float4*>(srcB0), rb1 = *reinterpret_cast<const float4*>(srcB1);
    split_store(ra0, smem + dA0, PLANE); split_store(ra1, smem + dA1, PLANE);
    split_store(rb0, smem + dB0, PLANE); split_store(rb1, smem + dB1, PLANE);
    __syncthreads();
    const int nchunks = K / BK;
    for (int c = 0; c < nchunks; ++c) {
        const __bf16* st = smem + (c & 1) * STAGE;
        const int cn = c + 1 < nchunks ? c + 1 : c;
        ra0 = *reinterpret_cast<const float4*>(srcA0 + (size_t)cn * BK); ra1 = *reinterpret_cast<const float4*>(srcA1 + (size_t)cn * BK);
        rb0 = *reinterpret_cast<const float4*>(srcB0 + (size_t)cn * BK); rb1 = *reinterpret_cast<const float4*>(srcB1 + (size_t)cn * BK);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[3], b[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                a[p] = *reinterpret_cast<const bf16x8*>(st + p * PLANE + (wm * 32 + l31) * RS + ks * 16 + kh * 8);
                b[p] = *reinterpret_cast<const bf16x8*>(st + (3 + p) * PLANE + (wn * 32 + l31) * RS + ks * 16 + kh * 8);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
        }
        __bf16* nx = smem + ((c + 1) & 1) * STAGE;
        split_store(ra0, nx + dA0, PLANE); split_store(ra1, nx + dA1, PLANE);
        split_store(rb0, nx + dB0, PLANE); split_store(rb1, nx + dB1, PLANE);
        __syncthreads();
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int rr = row0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        C[(size_t)rr * N + col0 + wn * 32 + l31] = acc[e];
    }
}

__global__ void reduce_slices(const float4* __restrict__ part, float4* __restrict__ out, size_t n4, int S) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 a = part[i];
        for (int s = 1; s < S; ++s) { const float4 b = part[(size_t)s * n4 + i]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
        out[i] = a;
    }
}

template <int WT, int BK, bool DB>
static double run(const __bf16* Ap, const __bf16* Bp, float* C, int M, int N, int K, int iters, int S = 1, float* Cpart = nullptr) {
    constexpr int ROWS = 64 * WT;
    const size_t lds = (size_t)(DB ? 2 : 1) * 6 * ROWS * (BK + 8) * sizeof(__bf16);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16x6<WT, BK, DB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = (M / ROWS) * (N / ROWS);
    auto go = [&]() {
        hipLaunchKernelGGL((gemm_bf16x6<WT, BK, DB>), dim3(grid, S), dim3(256), lds, 0, Ap, Bp, S > 1 ? Cpart : C, M, N, K, K / S);
        if (S > 1) hipLaunchKernelGGL(reduce_slices, dim3(512), dim3(256), 0, 0, (const float4*)Cpart, (float4*)C, (size_t)M * N / 4, S);
    };
    for (int i = 0; i < 3; ++i) go();
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) go();
    CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / iters;
}

int main() {
    struct Shape { int M, N, K; } shapes[] = {{1024, 2048, 1024}, {1024, 1024, 1024}, {2048, 1024, 1024}, {1024, 1024, 2048}, {4096, 4096, 1024}, {4096, 4096, 4096}};
    for (auto& s : shapes) {
        const int M = s.M, N = s.N, K = s.K;
        std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
        srand(1);
        for (auto& v : hA) v = (float)rand() / RAND_MAX * 2.f - 1.f;
        for (auto& v : hB) v = ((float)rand() / RAND_MAX * 2.f - 1.f) / sqrtf((float)K);
        float *dA, *dB, *dC; __bf16 *pA, *pB;
        CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
        CK(hipMalloc(&pA, hA.size() * 6)); CK(hipMalloc(&pB, hB.size() * 6));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(split3, dim3(1024), dim3(256), 0, 0, dA, pA, hA.size());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(split3, dim3(1024), dim3(256), 0, 0, dA, pA, hA.size());
        CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
        float ms_split; CK(hipEventElapsedTime(&ms_split, e0, e1));
        hipLaunchKernelGGL(split3, dim3(1024), dim3(256), 0, 0, dB, pB, hB.size());
        CK(hipDeviceSynchronize());
        const double flop = 2.0 * M * N * K;
        float* dPart; CK(hipMalloc(&dPart, (size_t)8 * M * N * 4));
        struct V { int wt, bk, db, S; } vs[] = {{1, 16, 1, 1}, {2, 16, 1, 1}, {1, 32, 1, 1}, {1, 32, 0, 1}, {1, 64, 1, 1}, {1, 64, 0, 1}, {2, 32, 1, 1}, {2, 64, 0, 1},
                                              {2, 32, 1, 2}, {2, 32, 1, 4}, {2, 32, 1, 8}, {1, 32, 1, 2}};
        for (auto& v : vs) {
            CK(hipMemset(dC, 0, (size_t)M * N * 4));
            const int iters = K >= 4096 ? 5 : 20;
            double us = 0;
            if (v.S > 1 && (M * (size_t)N > 2048u * 2048u)) continue;       // K slices: the step's sizes only
#define V_(W, Bk, D) if (v.wt == W && v.bk == Bk && v.db == D) us = run<W, Bk, (D != 0)>(pA, pB, dC, M, N, K, iters, v.S, dPart);
            V_(1, 16, 1) V_(2, 16, 1) V_(1, 32, 1) V_(1, 32, 0) V_(1, 64, 1) V_(1, 64, 0) V_(2, 32, 1) V_(2, 64, 0)
#undef V_
            std::vector<float> hC((size_t)M * N);
            CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
            double num = 0, den = 0, num32 = 0;
            for (int t = 0; t < 200; ++t) {                       // sampled entries against fp64 (and a plain fp32 dot product)
                const int r = rand() % M, c = rand() % N;
                double ref = 0; float f32 = 0.f;
                for (int k = 0; k < K; ++k) { ref += (double)hA[(size_t)r * K + k] * hB[(size_t)c * K + k]; f32 += hA[(size_t)r * K + k] * hB[(size_t)c * K + k]; }
                num += (hC[(size_t)r * N + c] - ref) * (hC[(size_t)r * N + c] - ref); den += ref * ref; num32 += (f32 - ref) * (f32 - ref);
            }
            printf("%5d x %5d x %5d  wave tile %3d^2, K chunk %2d, %d LDS stage(s), %d K slice(s) (%4d workgroups): %8.1f us  %6.1f TFLOP/s fp32-equivalent   rel L2 err %.2e (serial fp32 dot: %.2e)   [split pass of A: %.1f us]\n",
                   M, N, K, 32 * v.wt, v.bk, v.db ? 2 : 1, v.S, v.S * (M / (64 * v.wt)) * (N / (64 * v.wt)), us, flop / us / 1e6, sqrt(num / den), sqrt(num32 / den), ms_split * 1e3 / 10);
        }
        {
            constexpr size_t lds = (size_t)2 * 6 * 64 * 40 * sizeof(__bf16);
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16x6_insplit), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            CK(hipMemset(dC, 0, (size_t)M * N * 4));
            const int grid = (M / 64) * (N / 64), iters = K >= 4096 ? 5 : 20;
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm_bf16x6_insplit, dim3(grid), dim3(256), lds, 0, dA, dB, dC, M, N, K);
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(gemm_bf16x6_insplit, dim3(grid), dim3(256), lds, 0, dA, dB, dC, M, N, K);
            CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / iters;
            std::vector<float> hC((size_t)M * N);
            CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
            double num = 0, den = 0;
            for (int t = 0; t < 200; ++t) {
                const int r = rand() % M, c = rand() % N;
                double ref = 0;
                for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)r * K + k] * hB[(size_t)c * K + k];
                num += (hC[(size_t)r * N + c] - ref) * (hC[(size_t)r * N + c] - ref); den += ref * ref;
            }
            printf("%5d x %5d x %5d  SPLIT IN THE KERNEL from fp32 operands, wave tile 32^2, K chunk 32, 2 LDS stages (%4d workgroups): %8.1f us  %6.1f TFLOP/s fp32-equivalent   rel L2 err %.2e\n",
                   M, N, K, grid, us, flop / us / 1e6, sqrt(num / den));
        }
        hipFree(dPart); hipFree(dA); hipFree(dB); hipFree(dC); hipFree(pA); hipFree(pB);
    }
    return 0;
}
