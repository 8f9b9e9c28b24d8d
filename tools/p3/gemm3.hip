// Host side of the P3 GEMM (gemm3_kernel.h): tile configurations, launch, and the fp32 -> P3 split kernels for operands whose
// producer does not emit planes (input panels assembled by several row kernels, parameters after the host wrote them).
#include "gemm3_kernel.h"

namespace fbhip {

// ---- fp32 -> P3 ----------------------------------------------------------------------------------------------------
// grouped: up to P3_SPLIT_MAX views per launch, one thread per 8 consecutive columns (two float4 in, three 16-byte stores out)
__global__ void __launch_bounds__(256) p3_split_kernel(const P3SplitJobs jobs) {
    int ji = 0;
#pragma unroll
    for (int i = 1; i < P3_SPLIT_MAX; ++i)
        if (i < jobs.n && (int)blockIdx.x >= jobs.j[i].block_start) ji = i;
    const P3SplitJob& j = jobs.j[ji];
    const int per_row = j.cols >> 3;
    const long idx = (long)((int)blockIdx.x - j.block_start) * 256 + threadIdx.x;
    if (idx >= (long)j.rows * per_row) return;
    const int r = (int)(idx / per_row), c = 8 * (int)(idx % per_row);
    const float* src = j.x + (size_t)r * j.ld + c;
    const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    unsigned hw[4], mw[4], lw[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const P3Triple t0 = p3_split(v[2 * e]), t1 = p3_split(v[2 * e + 1]);
        hw[e] = (unsigned)t0.h | ((unsigned)t1.h << 16);
        mw[e] = (unsigned)t0.m | ((unsigned)t1.m << 16);
        lw[e] = (unsigned)t0.l | ((unsigned)t1.l << 16);
    }
    char* d = j.x3 + p3_offset((size_t)r, (size_t)c, (size_t)j.ld);
    *reinterpret_cast<uint4*>(d) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
    *reinterpret_cast<uint4*>(d + 64) = make_uint4(mw[0], mw[1], mw[2], mw[3]);
    *reinterpret_cast<uint4*>(d + 128) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
}

hipError_t launch_p3_split_group(P3SplitJobs jobs, hipStream_t s) {
    int start = 0;
    for (int i = 0; i < jobs.n; ++i) {
        P3SplitJob& j = jobs.j[i];
        if ((j.ld & 31) || (j.cols & 7) || ((uintptr_t)j.x & 15) || ((uintptr_t)j.x3 & 15)) return hipErrorInvalidValue;
        j.block_start = start;
        start += (int)(((long)j.rows * (j.cols >> 3) + 255) / 256);
    }
    if (start <= 0) return hipSuccess;
    hipLaunchKernelGGL(p3_split_kernel, dim3(start), dim3(256), 0, s, jobs);
    return hipGetLastError();
}

hipError_t launch_p3_split(const float* x, int ld, char* x3, int rows, int cols, hipStream_t s) {
    P3SplitJobs jobs{};
    jobs.n = 1;
    jobs.j[0] = P3SplitJob{x, x3, rows, cols, ld, 0};
    return launch_p3_split_group(jobs, s);
}

// ---- tile configurations -------------------------------------------------------------------------------------------
#define FBHIP_G3_CFGS(X) X(G3_128x128, 2, 2, 3) X(G3_128x64, 2, 1, 3) X(G3_64x128, 1, 2, 3) X(G3_64x64, 1, 1, 3)

int gemm3_cfg_bm(int cfg) {
    switch (cfg) {
#define X(id, tm, tn, s) case id: return 64 * tm;
        FBHIP_G3_CFGS(X)
#undef X
        default: return 64;
    }
}
int gemm3_cfg_bn(int cfg) {
    switch (cfg) {
#define X(id, tm, tn, s) case id: return 64 * tn;
        FBHIP_G3_CFGS(X)
#undef X
        default: return 64;
    }
}

hipError_t gemm3_init() {
    static bool done = false;
    if (done) return hipSuccess;
#define X(id, tm, tn, s)                                                                                           \
    {                                                                                                              \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm3_kernel<tm, tn, s>),                \
                                           hipFuncAttributeMaxDynamicSharedMemorySize,                             \
                                           (int)G3Geom<tm, tn, s>::LDS_BYTES);                                     \
        if (e != hipSuccess) return e;                                                                             \
    }
    FBHIP_G3_CFGS(X)
#undef X
    done = true;
    return hipSuccess;
}

// An operand is staged either from its P3 image (whole 192-byte blocks: leading dimension a multiple of 32 floats) or from fp32
// (16-byte aligned rows); K is a multiple of the 32-deep chunk either way
bool gemm3_problem_ok(const Gemm3Problem& p) {
    auto operand_ok = [](const float* x, const char* x3, int ld, int kcontig, int nr) {
        if (x3 != nullptr) return (ld & 31) == 0 && ((uintptr_t)x3 & 15) == 0 && (kcontig || (nr & 31) == 0 || ld >= ((nr + 31) & ~31));
        return x != nullptr && ((uintptr_t)x & 15) == 0 && (ld & 3) == 0 && ld >= 4 && (kcontig || nr >= 4);
    };
    return operand_ok(p.A, p.A3, p.lda, p.a_kcontig, p.M) && operand_ok(p.B, p.B3, p.ldb, p.b_kcontig, p.N) && p.K >= 32 && (p.K & 31) == 0 &&
           (p.C3 == nullptr || ((p.ldc & 31) == 0 && ((uintptr_t)p.C3 & 15) == 0));
}

void gemm3_problem_finalize(Gemm3Problem& p, int cfg) {
    if (p.kslices < 1) p.kslices = 1;
    if (p.kslices == 1) p.kper = p.K / 32;
    const int BM = gemm3_cfg_bm(cfg), BN = gemm3_cfg_bn(cfg);
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
}

hipError_t launch_gemm3_group(const Gemm3Group& g, int cfg, hipStream_t stream) {
    if (g.total_tiles <= 0) return hipSuccess;
    for (int i = 0; i < g.n; ++i)
        if (!gemm3_problem_ok(g.p[i])) return hipErrorInvalidValue;
    switch (cfg) {
#define X(id, tm, tn, s)                                                                                                  \
    case id:                                                                                                               \
        hipLaunchKernelGGL((gemm3_kernel<tm, tn, s>), dim3(g.total_tiles), dim3(512), (G3Geom<tm, tn, s>::LDS_BYTES), stream, g); \
        break;
        FBHIP_G3_CFGS(X)
#undef X
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace fbhip
