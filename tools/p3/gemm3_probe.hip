// Standalone probe of the P3 GEMM kernel (controllable_agent_amd/csrc/gemm3_kernel.h): correctness against an fp64 host
// reference (sampled entries + the P3 image of C) and timing per shape / orientation / tile configuration, each shape alone and
// as a 4-problem group.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I tools/p3 -I controllable_agent_amd/csrc -I include
//                               tools/p3/gemm3_probe.hip -o tools/scratch/gemm3_probe
// -DG3_KNOCK: the knock-out build (tools/scratch/gemm3_probe_knock); the clean build is the one to quote times from
#include "gemm3_kernel.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

using namespace fbhip;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void split_kernel(const float* __restrict__ x, int ld, char* __restrict__ x3, int rows, int cols) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int per_row = cols / 2;
    if (i >= rows * per_row) return;
    const int r = i / per_row, c = 2 * (i % per_row);
    const float2 v = *reinterpret_cast<const float2*>(x + (size_t)r * ld + c);
    const P3Triple a = p3_split(v.x), b = p3_split(v.y);
    char* d = x3 + p3_offset(r, c, ld);
    *reinterpret_cast<unsigned*>(d) = (unsigned)a.h | ((unsigned)b.h << 16);
    *reinterpret_cast<unsigned*>(d + 64) = (unsigned)a.m | ((unsigned)b.m << 16);
    *reinterpret_cast<unsigned*>(d + 128) = (unsigned)a.l | ((unsigned)b.l << 16);
}

struct Mat {
    int rows, cols, ld;
    std::vector<float> h;
    float* d = nullptr;
    char* d3 = nullptr;
};

static Mat make(int rows, int cols, std::mt19937& rng, float scale = 1.f) {
    Mat m;
    m.rows = rows; m.cols = cols; m.ld = (cols + 31) & ~31;
    m.h.assign((size_t)rows * m.ld, 0.f);
    std::normal_distribution<float> nd(0.f, scale);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) m.h[(size_t)r * m.ld + c] = nd(rng);
    CK(hipMalloc(&m.d, m.h.size() * 4));
    CK(hipMalloc(&m.d3, m.h.size() * 6));
    CK(hipMemcpy(m.d, m.h.data(), m.h.size() * 4, hipMemcpyHostToDevice));
    const int n = rows * (m.ld / 2);
    hipLaunchKernelGGL(split_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, m.d, m.ld, m.d3, rows, m.ld);
    CK(hipDeviceSynchronize());
    return m;
}

static Mat make_out(int rows, int cols) {
    Mat m;
    m.rows = rows; m.cols = cols; m.ld = (cols + 31) & ~31;
    m.h.assign((size_t)rows * m.ld, 0.f);
    CK(hipMalloc(&m.d, m.h.size() * 4));
    CK(hipMalloc(&m.d3, m.h.size() * 6));
    CK(hipMemset(m.d, 0, m.h.size() * 4));
    CK(hipMemset(m.d3, 0, m.h.size() * 6));
    return m;
}

template <int TM, int TN, int S>
static void launch(const Gemm3Group& g, hipStream_t s) {
    using G = G3Geom<TM, TN, S>;
    static bool init = false;
    if (!init) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm3_kernel<TM, TN, S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES));
        init = true;
    }
    hipLaunchKernelGGL((gemm3_kernel<TM, TN, S>), dim3(g.total_tiles), dim3(512), G::LDS_BYTES, s, g);
}

static void launch_cfg(int cfg, const Gemm3Group& g, hipStream_t s) {
    switch (cfg) {
        case 0: launch<2, 2, 3>(g, s); break;
        case 1: launch<2, 1, 3>(g, s); break;
        case 2: launch<2, 1, 2>(g, s); break;
        case 3: launch<1, 1, 3>(g, s); break;
        case 4: launch<2, 2, 2>(g, s); break;
        case 5: launch<1, 2, 3>(g, s); break;
        case 6: launch<2, 1, 4>(g, s); break;
        default: exit(2);
    }
}
static const char* cfg_name[] = {"128x128 S3", "128x64 S3", "128x64 S2", "64x64 S3", "128x128 S2", "64x128 S3", "128x64 S4"};
static const int cfg_bm[] = {128, 128, 128, 64, 128, 64, 128};
static const int cfg_bn[] = {128, 64, 64, 64, 128, 128, 64};

// mode: 3 = NT (A [M,K], B [N,K]), 2 = NN (A [M,K], B [K,N]), 0 = TN (A [K,M], B [K,N])
struct Case { int M, N, K, mode, epi; int pa = 1, pb = 1; };      // pa / pb: stage the operand from its image (1) or from fp32 (0)

static Gemm3Problem problem(const Case& c, const Mat& A, const Mat& B, Mat& C, const float* bias, const float* aux, int ldaux,
                           float* colsum, int cfg) {
    Gemm3Problem p{};
    p.A = A.d; p.B = B.d; p.C = C.d; p.A3 = c.pa ? A.d3 : nullptr; p.B3 = c.pb ? B.d3 : nullptr; p.C3 = c.pa && c.pb ? C.d3 : nullptr;
    p.bias = bias; p.aux = aux; p.ldaux = ldaux; p.colsum = colsum;
    p.M = c.M; p.N = c.N; p.K = c.K; p.lda = A.ld; p.ldb = B.ld; p.ldc = C.ld;
    p.a_kcontig = (c.mode >> 1) & 1; p.b_kcontig = c.mode & 1; p.epi = c.epi;
    p.kslices = 1; p.kper = c.K / 32;
    p.tiles_m = (c.M + cfg_bm[cfg] - 1) / cfg_bm[cfg]; p.tiles_n = (c.N + cfg_bn[cfg] - 1) / cfg_bn[cfg];
    return p;
}

static double ref_entry(const Case& c, const Mat& A, const Mat& B, int m, int n) {
    double s = 0;
    for (int k = 0; k < c.K; ++k) {
        const double a = (c.mode & 2) ? A.h[(size_t)m * A.ld + k] : A.h[(size_t)k * A.ld + m];
        const double b = (c.mode & 1) ? B.h[(size_t)n * B.ld + k] : B.h[(size_t)k * B.ld + n];
        s += a * b;
    }
    return s;
}

static void time_case(const Case& c, int group, int cfg, int iters, int knock, std::mt19937& rng, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    std::vector<Mat> As, Bs, Cs;
    for (int q = 0; q < group; ++q) {
        As.push_back((c.mode & 2) ? make(c.M, c.K, rng) : make(c.K, c.M, rng));
        Bs.push_back((c.mode & 1) ? make(c.N, c.K, rng) : make(c.K, c.N, rng));
        Cs.push_back(make_out(c.M, c.N));
    }
    Mat bias = make(1, c.N, rng), aux = make(c.M, c.N, rng, 0.6f);
#ifdef G3_KNOCK
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g3_knock_mask), &knock, sizeof(int)));
#else
    if (knock) { printf("(clean build: no knock-outs)\n"); return; }
#endif
    Gemm3Group g{};
    int start = 0;
    for (int q = 0; q < group; ++q) {
        g.p[q] = problem(c, As[q], Bs[q], Cs[q], bias.d, aux.d, aux.ld, nullptr, cfg);
        g.p[q].tile_start = start;
        start += g.p[q].tiles_m * g.p[q].tiles_n;
    }
    g.n = group; g.total_tiles = start;
    for (int w = 0; w < 3; ++w) launch_cfg(cfg, g, s);
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) launch_cfg(cfg, g, s);
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters, tf = 2.0 * c.M * c.N * c.K * group / (us * 1e-6) * 1e-12;
    printf("single %4dx%4dx%4d mode %d x%d  %-11s knock %d images %d%d  %4d wgs  %8.1f us  %6.1f TFLOP/s fp32-equivalent\n", c.M, c.N, c.K, c.mode, group,
           cfg_name[cfg], knock, c.pa, c.pb, start, us, tf);
#ifdef G3_KNOCK
    const int zero = 0;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g3_knock_mask), &zero, sizeof(int)));
#endif
}

int main(int argc, char** argv) {
    std::mt19937 rng(1234);
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    if (argc > 1 && !strcmp(argv[1], "single")) {        // single M N K mode group cfg iters knock
        Case c{atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), EPI_BIAS_RELU};
        if (argc > 11) { c.pa = atoi(argv[10]); c.pb = atoi(argv[11]); }
        time_case(c, atoi(argv[6]), atoi(argv[7]), atoi(argv[8]), argc > 9 ? atoi(argv[9]) : 0, rng, s, e0, e1);
        return 0;
    }
    const int iters = argc > 1 ? atoi(argv[1]) : 20;

    // ---------------------------------------------------------------- correctness
    const Case checks[] = {{256, 256, 128, 3, EPI_BIAS_RELU}, {256, 192, 96, 3, EPI_BIAS}, {200, 160, 64, 3, EPI_NONE},
                           {256, 256, 128, 2, EPI_MASK_RELU}, {192, 320, 160, 2, EPI_TANH_BWD}, {256, 128, 256, 0, EPI_NONE},
                           {320, 96, 192, 0, EPI_NONE},      {1024, 2048, 1024, 3, EPI_BIAS_RELU}, {1024, 1024, 2048, 2, EPI_MASK_RELU},
                           {2048, 1024, 1024, 0, EPI_NONE}};
    int bad = 0;
    for (const Case& c0 : checks) {
      for (int om = 0; om < 3; ++om) {
        Case c = c0;
        c.pa = om == 0; c.pb = om != 2;                        // images for both | A from fp32 | both from fp32
        for (int cfg = 0; cfg < 7; ++cfg) {
            Mat A = (c.mode & 2) ? make(c.M, c.K, rng) : make(c.K, c.M, rng);
            Mat B = (c.mode & 1) ? make(c.N, c.K, rng) : make(c.K, c.N, rng);
            Mat C = make_out(c.M, c.N);
            Mat bias = make(1, c.N, rng), aux = make(c.M, c.N, rng, 0.6f);
            float* colsum = nullptr;
            if (c.mode == 0) CK(hipMalloc(&colsum, c.M * 4));
            Gemm3Group g{};
            g.p[0] = problem(c, A, B, C, bias.d, aux.d, aux.ld, colsum, cfg);
            g.p[0].tile_start = 0; g.n = 1; g.total_tiles = g.p[0].tiles_m * g.p[0].tiles_n;
            launch_cfg(cfg, g, s);
            CK(hipStreamSynchronize(s));
            std::vector<float> hc(C.h.size());
            std::vector<unsigned short> h3(C.h.size() * 3);
            CK(hipMemcpy(hc.data(), C.d, hc.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(h3.data(), C.d3, h3.size() * 2, hipMemcpyDeviceToHost));
            double num = 0, den = 0, p3err = 0;
            std::uniform_int_distribution<int> um(0, c.M - 1), un(0, c.N - 1);
            for (int t = 0; t < 400; ++t) {
                const int m = t < 4 ? (t & 1 ? c.M - 1 : 0) : um(rng), n = t < 4 ? (t & 2 ? c.N - 1 : 0) : un(rng);
                double r = ref_entry(c, A, B, m, n);
                if (c.epi == EPI_BIAS) r += bias.h[n];
                else if (c.epi == EPI_BIAS_RELU) r = std::max(r + (double)bias.h[n], 0.0);
                else if (c.epi == EPI_MASK_RELU) r = aux.h[(size_t)m * aux.ld + n] > 0.f ? r : 0.0;
                else if (c.epi == EPI_TANH_BWD) { const double y = aux.h[(size_t)m * aux.ld + n]; r *= (1.0 - y * y); }
                const double got = hc[(size_t)m * C.ld + n];
                num += (got - r) * (got - r); den += r * r;
                const size_t o = p3_offset(m, n, C.ld) / 2;
                auto bf = [&](unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return (double)f; };
                const double rec = bf(h3[o]) + bf(h3[o + 32]) + bf(h3[o + 64]);
                if (c.pa && c.pb) p3err = std::max(p3err, std::fabs(rec - got) / (std::fabs(got) + 1e-30));
            }
            double cserr = 0;
            if (colsum) {
                std::vector<float> hs(c.M);
                CK(hipMemcpy(hs.data(), colsum, c.M * 4, hipMemcpyDeviceToHost));
                for (int m = 0; m < c.M; m += 7) {
                    double r = 0, ra = 0;
                    for (int k = 0; k < c.K; ++k) { r += A.h[(size_t)k * A.ld + m]; ra += std::fabs(A.h[(size_t)k * A.ld + m]); }
                    cserr = std::max(cserr, std::fabs(hs[m] - r) / (ra + 1e-30));
                }
            }
            const double rel = std::sqrt(num / (den + 1e-300));
            const bool ok = rel < 2e-6 && p3err < 2e-7 && cserr < 1e-6;
            if (!ok) ++bad;
            printf("check %4dx%4dx%4d mode %d epi %d images %d%d  %-11s rel L2 err %.3e  P3(C) max rel %.2e  colsum %.2e  %s\n", c.M, c.N, c.K, c.mode, c.epi,
                   c.pa, c.pb, cfg_name[cfg], rel, p3err, cserr, ok ? "ok" : "FAIL");
            CK(hipFree(A.d)); CK(hipFree(A.d3)); CK(hipFree(B.d)); CK(hipFree(B.d3)); CK(hipFree(C.d)); CK(hipFree(C.d3));
            CK(hipFree(bias.d)); CK(hipFree(bias.d3)); CK(hipFree(aux.d)); CK(hipFree(aux.d3));
            if (colsum) CK(hipFree(colsum));
        }
      }
    }
    printf("correctness: %d failing\n", bad);

    // ---------------------------------------------------------------- timing
    const Case shapes[] = {{1024, 2048, 1024, 3, EPI_BIAS_RELU}, {1024, 1024, 1024, 3, EPI_BIAS_RELU}, {1024, 512, 1024, 3, EPI_BIAS_RELU},
                           {1024, 1024, 2048, 2, EPI_MASK_RELU}, {1024, 1024, 512, 2, EPI_NONE},        {2048, 1024, 1024, 0, EPI_NONE},
                           {512, 1024, 1024, 0, EPI_NONE},       {1024, 1024, 96, 3, EPI_BIAS},           {4096, 4096, 1024, 3, EPI_NONE},
                           {4096, 4096, 4096, 3, EPI_NONE}};
    for (const Case& c : shapes) {
        for (int group = 1; group <= 4; group += 3) {
            if (group > 1 && c.M > 2048) continue;
            std::vector<Mat> As, Bs, Cs;
            for (int q = 0; q < group; ++q) {
                As.push_back((c.mode & 2) ? make(c.M, c.K, rng) : make(c.K, c.M, rng));
                Bs.push_back((c.mode & 1) ? make(c.N, c.K, rng) : make(c.K, c.N, rng));
                Cs.push_back(make_out(c.M, c.N));
            }
            Mat bias = make(1, c.N, rng), aux = make(c.M, c.N, rng, 0.6f);
            for (int cfg = 0; cfg < 7; ++cfg) {
                Gemm3Group g{};
                int start = 0;
                for (int q = 0; q < group; ++q) {
                    g.p[q] = problem(c, As[q], Bs[q], Cs[q], bias.d, aux.d, aux.ld, nullptr, cfg);
                    g.p[q].tile_start = start;
                    start += g.p[q].tiles_m * g.p[q].tiles_n;
                }
                g.n = group; g.total_tiles = start;
                for (int w = 0; w < 3; ++w) launch_cfg(cfg, g, s);
                CK(hipStreamSynchronize(s));
                CK(hipEventRecord(e0, s));
                for (int i = 0; i < iters; ++i) launch_cfg(cfg, g, s);
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                const double us = ms * 1e3 / iters, tf = 2.0 * c.M * c.N * c.K * group / (us * 1e-6) * 1e-12;
                printf("time  %4dx%4dx%4d mode %d x%d  %-11s %4d wgs  %8.1f us  %6.1f TFLOP/s fp32-equivalent\n", c.M, c.N, c.K, c.mode, group,
                       cfg_name[cfg], start, us, tf);
            }
            for (auto& m : As) { CK(hipFree(m.d)); CK(hipFree(m.d3)); }
            for (auto& m : Bs) { CK(hipFree(m.d)); CK(hipFree(m.d3)); }
            for (auto& m : Cs) { CK(hipFree(m.d)); CK(hipFree(m.d3)); }
            CK(hipFree(bias.d)); CK(hipFree(bias.d3)); CK(hipFree(aux.d)); CK(hipFree(aux.d3));
        }
    }
    return bad ? 1 : 0;
}
