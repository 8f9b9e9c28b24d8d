// Diagnostic (not part of the product): cycle-stamp trace of the grouped GEMM kernel's steady loop.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DFBHIP_TRACE -Iinclude -Icontrollable_agent_amd/csrc tools/gemm_trace.hip -o tools/scratch/gemm_trace
//   tools/scratch/gemm_trace M N K [groups]
#include "gemm.hip"

#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>

using namespace fbhip;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 1024, N = argc > 2 ? atoi(argv[2]) : 2048, K = argc > 3 ? atoi(argv[3]) : 1024;
    const int G = argc > 4 ? atoi(argv[4]) : 1;
    float *A, *B, *C;
    CK(hipMalloc(&A, (size_t)G * M * K * 4)); CK(hipMalloc(&B, (size_t)G * N * K * 4)); CK(hipMalloc(&C, (size_t)G * M * N * 4));
    std::vector<float> h((size_t)G * std::max(M, N) * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    CK(hipMemcpy(A, h.data(), (size_t)G * M * K * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, h.data(), (size_t)G * N * K * 4, hipMemcpyHostToDevice));
    CK(gemm_init());
    GemmGroup g{};
    int start = 0;
    for (int i = 0; i < G; ++i) {
        GemmProblem p{};
        p.A = A + (size_t)i * M * K; p.B = B + (size_t)i * N * K; p.C = C + (size_t)i * M * N;
        p.M = M; p.N = N; p.K = K; p.lda = K; p.ldb = K; p.ldc = N; p.a_kcontig = 1; p.b_kcontig = 1; p.epi = 0; p.kslices = 1;
        gemm_problem_finalize(p, CFG_2x2x1);
        p.tile_start = start; start += p.tiles_m * p.tiles_n;
        g.p[i] = p;
    }
    g.n = G; g.total_tiles = start;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) CK(launch_gemm_group(g, CFG_2x2x1, 0));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 20; ++i) CK(launch_gemm_group(g, CFG_2x2x1, 0));
    CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1000 / 20;
    printf("%dx%dx%d x%d: %.1f us per launch, %.1f TF/s, %d workgroups\n", M, N, K, G, us, 2.0 * G * M * N * K / us / 1e6, start);

    const int nwg = std::min(start, (int)TRACE_WGS), nt = std::min(K / 32, (int)TRACE_IT);
    std::vector<unsigned long long> tr((size_t)TRACE_WGS * 2 * TRACE_IT * TRACE_SLOTS);
    std::vector<unsigned> hw(TRACE_WGS);
    CK(hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(fbhip::g_trace), tr.size() * 8));
    CK(hipMemcpyFromSymbol(hw.data(), HIP_SYMBOL(fbhip::g_trace_hw), hw.size() * 4));
    auto T = [&](int wg, int role, int it, int slot) { return (double)tr[(((size_t)wg * 2 + role) * TRACE_IT + it) * TRACE_SLOTS + slot]; };
    // steady-state chunks 4 .. nt-4
    double c_span = 0, c_wait = 0, c_period = 0, p_issue = 0, p_land = 0, p_wait = 0; int cnt = 0;
    for (int w = 0; w < nwg; ++w)
        for (int it = 4; it < nt - 4; ++it) {
            c_span += T(w, 0, it, 1) - T(w, 0, it, 0);
            c_wait += T(w, 0, it, 2) - T(w, 0, it, 1);
            c_period += T(w, 0, it + 1, 0) - T(w, 0, it, 0);
            p_issue += T(w, 1, it, 1) - T(w, 1, it, 0);
            p_land += T(w, 1, it, 2) - T(w, 1, it, 1);
            p_wait += T(w, 1, it, 3) - T(w, 1, it, 2);
            ++cnt;
        }
    printf("steady chunks (4..%d) averaged over %d workgroups, cycles of the stamp clock:\n", nt - 4, nwg);
    printf("  consumer wave 0: chunk period %.0f = MFMA span (top -> last MFMA issued) %.0f + barrier wait %.0f + rest\n", c_period / cnt, c_span / cnt, c_wait / cnt);
    printf("  producer wave 4: load issue %.0f, landing wait + LDS store %.0f, barrier wait %.0f\n", p_issue / cnt, p_land / cnt, p_wait / cnt);
    // whole-kernel picture per workgroup
    double first = 1e30, last = 0, pro = 0, loop = 0;
    for (int w = 0; w < nwg; ++w) { first = std::min(first, T(w, 1, 0, 0)); last = std::max(last, T(w, 0, nt - 1, 2)); }
    for (int w = 0; w < nwg; ++w) { pro += T(w, 0, 0, 0) - first; loop += T(w, 0, nt - 1, 2) - T(w, 0, 0, 0); }
    printf("  first producer stamp -> last consumer stamp: %.0f cycles; mean (loop start - first stamp) %.0f, mean loop length %.0f (= %.0f per chunk)\n",
           last - first, pro / nwg, loop / nwg, loop / nwg / nt);
    // co-residency: workgroups per (xcc, se, cu)
    std::map<unsigned, std::vector<int>> cu;
    for (int w = 0; w < nwg; ++w) cu[hw[w] & 0xff00ff00u].push_back(w);   // drop wave / simd bits
    std::map<int, int> hist;
    for (auto& kv : cu) hist[(int)kv.second.size()]++;
    printf("  compute units by resident workgroups:");
    for (auto& kv : hist) printf(" %d x%d", kv.first, kv.second);
    printf("\n");
    // per-chunk listing of one CU's workgroups
    for (auto& kv : cu) {
        if (kv.second.size() < 2) continue;
        printf("  one CU (hw %08x): workgroups", kv.first);
        for (int w : kv.second) printf(" %d", w);
        printf("\n   it |");
        for (size_t j = 0; j < kv.second.size(); ++j) printf("  C.top  C.span C.wait | P.issue P.land P.wait |");
        printf("\n");
        const double t0 = T(kv.second[0], 0, 0, 0);
        for (int it = 0; it < nt; ++it) {
            printf("  %3d |", it);
            for (int w : kv.second)
                printf(" %7.0f %6.0f %6.0f | %6.0f %6.0f %6.0f |", T(w, 0, it, 0) - t0, T(w, 0, it, 1) - T(w, 0, it, 0), T(w, 0, it, 2) - T(w, 0, it, 1),
                       T(w, 1, it, 1) - T(w, 1, it, 0), T(w, 1, it, 2) - T(w, 1, it, 1), T(w, 1, it, 3) - T(w, 1, it, 2));
            printf("\n");
        }
        break;
    }
    return 0;
}
