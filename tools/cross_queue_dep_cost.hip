// What one fork + join between two streams costs INSIDE a hipGraph (DESIGN.md section 6 "Runtime configuration", section 7): a graph
// of N rounds [kernel on L] -> fork -> [kernel on side || kernel on L] -> join -> [kernel on L] against the same 4 N tiny kernels on ONE
// stream (where nothing overlaps, so the branched form could be one kernel time per round FASTER if forks and joins were free).
// Run it under the runtime's default and under ROC_CPU_WAIT_FOR_SIGNAL=1 (two processes), optionally with extra live streams:
//   cross_queue_dep_cost [rounds=200] [extra_streams=0] [launch_stream_priority: 0 normal | 1 high]
// (link against / LD_PRELOAD the HIP runtime of the process you care about: torch 2.10 bundles ROCm 7.0's)
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); exit(3); } } while (0)

__global__ void tiny(float* p) { p[threadIdx.x] += 1.0f; }

static double time_graph(hipGraphExec_t ex, hipStream_t L, int launches) {
    CK(hipGraphLaunch(ex, L));
    CK(hipStreamSynchronize(L));
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < launches; ++i) CK(hipGraphLaunch(ex, L));
    CK(hipStreamSynchronize(L));
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / launches;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 200, extra = argc > 2 ? atoi(argv[2]) : 0, high = argc > 3 ? atoi(argv[3]) : 1;
    float* d = nullptr;
    CK(hipMalloc(&d, 4096));
    CK(hipMemset(d, 0, 4096));
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t L, side;
    CK(hipStreamCreateWithPriority(&L, hipStreamNonBlocking, high ? hi : lo));
    CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    std::vector<hipStream_t> others(extra);
    for (auto& s : others) { CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d + 512); }
    CK(hipDeviceSynchronize());
    std::vector<hipEvent_t> ev(2 * rounds);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipGraph_t g = nullptr;
    hipGraphExec_t branched = nullptr, linear = nullptr;
    CK(hipStreamBeginCapture(L, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < rounds; ++i) {
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, L, d);
        CK(hipEventRecord(ev[2 * i], L));
        CK(hipStreamWaitEvent(side, ev[2 * i], 0));
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, side, d + 256);
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, L, d);             // (runs beside the side stream's kernel: a real branch, not a chain)
        CK(hipEventRecord(ev[2 * i + 1], side));
        CK(hipStreamWaitEvent(L, ev[2 * i + 1], 0));
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, L, d);
    }
    CK(hipStreamEndCapture(L, &g));
    CK(hipGraphInstantiate(&branched, g, nullptr, nullptr, 0));
    CK(hipGraphDestroy(g));
    CK(hipStreamBeginCapture(L, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 4 * rounds; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, L, d);
    CK(hipStreamEndCapture(L, &g));
    CK(hipGraphInstantiate(&linear, g, nullptr, nullptr, 0));
    CK(hipGraphDestroy(g));
    const double tb = time_graph(branched, L, 20), tl = time_graph(linear, L, 20);
    const char* cw = getenv("ROC_CPU_WAIT_FOR_SIGNAL");
    printf("ROC_CPU_WAIT_FOR_SIGNAL=%s  launch stream %s priority, %d extra live streams, %d rounds: branched %.1f us (%.2f us / round), "
           "one stream %.1f us (%.2f us / 4 kernels): branched - one stream = %+.2f us / round\n",
           cw ? cw : "(unset)", high ? "high" : "normal", extra, rounds, tb, tb / rounds, tl, tl / rounds, (tb - tl) / rounds);
    return 0;
}
