// Round-3 root cause of the "release-build segfault" (VERDICT r02 item 2): hipGraphLaunch of a graph with two parallel branches
// dereferences past the end of the exec's parallel-stream list (hip::Graph::UpdateStreams, libamdhip64 of ROCm 7.0 / torch 2.10).
//
// What the runtime does (disassembly of UpdateStreams / GraphExec::Init / CreateStreams): an exec whose
// graph needs n > 1 streams creates n extra NORMAL-priority streams at instantiate time and, at every launch, assigns them to
// the slots 1 .. n-1, SKIPPING every extra stream that shares its hardware queue with the launch stream -- with no bound on
// the index.  One collision is survivable (n created, n - 1 needed); if TWO of the n extra streams sit on the launch stream's
// hardware queue the loop walks off the vector and the process dies.  Streams get the least-used hardware queue of their
// priority class (4 per class), so whether that happens depends on the history of stream creations / destructions in the
// process: a test process that builds and drops many contexts (each with a side stream and a few two-branch execs) hits it
// once in a while; a training run that builds one agent practically never.
//
// This program provokes it on purpose -- random stream churn, then capture / instantiate / launch of a two-branch graph -- with
// the launch stream (a) the null stream, (b) a normal-priority stream, (c) a HIGH-priority stream, whose hardware queue comes
// from another pool and can therefore never be shared with the exec's normal-priority streams.  Each mode runs in a forked
// child; the parent reports how many trials survived.   usage: graph_queue_collision [trials]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <random>
#include <sys/wait.h>
#include <unistd.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); _exit(3); } } while (0)

__global__ void touch(int* p) { atomicAdd(p, 1); }

static int run_mode(int mode, int trials, int wfd) {
    std::mt19937 rng(99 + mode);
    int* d = nullptr;
    CK(hipMalloc(&d, 4));
    CK(hipMemset(d, 0, 4));
    hipStream_t L = nullptr;
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    if (mode == 1) CK(hipStreamCreateWithFlags(&L, hipStreamNonBlocking));
    if (mode == 2) CK(hipStreamCreateWithPriority(&L, hipStreamNonBlocking, hi));
    hipStream_t side;
    CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreateWithFlags(&e0, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
    std::vector<hipStream_t> pool;
    std::vector<hipGraphExec_t> execs;
    for (int t = 0; t < trials; ++t) {
        // churn: the pool of live normal-priority streams and of live two-branch execs changes size at random
        const int ops = 1 + rng() % 6;
        for (int o = 0; o < ops; ++o) {
            if (pool.size() < 24 && (pool.empty() || rng() % 2)) {
                hipStream_t s;
                CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
                pool.push_back(s);
            } else if (!pool.empty()) {
                const size_t i = rng() % pool.size();
                CK(hipStreamDestroy(pool[i]));
                pool.erase(pool.begin() + i);
            }
        }
        while (execs.size() > 6 || (!execs.empty() && rng() % 3 == 0)) {
            const size_t i = rng() % execs.size();
            CK(hipGraphExecDestroy(execs[i]));
            execs.erase(execs.begin() + i);
        }
        hipGraph_t g = nullptr;
        CK(hipStreamBeginCapture(L, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(touch, dim3(1), dim3(1), 0, L, d);
        CK(hipEventRecord(e0, L));
        CK(hipStreamWaitEvent(side, e0, 0));
        hipLaunchKernelGGL(touch, dim3(1), dim3(1), 0, side, d);
        hipLaunchKernelGGL(touch, dim3(1), dim3(1), 0, L, d);
        CK(hipEventRecord(e1, side));
        CK(hipStreamWaitEvent(L, e1, 0));
        hipLaunchKernelGGL(touch, dim3(1), dim3(1), 0, L, d);
        CK(hipStreamEndCapture(L, &g));
        hipGraphExec_t ex = nullptr;
        CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        CK(hipGraphDestroy(g));
        CK(hipGraphLaunch(ex, L));
        CK(hipGraphLaunch(ex, L));
        CK(hipStreamSynchronize(L));
        execs.push_back(ex);
        const int done = t + 1;
        if (write(wfd, &done, sizeof(done)) != sizeof(done)) _exit(4);
    }
    int h = 0;
    CK(hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost));
    return h == 8 * trials ? 0 : 5;
}

int main(int argc, char** argv) {
    const int trials = argc > 1 ? atoi(argv[1]) : 2000;
    const char* names[3] = {"null stream", "normal-priority stream", "high-priority stream"};
    int worst = 0;
    for (int mode = 0; mode < 3; ++mode) {
        int fds[2];
        if (pipe(fds) != 0) return 9;
        const pid_t pid = fork();
        if (pid == 0) {
            close(fds[0]);
            _exit(run_mode(mode, trials, fds[1]));
        }
        close(fds[1]);
        int done = 0, v = 0;
        while (read(fds[0], &v, sizeof(v)) == sizeof(v)) done = v;
        int status = 0;
        waitpid(pid, &status, 0);
        if (WIFSIGNALED(status)) printf("launch stream = %-24s DIED with signal %d after %d of %d trials\n", names[mode], WTERMSIG(status), done, trials);
        else printf("launch stream = %-24s exit code %d, %d of %d trials\n", names[mode], WEXITSTATUS(status), done, trials);
        if (mode == 2 && (WIFSIGNALED(status) || WEXITSTATUS(status) != 0)) worst = 1;
        fflush(stdout);
    }
    return worst;
}
