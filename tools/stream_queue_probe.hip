// Diagnostic only (round 3, the hipGraph parallel-stream crash): prints, for live streams, the value hip::Graph::UpdateStreams
// compares -- stream object + 0x1a8 -> virtual slot 2 (libamdhip64 of this image; offsets from its disassembly) -- to see what it
// is (a hardware-queue identity?) and how the runtime hands it out as streams come and go.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void nop() {}
static uintptr_t qid(hipStream_t s) {
    if (!s) return 0;
    void* obj = *(void**)((char*)s + 0x1a8);
    if (!obj) return 1;
    void** vt = *(void***)obj;
    auto fn = (uintptr_t (*)(void*))vt[2];
    return fn(obj);
}
int main() {
    CK(hipSetDevice(0));
    int lo, hi;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    std::vector<hipStream_t> v;
    for (int i = 0; i < 12; ++i) { hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); v.push_back(s); printf("create normal #%d -> %lx\n", i, (unsigned long)qid(s)); }
    for (int i = 0; i < 4; ++i) { hipStream_t s; CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi)); printf("create high #%d -> %lx\n", i, (unsigned long)qid(s)); CK(hipStreamDestroy(s)); }
    for (int i = 0; i < 12; ++i) hipLaunchKernelGGL(nop, dim3(1), dim3(1), 0, v[i]);
    CK(hipDeviceSynchronize());
    for (int i = 0; i < 12; ++i) printf("after use+sync normal #%d -> %lx\n", i, (unsigned long)qid(v[i]));
    // destroy the streams that share #0's value except #0 itself, then create new ones: where do they go?
    const uintptr_t q0 = qid(v[0]);
    int destroyed = 0;
    for (int i = 11; i >= 1; --i) if (qid(v[i]) == q0) { CK(hipStreamDestroy(v[i])); v[i] = nullptr; ++destroyed; }
    printf("destroyed %d streams sharing #0's value\n", destroyed);
    for (int i = 0; i < 4; ++i) { hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); printf("new stream %d -> %lx %s\n", i, (unsigned long)qid(s), qid(s) == q0 ? "(== #0)" : ""); }
    return 0;
}
