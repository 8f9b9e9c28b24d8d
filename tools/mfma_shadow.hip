// Diagnostic (not part of the product): what can a wave issue in the shadow of its own v_mfma_f32_32x32x2_f32?
// One wave per SIMD (256 threads a workgroup, one workgroup per CU); a loop of MFMAs on ONE accumulator (dependent chain) or on
// FOUR (independent), with N filler instructions of one kind between two MFMAs; prints shader cycles per MFMA.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_shadow.hip -o tools/scratch/mfma_shadow
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float floatx16 __attribute__((ext_vector_type(16)));

#define FENCE __builtin_amdgcn_sched_barrier(0)

template <int KIND, int N, int CHAINS>
__global__ void __launch_bounds__(256) probe(unsigned long long* out, int iters, float a, float b, float* sink) {
    __shared__ __attribute__((aligned(16))) float lds[256 * 4 * 4];
    floatx16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    float x0 = a, x1 = b, x2 = a + 1, x3 = b + 1;
    float4 w = make_float4(a, b, a, b);
    float* my = lds + threadIdx.x * 4;
    const float* rd = lds + (threadIdx.x & 63);
    float r0 = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            acc[j % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j % CHAINS], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < N; ++k) {
                if (KIND == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x0) : "v"(x1));                 // dependent VALU chain
                if (KIND == 2) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(x0) : "v"(x1)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(x2) : "v"(x3)); }
                if (KIND == 3) *reinterpret_cast<float4*>(my) = w;                                          // ds_write_b128
                if (KIND == 4) { float t; asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"((unsigned)(size_t)rd)); r0 = t; }
                if (KIND == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x0) : "v"(x1));
                if (KIND == 6) asm volatile("s_nop 3");
                if (KIND == 7) asm volatile("v_accvgpr_mov_b32 a255, a254");
            }
            FENCE;
        }
        if (KIND == 4) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = x0 + x2 + r0;
    for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
    if (s == 12345.f) sink[0] = s + lds[5];
    if ((threadIdx.x & 63) == 0) out[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// Two waves per SIMD: waves 0-3 of a 512-thread workgroup run the bare dependent MFMA chain, waves 4-7 spin on ONE kind of
// instruction until the chain waves are done (flag in LDS).  What does the chain lose to a NEIGHBOUR wave's instruction stream?
template <int KIND, int CHAIN>
__global__ void __launch_bounds__(512) neighbour(unsigned long long* out, int iters, float a, float b, float* sink) {
    __shared__ __attribute__((aligned(16))) float lds[512 * 4];
    __shared__ int done;
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    const int wid = threadIdx.x >> 6;
    if (wid < 4) {
        // CHAIN 0: no MFMA (sleep); 1: one dependent chain; 2: four independent accumulators; 3 / 4: dependent chain with 2 / 3
        // s_nop 15 between two MFMAs; 5: dependent chain at s_setprio 0 against neighbours at s_setprio 3; 6: s_sleep 1 between MFMAs
        floatx16 acc, acc1, acc2, acc3;
        for (int i = 0; i < 16; ++i) { acc[i] = 0.f; acc1[i] = 0.f; acc2[i] = 0.f; acc3[i] = 0.f; }
        if (CHAIN == 5) __builtin_amdgcn_s_setprio(0);
        const unsigned long long t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (CHAIN == 0) __builtin_amdgcn_s_sleep(16);
                else if (CHAIN == 2) {
                    if ((j & 3) == 0) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                    if ((j & 3) == 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
                    if ((j & 3) == 2) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc2, 0, 0, 0);
                    if ((j & 3) == 3) acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc3, 0, 0, 0);
                } else {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                    if (CHAIN == 3) { asm volatile("s_nop 15\n s_nop 15"); }
                    if (CHAIN == 4) { asm volatile("s_nop 15\n s_nop 15\n s_nop 15"); }
                    if (CHAIN == 6) __builtin_amdgcn_s_sleep(1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        float s = 0.f;
        for (int i = 0; i < 16; ++i) s += acc[i] + acc1[i] + acc2[i] + acc3[i];
        const unsigned long long t1 = __builtin_readcyclecounter();
        if (s == 12345.f) sink[0] = s;
        if ((threadIdx.x & 63) == 0) { out[(size_t)blockIdx.x * 4 + wid] = t1 - t0; __hip_atomic_fetch_add(&done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
    } else {
        float x0 = a, x1 = b, x2 = a + 1, x3 = b + 1;
        float4 w = make_float4(a, b, a, b);
        float* my = lds + threadIdx.x * 4;
        if (CHAIN == 5) __builtin_amdgcn_s_setprio(3);
        int guard = 0;
        const unsigned long long n0 = __builtin_readcyclecounter();
        while (__hip_atomic_load(&done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < 4 && guard < (1 << 22)) {
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                if (KIND == 1) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(x0) : "v"(x1)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(x2) : "v"(x3)); }
                if (KIND == 2) { *reinterpret_cast<float4*>(my) = w; asm volatile("" ::: "memory"); }
                if (KIND == 3) asm volatile("s_nop 7");
                if (KIND == 4) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x0) : "v"(x1)); asm volatile("v_mov_b32 %0, %1" : "=v"(x2) : "v"(x3)); }
            }
            ++guard;
        }
        const unsigned long long n1 = __builtin_readcyclecounter();
        if ((threadIdx.x & 63) == 0) { out[1024 + (size_t)blockIdx.x * 8 + 2 * (wid - 4)] = n1 - n0; out[1024 + (size_t)blockIdx.x * 8 + 2 * (wid - 4) + 1] = (unsigned long long)guard; }
        if (x0 + x2 == 12345.f) sink[1] = x0 + lds[3];
    }
}

template <int KIND, int CHAIN = 1>
void run_neighbour(const char* name, unsigned long long* d, float* sink) {
    const int iters = 200, wgs = 256;
    neighbour<KIND, CHAIN><<<wgs, 512>>>(d, iters, 1.0f, 0.5f, sink);
    neighbour<KIND, CHAIN><<<wgs, 512>>>(d, iters, 1.0f, 0.5f, sink);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(wgs * 4);
    (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    std::vector<unsigned long long> nb(wgs * 8);
    (void)hipMemcpy(nb.data(), d + 1024, nb.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> per;
    for (int i = 0; i < wgs * 4; ++i) if (nb[2 * i + 1] > 0) per.push_back((double)nb[2 * i] / ((double)nb[2 * i + 1] * 32.0));
    std::sort(per.begin(), per.end());
    printf("neighbour wave spinning on %-18s: %7.1f cycles per MFMA of the chain wave; the neighbour: %6.1f cycles per loop slot (median)\n", name,
           (double)h[h.size() / 2] / (iters * 16.0), per.empty() ? 0.0 : per[per.size() / 2]);
}

template <int KIND, int N, int CHAINS>
void run(const char* name, unsigned long long* d, float* sink) {
    const int iters = 200, wgs = 256;
    probe<KIND, N, CHAINS><<<wgs, 256>>>(d, iters, 1.0f, 0.5f, sink);
    probe<KIND, N, CHAINS><<<wgs, 256>>>(d, iters, 1.0f, 0.5f, sink);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(wgs * 4);
    (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-28s N=%2d chains=%d: %7.1f cycles per MFMA (median wave)\n", name, N, CHAINS, (double)h[h.size() / 2] / (iters * 16.0));
}

int main() {
    unsigned long long* d; float* sink;
    (void)hipMalloc(&d, (1024 + 2048) * 8); (void)hipMalloc(&sink, 16);
    run<0, 0, 1>("bare", d, sink);
    run<0, 0, 4>("bare", d, sink);
    run<1, 4, 1>("v_add dependent", d, sink);  run<1, 8, 1>("v_add dependent", d, sink);  run<1, 12, 1>("v_add dependent", d, sink);
    run<1, 8, 4>("v_add dependent", d, sink);  run<1, 16, 4>("v_add dependent", d, sink);
    run<2, 4, 1>("v_add 2 chains (x2)", d, sink); run<2, 6, 1>("v_add 2 chains (x2)", d, sink);
    run<5, 8, 1>("v_cndmask", d, sink); run<5, 12, 1>("v_cndmask", d, sink);
    run<3, 1, 1>("ds_write_b128", d, sink); run<3, 2, 1>("ds_write_b128", d, sink); run<3, 1, 4>("ds_write_b128", d, sink);
    run<4, 1, 1>("ds_read_b32", d, sink); run<4, 2, 1>("ds_read_b32", d, sink); run<4, 4, 1>("ds_read_b32", d, sink);
    run<6, 4, 1>("s_nop 3", d, sink); run<6, 8, 1>("s_nop 3", d, sink); run<6, 12, 1>("s_nop 3", d, sink);
    run<7, 8, 1>("v_accvgpr_mov", d, sink); run<7, 16, 1>("v_accvgpr_mov", d, sink);
    run_neighbour<0>("nothing (flag poll)", d, sink);
    run_neighbour<1>("v_add_f32", d, sink);
    run_neighbour<4>("v_cndmask / v_mov", d, sink);
    run_neighbour<2>("ds_write_b128", d, sink);
    run_neighbour<3>("s_nop 7", d, sink);
    run_neighbour<1, 0>("v_add_f32 (NO chain)", d, sink);
    run_neighbour<2, 0>("ds_write (NO chain)", d, sink);
    run_neighbour<1, 2>("v_add_f32 (4 accs)", d, sink);
    run_neighbour<2, 2>("ds_write (4 accs)", d, sink);
    run_neighbour<1, 3>("v_add_f32 (2 nops)", d, sink);
    run_neighbour<1, 4>("v_add_f32 (3 nops)", d, sink);
    run_neighbour<2, 4>("ds_write (3 nops)", d, sink);
    run_neighbour<1, 5>("v_add_f32 (setprio)", d, sink);
    run_neighbour<2, 5>("ds_write (setprio)", d, sink);
    run_neighbour<1, 6>("v_add_f32 (s_sleep 1)", d, sink);
    return 0;
}
