// Diagnostic (not part of the product): what clock does the chip hold under a sustained fp32 MFMA load?
// Each wave runs a dependent chain of v_mfma_f32_32x32x2_f32 and stamps it with the shader clock (s_memtime) and the
// 100 MHz reference clock (s_memrealtime).   hipcc --offload-arch=gfx950 -O3 tools/mfma_clock.hip -o tools/scratch/mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float floatx16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(512) chain(unsigned long long* out, int n, float a, float b) {
    floatx16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i];
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        const size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 64;
        out[2 * w] = t1 - t0 + (s == 12345.f);
        out[2 * w + 1] = r1 - r0;
    }
}

int main() {
    unsigned long long* d;
    const int maxw = 256 * 8 * 8;
    hipMalloc(&d, maxw * 16);
    const int n = 1 << 15;
    struct Cfg { int wgs, threads; const char* what; } cfgs[] = {
        {8, 256, "8 workgroups x 4 waves (a few CUs busy)"},
        {256, 256, "256 workgroups x 4 waves (one wave per SIMD, every CU)"},
        {256, 512, "256 workgroups x 8 waves (two waves per SIMD)"},
        {512, 512, "512 workgroups x 8 waves (two workgroups per CU, four waves per SIMD)"},
    };
    for (auto& c : cfgs) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(chain, dim3(c.wgs), dim3(c.threads), 0, 0, d, n, 1.0f, 0.5f);
            hipEventRecord(e1, 0); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const int waves = c.wgs * c.threads / 64;
            std::vector<unsigned long long> h(2 * waves);
            hipMemcpy(h.data(), d, waves * 16, hipMemcpyDeviceToHost);
            double cyc = 0, ref = 0;
            for (int w = 0; w < waves; ++w) { cyc += h[2 * w]; ref += h[2 * w + 1]; }
            cyc /= waves; ref /= waves;
            const double mhz = cyc / (ref / 100.0);
            const double tf = (double)waves * n * 4096 / (ms * 1e-3) / 1e12;
            if (rep == 1)
                printf("%-75s %7.1f shader cycles per MFMA per wave, shader clock %6.0f MHz, kernel %.1f us, %.1f TFLOP/s\n", c.what, cyc / n, mhz, ms * 1000, tf);
        }
    }
    return 0;
}
