// P3: the three-plane bf16 image of an fp32 tensor, the operand format of the grouped GEMM of gemm3.hip.
//
// x = hi + mid + lo with hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid) (round to nearest even; the two
// residuals are exact in fp32), so hi + mid + lo carries 24 significant bits and a product a.b is reproduced to fp32
// accuracy by six bf16 MFMA products (hi.hi, hi.mid, mid.hi, mid.mid, hi.lo, lo.hi; the dropped terms are <= 2^-24 relative)
// accumulated in fp32 -- at 6 x 32 matrix-pipe cycles per 32x32x16 block instead of 8 x 64 for v_mfma_f32_32x32x2_f32.
//
// Layout: 32 consecutive floats (one 128-byte block, 128-byte aligned inside its buffer) map to ONE 192-byte block
//   [ hi x 32 | mid x 32 | lo x 32 ]   (bf16)
// at 1.5 x the fp32 byte offset inside a shadow buffer of 1.5 x the size.  Every fp32 address of a shadowed buffer therefore
// has a P3 address by arithmetic alone (no per-tensor bookkeeping): rows keep their leading dimension (a multiple of 32
// floats), sub-views start at multiples of 32 columns.  Whichever way a GEMM walks the matrix -- k along a row (forward,
// the A side of dgrad) or k down the rows (wgrad, the B side of dgrad) -- a tile is made of whole 192-byte blocks, so the
// LDS-DMA staging reads 192 contiguous bytes per block in both orientations.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fbhip {

constexpr int P3_BLOCK = 32;            // floats per block
constexpr int P3_BLOCK_BYTES = 192;     // 3 planes x 32 bf16

struct P3Triple { unsigned short h, m, l; };

__device__ __forceinline__ unsigned short p3_bf16_bits(float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ float p3_bf16_float(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }

__device__ __forceinline__ P3Triple p3_split(float x) {
    P3Triple t;
    t.h = p3_bf16_bits(x);
    const float r1 = x - p3_bf16_float(t.h);
    t.m = p3_bf16_bits(r1);
    const float r2 = r1 - p3_bf16_float(t.m);
    t.l = p3_bf16_bits(r2);
    return t;
}

// byte offset of element (row, col) of a row-major [.][ld] matrix (ld % 32 == 0) inside its P3 image, plane 0
__host__ __device__ __forceinline__ size_t p3_offset(size_t row, size_t col, size_t ld) {
    return (row * (ld / P3_BLOCK) + col / P3_BLOCK) * P3_BLOCK_BYTES + (col % P3_BLOCK) * 2;
}

// fp32 -> P3 for a [rows][cols] view (cols % 32 == 0 blocks are written whole; ld % 32 == 0): one thread per pair of elements
hipError_t launch_p3_split(const float* x, int ld, char* x3, int rows, int cols, hipStream_t s);

}  // namespace fbhip
