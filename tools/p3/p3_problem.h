// Problem / group structs of the archived P3 GEMM (tools/p3): the product's GemmProblem plus the three image pointers.
#pragma once
#include "common.h"
#include "p3.h"

namespace fbhip {

// P3 images (p3.h) of the operands / the output: the block of element (0, 0); nullptr: stage that operand from fp32 /
// fp32 output only
struct Gemm3Problem : GemmProblem {
    const char* A3;
    const char* B3;
    char* C3;
};
struct Gemm3Group {
    Gemm3Problem p[MAX_GROUP];
    int n;
    int total_tiles;
};

enum Gemm3Cfg { G3_128x128 = 0, G3_128x64 = 1, G3_64x128 = 2, G3_64x64 = 3, G3_CFG_COUNT };
hipError_t gemm3_init();                                  // one-time kernel attribute setup (outside graph capture)
int gemm3_cfg_bm(int cfg);
int gemm3_cfg_bn(int cfg);
bool gemm3_problem_ok(const Gemm3Problem& p);             // whole blocks (ld % 32, K % 32), aligned
void gemm3_problem_finalize(Gemm3Problem& p, int cfg);    // fills tiles_* (and kper of an unsliced problem)
hipError_t launch_gemm3_group(const Gemm3Group& g, int cfg, hipStream_t stream);
// fp32 -> P3 for operands whose producer does not emit planes; views of whole 8-column groups, ld % 32 == 0
struct P3SplitJob { const float* x; char* x3; int rows, cols, ld, block_start; };
constexpr int P3_SPLIT_MAX = 8;
struct P3SplitJobs { P3SplitJob j[P3_SPLIT_MAX]; int n; };
hipError_t launch_p3_split_group(P3SplitJobs jobs, hipStream_t s);

}  // namespace fbhip
