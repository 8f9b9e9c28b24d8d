// Peer-access sum-all-reduce of a flat fp32 gradient bucket over the ranks of ONE node, as plain kernels that can be captured
// into the update's hipGraph ("keep a hand-written peer-write all-reduce kernel as the fallback").  The
// reference has no distributed code; RCCL through torch.distributed stays the default transport of the data-parallel
// schedule (distributed.py) -- but a collective issued from the host cuts the step's graph into pieces (three graph launches
// and three c10d calls per step).  With the peers' buckets mapped into this process (hipIpc; the host does the mapping,
// fbhip_dp_bind_peers only receives the pointers) the whole data-parallel step is ONE graph launch per rank.
//
// Two-shot direct all-reduce for a fully connected xGMI node (every pair of the 8 GPUs has its own link, ~50 GB/s each way):
//   K1  reduce-scatter  rank r sums chunk r of EVERY rank's bucket (W - 1 remote reads + 1 local, in rank order 0..W-1: the
//                       same order on every rank and every replay -> deterministic, and chunk r is computed once, so the
//                       replicas end bit-identical by construction) and writes it over chunk r of its own bucket
//   K2  all-gather      rank r copies the reduced chunk q from rank q, for every q != r
//   K3  release         nobody may overwrite its bucket (the next step's backward) before every peer has finished K2
// Each kernel opens with a cross-rank barrier: the local rank STORES its epoch into slot [kernel][rank] of every peer's flag
// array (system scope, after a system-scope release fence) and every workgroup then polls the W slots of its OWN array (relaxed,
// system scope; one acquire fence once they all arrived) -- signal first, wait second, no circular wait.  Grid-wide ordering
// inside a rank comes from the three kernel boundaries, so no workgroup ever waits for another workgroup of the same launch
// (residency-independent: the kernels run beside the next step's sampling / forward passes on the other stream).  Epochs are
// device-resident and advance with every replay, flags only ever grow, nothing is reset.  Every spin is bounded: a rank that
// does not see its peers within ~2^28 polls writes PEER_TIMEOUT into its status word and carries on (the host checks it);
// a lost peer never hangs the GPU.
//
// Visibility.  A bucket is written by earlier kernels of the same stream (kernel boundary = agent-scope release of every
// XCD's L2) and read by peers only after they have seen this rank's K1 flag; K1 / K2 readers run a SYSTEM-scope acquire after
// the flags arrive and the writers a system-scope release before they signal.  This was exercised with two processes sharing
// one MI355X (tests/test_distributed_gpu.py: every access is then device-local but still crosses XCDs and processes); on a
// multi-GPU node the same fences are what the HSA memory model requires for peer accesses -- not yet measured there.
#include "common.h"
#include "fbhip.h"

namespace fbhip {

namespace {

constexpr unsigned SPIN_LIMIT = 1u << 28;

// cross-rank barrier at the head of kernel ``k`` (0, 1, 2); returns false on a timeout
__device__ __forceinline__ bool peer_barrier(const PeerComm& pc, int k) {
    __shared__ int ok;
    const int tid = threadIdx.x, W = pc.world;
    const int target = pc.state->epoch[k] + 1;                   // epoch[k] = completed executions of kernel k (bumped by kernel k + 1)
    if (blockIdx.x == 0) {
        if (tid == 0) pc.state->epoch[(k + 2) % 3] += 1;          // the previous kernel of the cycle has completed (stream order)
        if (tid < W) {
            // everything this rank wrote before this launch must be visible to the peer before the flag is
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(pc.flags[tid] + k * PEER_MAX_WORLD + pc.rank, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (tid == 0) ok = 1;
    __syncthreads();
    if (tid < W) {
        const int* slot = pc.flags[pc.rank] + k * PEER_MAX_WORLD + tid;
        unsigned spins = 0;
        while (__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < target) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > SPIN_LIMIT) { ok = 0; pc.state->status = PEER_TIMEOUT; break; }
        }
    }
    __syncthreads();
    if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    __syncthreads();
    return ok != 0;
}

__device__ __forceinline__ void chunk_of(int64_t n4, int W, int r, int64_t& lo, int64_t& hi) {
    const int64_t per = (n4 + W - 1) / W;                         // float4 elements per rank
    lo = per * r < n4 ? per * r : n4;
    hi = lo + per < n4 ? lo + per : n4;
}

__global__ void __launch_bounds__(256) peer_reduce_scatter_kernel(const PeerComm pc, const int which, const int64_t n4) {
    if (!peer_barrier(pc, 0)) return;
    int64_t lo, hi;
    chunk_of(n4, pc.world, pc.rank, lo, hi);
    float4* mine = reinterpret_cast<float4*>(pc.bucket[which][pc.rank]);
    for (int64_t i = lo + (int64_t)blockIdx.x * 256 + threadIdx.x; i < hi; i += (int64_t)gridDim.x * 256) {
        float4 s = reinterpret_cast<const float4*>(pc.bucket[which][0])[i];
        for (int q = 1; q < pc.world; ++q) {
            const float4 v = reinterpret_cast<const float4*>(pc.bucket[which][q])[i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        mine[i] = s;
    }
    // the reduced chunk is read by every peer in K2: write this XCD's L2 back before the launch ends (K2's flag is signalled by
    // ONE workgroup, whose release fence only reaches its own XCD's L2)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
}

__global__ void __launch_bounds__(256) peer_all_gather_kernel(const PeerComm pc, const int which, const int64_t n4) {
    if (!peer_barrier(pc, 1)) return;
    float4* mine = reinterpret_cast<float4*>(pc.bucket[which][pc.rank]);
    for (int q = 0; q < pc.world; ++q) {
        if (q == pc.rank) continue;
        int64_t lo, hi;
        chunk_of(n4, pc.world, q, lo, hi);
        const float4* src = reinterpret_cast<const float4*>(pc.bucket[which][q]);
        for (int64_t i = lo + (int64_t)blockIdx.x * 256 + threadIdx.x; i < hi; i += (int64_t)gridDim.x * 256) mine[i] = src[i];
    }
}

__global__ void __launch_bounds__(64) peer_release_kernel(const PeerComm pc) { (void)peer_barrier(pc, 2); }

// The bucket was written by the backward's GEMM / reduce kernels on all 8 XCDs; their dirty lines may still sit in those XCDs'
// L2s when K1's one signalling workgroup runs.  This launch puts a system-scope release on every XCD (workgroup b runs on XCD
// b % 8; 64 workgroups cover them whatever the dispatcher does) before K1 tells the peers that the bucket can be read.
__global__ void __launch_bounds__(64) peer_flush_kernel() {
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
}

}  // namespace

hipError_t launch_peer_allreduce(const PeerComm& pc, int which, int64_t numel, hipStream_t s) {
    if (pc.world < 2 || pc.world > PEER_MAX_WORLD || (numel & 3)) return hipErrorInvalidValue;
    const int64_t n4 = numel / 4;
    const int blocks = 128;                   // leaves the chip to the pass that runs beside it; a bucket is a few MB per chunk
    hipLaunchKernelGGL(peer_flush_kernel, dim3(64), dim3(64), 0, s);
    hipLaunchKernelGGL(peer_reduce_scatter_kernel, dim3(blocks), dim3(256), 0, s, pc, which, n4);
    hipLaunchKernelGGL(peer_all_gather_kernel, dim3(blocks), dim3(256), 0, s, pc, which, n4);
    hipLaunchKernelGGL(peer_release_kernel, dim3(1), dim3(64), 0, s, pc);
    return hipGetLastError();
}

}  // namespace fbhip
