"""GPU: head_kernel (csrc/fused.hip) against the launches it replaces, each as a hipGraph of 20 launches
(host-issue time excluded).  Not part of the product."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from controllable_agent_amd import _lib, kernels as K
from controllable_agent_amd._lib import ptr


def graph_time(fn, reps=20, iters=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters / reps * 1e3


def main():
    lib = _lib.load()
    dev = "cuda"
    torch.manual_seed(0)
    rows, H = 1024, 1024
    for (N, Kd) in ((50, 1024), (50, 576), (64, 1024)):
        x, w = torch.randn(rows, Kd, device=dev), torch.randn(N, Kd, device=dev)
        Np = (N + 3) // 4 * 4
        b = torch.randn(Np, device=dev)
        c, o2, nr = torch.zeros(rows, Np, device=dev), torch.zeros(rows, Np, device=dev), torch.zeros(rows, device=dev)
        sp = lambda: torch.cuda.current_stream().cuda_stream
        def base():
            lib.fbhip_gemm(ptr(x), Kd, 1, ptr(w), Kd, 1, ptr(c), Np, rows, N, Kd, ptr(b), None, 0, _lib.EPI_BIAS, None, sp())
            lib.fbhip_l2norm_fwd(ptr(c), Np, ptr(o2), Np, ptr(nr), rows, N, sp())
        print(f"head {rows}x{N}x{Kd}: gemm + l2norm_fwd (one problem alone, no split-K slab outside a context) {graph_time(base):7.2f} us")
        for rep in (1, 2, 3, 4):
            for norm in (0, 1):
                f = lambda: lib.fbhip_head(ptr(x), Kd, ptr(w), Kd, ptr(b), ptr(c), Np, ptr(o2) if norm else None, Np,
                                           ptr(nr) if norm else None, 7.0, rows, N, Kd, rep, sp())
                print(f"   head_kernel x{rep} problems per launch, normalize {norm}: {graph_time(f):7.2f} us")


if __name__ == "__main__":
    main()
