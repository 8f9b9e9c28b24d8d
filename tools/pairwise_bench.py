"""pairwise_kernel + pairwise_reduce_kernel alone on the GPU box (buffers preallocated, events on the launch stream); run it under
`rocprofv3 --kernel-trace --stats` for the split between the two kernels.  Not part of the product.
Algorithmic work of the loss and its three gradients = 11 products of 2 B^2 d flops (5 tiles + 6 contractions)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from controllable_agent_amd import _lib
from controllable_agent_amd.kernels import check, ptr, stream_ptr


def main():
    lib = _lib.load()
    torch.manual_seed(0)
    shapes = [(1024, 50), (2048, 100), (2048, 50), (1024, 100), (4096, 100), (512, 32)]
    dbg = "--dbg" in sys.argv          # instrumented builds (not in the tree) leave cycle stamps behind the scalar partials
    args = [a for a in sys.argv[1:] if a != "--dbg"]
    if args:
        shapes = [tuple(int(x) for x in a.split("x")) for a in args]
    for Bn, d in shapes:
        ld = (d + 3) // 4 * 4
        ins = [torch.randn(Bn, ld, device="cuda") for _ in range(6)]
        outs = [torch.empty(Bn, ld, device="cuda") for _ in range(3)]
        disc = torch.full((Bn,), 0.98, device="cuda")
        metrics = torch.zeros(_lib.NUM_METRICS, device="cuda")
        nscr = lib.fbhip_pairwise_scratch_floats(Bn, d)
        scratch = torch.zeros(nscr + 1024, device="cuda")

        def run():
            check(lib.fbhip_pairwise_fb(*(ptr(t) for t in ins), ptr(disc), Bn, d, ld, 1.0, *(ptr(t) for t in outs), ptr(metrics),
                                        ptr(scratch), stream_ptr()))
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 50
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        if dbg:
            tail = scratch.cpu()
            nz = tail.nonzero().flatten()
            last = int(nz[-1]) if len(nz) else 0
            blk = tail[max(0, last - 200):last + 1]
            print("stamps:", [round(float(x)) for x in blk if x != 0][-80:])
        fl = 11 * 2 * Bn * Bn * d
        print(f"pairwise B={Bn} d={d}: {us:8.1f} us per call (kernel + reduce, back to back)  {fl / us / 1e6:7.2f} TFLOP/s "
              f"= {fl / us / 1e6 / 157.3:.3f} of the fp32 MFMA peak")


if __name__ == "__main__":
    main()
