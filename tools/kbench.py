"""Per-kernel timing on the GPU box (torch.cuda.Event on the launch stream).  Not part of the product."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from controllable_agent_amd import kernels as K


def timeit(fn, iters=50, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3     # us


def main():
    torch.manual_seed(0)
    print(f"{'shape':>28} {'layout':>6} {'cfg':>4} {'us':>9} {'TF/s':>8}")
    shapes = [(1024, 1024, 1024), (1024, 2048, 1024), (1024, 512, 1024), (1024, 1024, 2048), (1024, 1024, 512),
              (1024, 50, 1024), (1024, 1024, 74), (2048, 1024, 1024), (1024, 526, 526), (50, 1024, 1024), (4096, 4096, 4096)]
    for (M, N, Kd) in shapes:
        for lay, (akc, bkc) in (("NT", (True, True)), ("NN", (True, False)), ("TN", (False, False))):
            A = torch.randn((M, Kd) if akc else (Kd, M), device="cuda")
            B = torch.randn((N, Kd) if bkc else (Kd, N), device="cuda")
            C = torch.empty(M, N, device="cuda")
            us = timeit(lambda: K.gemm(A, B, a_kcontig=akc, b_kcontig=bkc, out=C))
            print(f"{str((M, N, Kd)):>28} {lay:>6} {'auto':>4} {us:9.1f} {2 * M * N * Kd / us / 1e6:8.2f}")
        if M * N >= 1024 * 512:
            A, B, C = torch.randn(M, Kd, device="cuda"), torch.randn(N, Kd, device="cuda"), torch.empty(M, N, device="cuda")
            for cfg in range(7):
                if cfg >= 5 and Kd % 32:
                    continue
                us = timeit(lambda: K.gemm(A, B, out=C, cfg=cfg))
                print(f"{str((M, N, Kd)):>28} {'NT':>6} {cfg:>4} {us:9.1f} {2 * M * N * Kd / us / 1e6:8.2f}")
    ref = torch.randn(1024, 1024, device="cuda")
    us = timeit(lambda: torch.mm(ref, ref))
    print(f"torch.mm 1024^3 (rocBLAS/hipBLASLt fp32): {us:.1f} us {2 * 1024 ** 3 / us / 1e6:.2f} TF/s")
    ref = torch.randn(4096, 4096, device="cuda")
    us = timeit(lambda: torch.mm(ref, ref), iters=10)
    print(f"torch.mm 4096^3: {us:.1f} us {2 * 4096 ** 3 / us / 1e6:.2f} TF/s")
    for (Bn, d) in ((1024, 50), (2048, 100), (256, 50)):
        args = [torch.randn(Bn, (d + 3) // 4 * 4, device="cuda")[:, :d] for _ in range(6)]
        disc = torch.full((Bn,), 0.99, device="cuda")
        us = timeit(lambda: K.pairwise_fb(*args, disc, 1.0), iters=20)
        print(f"pairwise_fb B={Bn} d={d}: {us:.1f} us (incl. host alloc) ~{11 * 2 * Bn * Bn * d / us / 1e6:.2f} TF/s")
    x, g, b = torch.randn(1024, 1024, device="cuda"), torch.ones(1024, device="cuda"), torch.zeros(1024, device="cuda")
    print(f"ln_tanh_fwd 1024x1024: {timeit(lambda: K.ln_tanh_fwd(x, g, b)):.1f} us")
    y, st = K.ln_tanh_fwd(x, g, b)
    print(f"ln_tanh_bwd 1024x1024: {timeit(lambda: K.ln_tanh_bwd(x, y, x, st, g)):.1f} us")
    n = 3_686_956 // 4 * 4
    p, gr, m, v, t = (torch.randn(n, device="cuda") for _ in range(5))
    v.abs_()
    us = timeit(lambda: K.adam_ema(p, gr, m, v, t, 1e-4, 3, tau=0.01))
    print(f"adam_ema {n} params: {us:.1f} us  {n * 4 * 9 / us / 1e3:.1f} GB/s")


if __name__ == "__main__":
    main()
