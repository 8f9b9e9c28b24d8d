"""One GEMM shape under every tile configuration (not part of the product): python tools/kbench_shape.py M N K [NT|NN|TN]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from controllable_agent_amd import kernels as K
from tools.kbench import timeit
M, N, Kd = (int(x) for x in sys.argv[1:4])
lay = sys.argv[4] if len(sys.argv) > 4 else "NT"
akc, bkc = {"NT": (True, True), "NN": (True, False), "TN": (False, False)}[lay]
A = torch.randn((M, Kd) if akc else (Kd, M), device="cuda")
B = torch.randn((N, Kd) if bkc else (Kd, N), device="cuda")
C = torch.empty(M, N, device="cuda")
us = timeit(lambda: K.gemm(A, B, a_kcontig=akc, b_kcontig=bkc, out=C))
print(f"{(M, N, Kd)} {lay} auto {us:8.1f} us {2 * M * N * Kd / us / 1e6:7.1f} TF/s")
for cfg in range(6):
    try:
        us = timeit(lambda: K.gemm(A, B, a_kcontig=akc, b_kcontig=bkc, out=C, cfg=cfg))
        print(f"{(M, N, Kd)} {lay} cfg{cfg} {us:8.1f} us {2 * M * N * Kd / us / 1e6:7.1f} TF/s")
    except Exception as e:
        print("cfg", cfg, "failed", str(e)[:80])
