"""Launch one GEMM shape repeatedly (for rocprofv3 --pmc runs)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from controllable_agent_amd import kernels as K
M, N, Kd, cfg, iters = (int(x) for x in sys.argv[1:6])
A, B, C = torch.randn(M, Kd, device="cuda"), torch.randn(N, Kd, device="cuda"), torch.empty(M, N, device="cuda")
for _ in range(iters):
    K.gemm(A, B, out=C, cfg=cfg)
torch.cuda.synchronize()
