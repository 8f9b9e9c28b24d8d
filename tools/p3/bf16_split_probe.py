#!/usr/bin/env python
"""How accurate would an fp32 GEMM emulated with bf16 MFMAs be?  (the fp32 MFMA peak is 157 TFLOP/s,
the bf16 one 2.5 PFLOP/s, so n bf16 products per fp32 product have an effective peak of 2500 / n.)

a = a0 + a1 + a2 with bf16 pieces (8 + 8 + 8 mantissa bits); products accumulate in fp32 like v_mfma_f32_32x32x16_bf16.
CPU emulation: bf16-rounded pieces held in fp32, fp32 matmul (each product of two bf16 values is exact in fp32).
Prints the relative L2 error against an fp64 product for the step's dominant shape, beside plain fp32."""
import torch

torch.manual_seed(0)
M, N, K = 1024, 2048, 1024


def split(x, n):
    out, r = [], x.clone()
    for _ in range(n):
        p = r.to(torch.bfloat16).to(torch.float32)
        out.append(p)
        r = r - p
    return out


A, B = torch.randn(M, K), torch.randn(K, N) / K ** 0.5
ref = A.double() @ B.double()
err = lambda C: float((C.double() - ref).norm() / ref.norm())
print(f"fp32 matmul                      : {err(A @ B):.2e}")
for na, terms in ((2, [(0, 0), (0, 1), (1, 0)]), (2, [(0, 0), (0, 1), (1, 0), (1, 1)]),
                  (3, [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)]),
                  (3, [(i, j) for i in range(3) for j in range(3)])):
    a, b = split(A, na), split(B, na)
    C = torch.zeros(M, N)
    for i, j in sorted(terms, key=lambda t: -(t[0] + t[1])):     # small terms first
        C += a[i] @ b[j]
    print(f"bf16 x{len(terms)} ({na} pieces per operand)   : {err(C):.2e}   effective peak {2500 / len(terms):.0f} TFLOP/s")
