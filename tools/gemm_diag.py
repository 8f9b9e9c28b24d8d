"""Bottleneck diagnosis of gemm_kernel (not part of the product): builds variants of csrc/gemm.hip with one pipeline
stage knocked out (results are WRONG by construction) and times the dominant shape with each.

  python tools/gemm_diag.py build      # here (hipcc cross-compiles): writes build/diag/libfbhip_<variant>.so
  python tools/gemm_diag.py run        # on the GPU box: swaps each variant in and times it
"""
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "controllable_agent_amd" / "csrc"
OUT = ROOT / "controllable_agent_amd" / "csrc" / "build" / "diag"
MFMA = "acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc, 0, 0, 0);"
LOADA = "for (int i = 0; i < QA; ++i) xa[i] = *reinterpret_cast<const float4*>(A + ga[i] + oa);"
LOADB = "for (int i = 0; i < QB; ++i) xb[i] = *reinterpret_cast<const float4*>(Bp + gb[i] + ob);"
STA = "for (int i = 0; i < QA; ++i) store_quad(dA + sa[i], step_a, xa[i]);"
STB = "for (int i = 0; i < QB; ++i) store_quad(dA + BKT * LDA_S + sb[i], step_b, xb[i]);"
PBAR = "            if constexpr (decltype(do_store)::value) store_chunk((it + 1) & 1, sa_, sb_);\n            __syncthreads();"
CBAR = "        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);              // last two MFMA pairs\n        __syncthreads();"


def variants(src):
    """Knock-outs of the LDS-DMA kernel (gemm_dma_kernel): results are wrong, timings tell what the stage costs."""
    v = {"base": src}
    v["dma_halfload"] = src.replace("            glds16(sa[1] + (size_t)c * adv_a, d + 256);\n", "") \
                           .replace("            glds16(sb[1] + (size_t)c * adv_b, d + 2048 + 256);\n", "")
    v["dma_noload"] = src.replace("            glds16(sa[0] + (size_t)c * adv_a, d);\n", "") \
                         .replace("            glds16(sa[1] + (size_t)c * adv_a, d + 256);\n", "") \
                         .replace("            glds16(sb[0] + (size_t)c * adv_b, d + 2048);\n", "") \
                         .replace("            glds16(sb[1] + (size_t)c * adv_b, d + 2048 + 256);\n", "")
    v["dma_sameaddr"] = src.replace("glds16(sa[0] + (size_t)c * adv_a, d);", "glds16(sa[0], d);") \
                           .replace("glds16(sa[1] + (size_t)c * adv_a, d + 256);", "glds16(sa[1], d + 256);") \
                           .replace("glds16(sb[0] + (size_t)c * adv_b, d + 2048);", "glds16(sb[0], d + 2048);") \
                           .replace("glds16(sb[1] + (size_t)c * adv_b, d + 2048 + 256);", "glds16(sb[1], d + 2048 + 256);")
    MF = "            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][m], b[t][m], acc, 0, 0, 0);\n            csum += a[t][m];"
    assert MF in src
    v["dma_nomfma"] = src.replace(MF, "            acc[(4 * t + m) % 16] += a[t][m] * b[t][m];")
    for k, s in v.items():
        assert k == "base" or s != src, k
    return v


def build():
    OUT.mkdir(parents=True, exist_ok=True)
    src = (CSRC / "gemm.hip").read_text()
    objs = [str(CSRC / "build" / f"{n}.o") for n in ("rowops", "pairwise", "optim", "sampler", "api")]
    for name, text in variants(src).items():
        d = OUT / name
        d.mkdir(exist_ok=True)
        (d / "gemm.hip").write_text(text)
        shutil.copy(CSRC / "common.h", d / "common.h")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT / 'include'}",
               "-Wno-unused-function", "-c", str(d / "gemm.hip"), "-o", str(d / "gemm.o")]
        subprocess.check_call(cmd)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               str(OUT / f"libfbhip_{name}.so"), str(d / "gemm.o")] + objs)
        print("built", name)


RUNNER = r'''
import sys, torch
sys.path.insert(0, %r)
from controllable_agent_amd import kernels as K
def timeit(fn, iters=50, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (M, N, Kd) in ((1024, 2048, 1024), (4096, 4096, 4096), (1024, 1024, 1024)):
    A, B, C = torch.randn(M, Kd, device="cuda"), torch.randn(N, Kd, device="cuda"), torch.empty(M, N, device="cuda")
    us = timeit(lambda: K.gemm(A, B, out=C, cfg=5), iters=20 if M > 2048 else 50)
    print(f"  {str((M, N, Kd)):>22} cfg5 {us:9.1f} us {2 * M * N * Kd / us / 1e6:8.2f} TF-equivalent")
'''


def run():
    lib = ROOT / "controllable_agent_amd" / "libfbhip.so"
    keep = lib.read_bytes()
    try:
        for so in sorted(OUT.glob("libfbhip_*.so")):
            lib.write_bytes(so.read_bytes())
            print(so.stem.replace("libfbhip_", ""), flush=True)
            subprocess.call([sys.executable, "-c", RUNNER % str(ROOT)])
    finally:
        lib.write_bytes(keep)


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
