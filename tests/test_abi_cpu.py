"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/fbhip.h
declares, reports the reference's parameter counts, and the product never touches oracle/."""
import ctypes as C
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build_lib()
    from controllable_agent_amd import _lib
    return _lib


def test_header_symbols_all_exported(lib):
    header = (ROOT / "include" / "fbhip.h").read_text()
    declared = set(re.findall(r"\b(fbhip_[a-z0-9_]+)\s*\(", header))
    declared -= {"fbhip_ctx"}
    l = lib.load()
    for name in sorted(declared):
        assert hasattr(l, name), f"{name} declared in include/fbhip.h but not exported by libfbhip.so"
    assert declared == set(lib.PROTOTYPES), "ctypes prototypes out of sync with include/fbhip.h"


def test_parameter_counts_match_reference(lib):
    """SURVEY.md appendix B (confirmed by instantiating the reference): walker, z_dim=50."""
    d = lib.Dims(1024, 24, 6, 24, 50, 1024, 512, 526, 0, 0, 1, 1)
    l = lib.load()
    assert l.fbhip_net_param_count(C.byref(d), lib.NET_FORWARD) == 3_363_940
    assert l.fbhip_net_param_count(C.byref(d), lib.NET_BACKWARD) == 317_754
    assert l.fbhip_net_param_count(C.byref(d), lib.NET_ACTOR) == 2_211_846
    assert [l.fbhip_layout_count(C.byref(d), n) for n in range(3)] == [20, 8, 16]


def test_layout_is_aligned_and_disjoint(lib):
    l = lib.load()
    for dims in (lib.Dims(16, 5, 3, 5, 8, 32, 16, 18, 0, 0, 1, 1), lib.Dims(1024, 78, 12, 2, 100, 1024, 512, 526, 1, 0, 1, 1)):
        for net in range(3):
            spans = []
            for i in range(l.fbhip_layout_count(C.byref(dims), net)):
                t = lib.TensorDesc()
                assert l.fbhip_layout_entry(C.byref(dims), net, i, C.byref(t)) == 0
                assert t.offset % 4 == 0 and t.ld % 4 == 0 and t.ld >= t.cols
                spans.append((t.offset, t.offset + t.rows * t.ld))
            spans.sort()
            for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
                assert a1 <= b0
            assert spans[-1][1] <= l.fbhip_net_numel(C.byref(dims), net)


def test_single_trunk_layout_follows_the_reference_module(lib):
    """preprocess == 0 (fb_modules.py:99-103, 174-178): one ``trunk`` mlp on the concatenated input, Linear layers at
    Sequential indices 0, 3, 5 and the LayerNorm at 1 -- names and order of nn.Module.state_dict()."""
    l = lib.load()
    d = lib.Dims(16, 5, 3, 5, 8, 32, 16, 18, 0, 0, 0, 1)
    def names(net):
        out = []
        for i in range(l.fbhip_layout_count(C.byref(d), net)):
            t = lib.TensorDesc()
            assert l.fbhip_layout_entry(C.byref(d), net, i, C.byref(t)) == 0
            out.append((t.name.decode(), t.rows, t.cols))
        return out
    trunk = lambda k: [("trunk.0.weight", 32, k), ("trunk.0.bias", 1, 32), ("trunk.1.weight", 1, 32), ("trunk.1.bias", 1, 32),
                       ("trunk.3.weight", 32, 32), ("trunk.3.bias", 1, 32), ("trunk.5.weight", 32, 32), ("trunk.5.bias", 1, 32)]
    assert names(lib.NET_ACTOR) == trunk(13) + [("policy.0.weight", 32, 32), ("policy.0.bias", 1, 32),
                                                ("policy.2.weight", 3, 32), ("policy.2.bias", 1, 3)]
    assert names(lib.NET_FORWARD)[:8] == trunk(16)
    assert [n for n, _, _ in names(lib.NET_FORWARD)[8:]] == [f"{h}.{i}.{w}" for h in ("F1", "F2") for i in (0, 2)
                                                             for w in ("weight", "bias")]


def test_boltzmann_actor_layout_follows_the_reference_module(lib):
    """boltzmann: DiagGaussianActor.policy = mlp(o + z, H, "ntanh", H, "relu", 2a) (fb_modules.py:138)."""
    l = lib.load()
    d = lib.Dims(16, 5, 3, 5, 8, 32, 16, 18, 0, 0, 1, 1, 1)
    got = []
    for i in range(l.fbhip_layout_count(C.byref(d), lib.NET_ACTOR)):
        t = lib.TensorDesc()
        assert l.fbhip_layout_entry(C.byref(d), lib.NET_ACTOR, i, C.byref(t)) == 0
        got.append((t.name.decode(), t.rows, t.cols))
    assert got == [("policy.0.weight", 32, 13), ("policy.0.bias", 1, 32), ("policy.1.weight", 1, 32), ("policy.1.bias", 1, 32),
                   ("policy.3.weight", 32, 32), ("policy.3.bias", 1, 32), ("policy.5.weight", 6, 32), ("policy.5.bias", 1, 6)]
    assert l.fbhip_net_param_count(C.byref(d), lib.NET_ACTOR) == 32 * 13 + 32 * 3 + 32 * 32 + 32 + 6 * 32 + 6
    ctx = C.c_void_p()
    d0 = lib.Dims(16, 5, 3, 5, 8, 32, 16, 18, 0, 0, 1, 1, 0)
    assert l.fbhip_create(C.byref(d0), C.byref(ctx)) == 0
    assert l.fbhip_set_policy_squash(ctx, 1.0, -5.0, 2.0) == -3          # not a boltzmann context
    assert l.fbhip_destroy(ctx) == 0


def test_discrete_layout_follows_the_reference_module(lib):
    """dims.discrete (discrete_fb.ForwardMap, discrete_fb.py:74-83): trunk on cat([obs, z]) -- no action columns --, heads
    z_dim * A wide; no actor; preprocess must be 0 like in the reference (its preprocess branch cannot run)."""
    l = lib.load()
    d = lib.Dims(16, 5, 4, 5, 8, 32, 16, 18, 0, 0, 0, 1, 0, 1)
    got = []
    for i in range(l.fbhip_layout_count(C.byref(d), lib.NET_FORWARD)):
        t = lib.TensorDesc()
        assert l.fbhip_layout_entry(C.byref(d), lib.NET_FORWARD, i, C.byref(t)) == 0
        got.append((t.name.decode(), t.rows, t.cols))
    assert got == [("trunk.0.weight", 32, 13), ("trunk.0.bias", 1, 32), ("trunk.1.weight", 1, 32), ("trunk.1.bias", 1, 32),
                   ("trunk.3.weight", 32, 32), ("trunk.3.bias", 1, 32), ("trunk.5.weight", 32, 32), ("trunk.5.bias", 1, 32),
                   ("F1.0.weight", 32, 32), ("F1.0.bias", 1, 32), ("F1.2.weight", 32, 32), ("F1.2.bias", 1, 32),
                   ("F2.0.weight", 32, 32), ("F2.0.bias", 1, 32), ("F2.2.weight", 32, 32), ("F2.2.bias", 1, 32)]
    assert l.fbhip_layout_count(C.byref(d), lib.NET_ACTOR) == 0 and l.fbhip_net_numel(C.byref(d), lib.NET_ACTOR) == 0
    bad = lib.Dims(16, 5, 4, 5, 8, 32, 16, 18, 0, 0, 1, 1, 0, 1)
    assert l.fbhip_net_numel(C.byref(bad), 0) < 0 and b"preprocess" in l.fbhip_last_error(None)


def test_bad_dims_fail_loudly(lib):
    l = lib.load()
    bad = lib.Dims(1024, 24, 6, 24, 50, 1022, 512, 526, 0, 0, 1, 1)       # hidden_dim % 4 != 0
    assert l.fbhip_net_numel(C.byref(bad), 0) < 0
    assert b"multiples of 4" in l.fbhip_last_error(None)
    ctx = C.c_void_p()
    assert l.fbhip_create(C.byref(bad), C.byref(ctx)) != 0


def test_unbound_context_is_an_error_not_a_crash(lib):
    l = lib.load()
    d = lib.Dims(16, 5, 3, 5, 8, 32, 16, 18, 0, 0, 1, 1)
    ctx = C.c_void_p()
    assert l.fbhip_create(C.byref(d), C.byref(ctx)) == 0
    hp = lib.HParams()
    assert l.fbhip_update(ctx, C.byref(hp), None, lib.PHASE_ALL, 0, None) == -3      # FBHIP_E_STATE
    assert b"not bound" in l.fbhip_last_error(ctx)
    assert l.fbhip_destroy(ctx) == 0


def test_struct_size_guard(lib):
    """a struct built against another header (wrong struct_size) is refused by every entry point that takes it"""
    l = lib.load()
    d = lib.Dims(16, 5, 3, 5, 8, 32, 16, 18, 0, 0, 1, 1)
    assert d.struct_size == C.sizeof(lib.Dims) == 17 * 4
    d.struct_size -= 4                                   # what a binding written against an older header (a field short) passes
    assert l.fbhip_net_numel(C.byref(d), 0) < 0 and b"struct_size" in l.fbhip_last_error(None)
    ctx = C.c_void_p()
    assert l.fbhip_create(C.byref(d), C.byref(ctx)) == -1
    d.struct_size += 4
    assert l.fbhip_create(C.byref(d), C.byref(ctx)) == 0
    hp = lib.HParams()
    assert hp.struct_size == C.sizeof(lib.HParams)
    hp.struct_size = 0
    assert l.fbhip_update(ctx, C.byref(hp), None, lib.PHASE_ALL, 0, None) in (-1, -3)
    assert l.fbhip_destroy(ctx) == 0


def test_plain_c_program_links_and_runs_against_the_header(lib, tmp_path):
    """a real non-Python consumer: gcc compiles tests/c_consumer/abi_consumer.c against include/fbhip.h, links libfbhip.so,
    and the program checks version / layout / struct_size guard / error path on its own"""
    import os, shutil, subprocess
    lib.load()
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = tmp_path / "abi_consumer"
    so_dir = lib.LIB_PATH.parent
    torch_lib = Path(__import__("torch").__file__).parent / "lib"
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-I", str(ROOT / "include"), str(ROOT / "tests" / "c_consumer" / "abi_consumer.c"),
           "-o", str(exe), "-L", str(so_dir), "-l:libfbhip.so", f"-Wl,-rpath,{so_dir}", f"-Wl,-rpath,{torch_lib}",
           f"-Wl,-rpath-link,{torch_lib}", "-Wl,--allow-shlib-undefined"]
    subprocess.run(cmd, check=True)
    env = dict(os.environ, LD_LIBRARY_PATH=f"{torch_lib}:/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout + out.stderr


def test_no_gpu_means_loud_failure(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lib.require_device()


def test_product_never_imports_oracle():
    for f in list((ROOT / "controllable_agent_amd").rglob("*.py")) + list((ROOT / "controllable_agent_amd").rglob("*.hip")):
        txt = f.read_text()
        assert "oracle" not in txt.lower().replace("# oracle", ""), f"{f} mentions oracle/"


def test_integration_doc_structs_match_the_header(lib):
    """INTEGRATION.md section 4 shows the ctypes structs a maintainer would write: their field lists must be the header's
    (round 1 shipped the doc one field short of ABI 12)."""
    header = (ROOT / "include" / "fbhip.h").read_text()
    doc = (ROOT / "INTEGRATION.md").read_text()

    def header_fields(struct):
        body = header[header.index(f"typedef struct {struct} {{"):header.index(f"}} {struct};")]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        return re.findall(r"\b(?:u?int32_t|float)\s+([a-z_0-9]+)\s*;", body)

    def doc_fields(cls):
        body = doc[doc.index(f"class {cls}(C.Structure)"):]
        body = body[:body.index("\nclass ") if "\nclass " in body[1:200 + len(body)] and body.index("\nclass ") < body.index("assert lib") else body.index("assert lib")]
        return re.findall(r'"([a-z_0-9]+)"', body)

    for struct, cls, ct in (("fbhip_dims", "Dims", lib.Dims), ("fbhip_hparams", "HParams", lib.HParams)):
        h = header_fields(struct)
        assert h == [n for n, _ in ct._fields_], f"_lib.{cls} out of sync with include/fbhip.h"
        assert doc_fields(cls) == h, f"INTEGRATION.md {cls} out of sync with include/fbhip.h"


def test_sf_layout_follows_the_reference_modules(lib):
    """dims.sf (sf.py:84-88, 194-200): NET_BACKWARD is feature_learner -- feature_net.{0,1,3,5} and, for icm, the
    inverse_dynamic_net mlp(2 z, Hb, relu, Hb, relu, a, tanh) at Sequential indices 0, 2, 4 -- in state_dict() order"""
    l = lib.load()
    def names(sf):
        d = lib.Dims(16, 5, 3, 5, 10, 32, 16, 20, 0, 0, 1, 1, 0, 0, sf)
        out = []
        for i in range(l.fbhip_layout_count(C.byref(d), lib.NET_BACKWARD)):
            t = lib.TensorDesc()
            assert l.fbhip_layout_entry(C.byref(d), lib.NET_BACKWARD, i, C.byref(t)) == 0
            out.append((t.name.decode(), t.rows, t.cols))
        return out
    feat = [("feature_net.0.weight", 20, 5), ("feature_net.0.bias", 1, 20), ("feature_net.1.weight", 1, 20), ("feature_net.1.bias", 1, 20),
            ("feature_net.3.weight", 20, 20), ("feature_net.3.bias", 1, 20), ("feature_net.5.weight", 10, 20), ("feature_net.5.bias", 1, 10)]
    assert names(2) == feat
    assert names(1) == feat + [("inverse_dynamic_net.0.weight", 20, 20), ("inverse_dynamic_net.0.bias", 1, 20),
                               ("inverse_dynamic_net.2.weight", 20, 20), ("inverse_dynamic_net.2.bias", 1, 20),
                               ("inverse_dynamic_net.4.weight", 3, 20), ("inverse_dynamic_net.4.bias", 1, 3)]
    head = lambda n, fin, fout: [(f"{n}.0.weight", 20, fin), (f"{n}.0.bias", 1, 20), (f"{n}.2.weight", 20, 20), (f"{n}.2.bias", 1, 20),
                                 (f"{n}.4.weight", fout, 20), (f"{n}.4.bias", 1, fout)]
    assert names(3) == feat                                                    # random: feature_net only (sf.py:84-92)
    assert names(4) == feat + head("decoder", 10, 5)                           # autoencoder: mlp(z, Hb, Hb, goal_dim), sf.py:253
    assert names(5) == feat + head("forward_dynamic_net", 13, 5)               # transition: mlp(z + a, Hb, Hb, goal_dim), sf.py:219
    assert names(0)[0][0] == "B.0.weight"
    assert names(6) == feat + [("mu_net.0.weight", 20, 8), ("mu_net.0.bias", 1, 20), ("mu_net.1.weight", 1, 20), ("mu_net.1.bias", 1, 20),
                               ("mu_net.3.weight", 20, 20), ("mu_net.3.bias", 1, 20), ("mu_net.5.weight", 10, 20), ("mu_net.5.bias", 1, 10)]   # svd_p, sf.py:340
    assert names(7) == feat + head("forward_dynamic_net", 13, 10)              # latent: mlp(z + a, Hb, Hb, z), sf.py:233 (its target net: the target buffer)
    assert names(8) == feat + [("mu_net.0.weight", 20, 5), ("mu_net.0.bias", 1, 20), ("mu_net.1.weight", 1, 20), ("mu_net.1.bias", 1, 20),
                               ("mu_net.3.weight", 20, 20), ("mu_net.3.bias", 1, 20), ("mu_net.5.weight", 10, 20), ("mu_net.5.bias", 1, 10)]   # svd_sr, sf.py:265
    assert names(9) == names(8)                                                # svd_srv2: the same modules (sf.py:304-308)
    assert names(10) == names(8)                                               # contrastive: mu_net on the (hindsight) goal (sf.py:121)
    assert names(11) == names(8)                                               # contrastivev2 (sf.py:162)
    ident = lib.Dims(16, 5, 3, 5, 5, 32, 16, 20, 0, 0, 1, 1, 0, 0, 12)         # identity: needs z_dim == goal_dim (sf.py:94-98)
    assert l.fbhip_net_numel(C.byref(ident), 0) > 0
    assert l.fbhip_net_numel(C.byref(lib.Dims(16, 5, 3, 5, 10, 32, 16, 20, 0, 0, 1, 1, 0, 0, 12)), 0) < 0 and b"z_dim == goal_dim" in l.fbhip_last_error(None)
    too = lib.Dims(16, 5, 3, 5, 10, 32, 16, 20, 0, 0, 1, 1, 0, 0, 13)
    assert l.fbhip_net_numel(C.byref(too), 0) < 0 and b"dims.sf" in l.fbhip_last_error(None)
    bad = lib.Dims(16, 5, 3, 5, 10, 32, 16, 20, 0, 0, 1, 1, 1, 0, 1)          # boltzmann + sf
    assert l.fbhip_net_numel(C.byref(bad), 0) < 0 and b"dims.sf" in l.fbhip_last_error(None)


def test_host_side_under_address_sanitizer(lib, tmp_path):
    """an ASan build of the library's HOST side (``make asan``: layout, schedule and C-ABI translation units;
    device code as usual) driven by the plain-C consumer -- layout queries over every configuration flag, context create /
    destroy, the struct_size guard and the unbound error path must be clean under AddressSanitizer."""
    import glob, os, shutil, subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    rts = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    if not rts:
        pytest.skip("no clang ASan runtime in this image")
    csrc = lib.LIB_PATH.parent / "csrc"
    subprocess.run(["make", "-C", str(csrc), "asan"], check=True, stdout=subprocess.DEVNULL)
    so_dir = lib.LIB_PATH.parent
    torch_lib = Path(__import__("torch").__file__).parent / "lib"
    exe = tmp_path / "abi_consumer_asan"
    subprocess.run(["gcc", "-std=c99", "-g", "-I", str(ROOT / "include"), str(ROOT / "tests" / "c_consumer" / "abi_consumer.c"), "-o", str(exe),
                    "-L", str(so_dir), "-l:libfbhip_asan.so", f"-Wl,-rpath,{so_dir}", f"-Wl,-rpath,{torch_lib}",
                    f"-Wl,-rpath-link,{torch_lib}", "-Wl,--allow-shlib-undefined"], check=True)
    env = dict(os.environ, LD_PRELOAD=rts[0], ASAN_OPTIONS="detect_leaks=0:abort_on_error=0",
               LD_LIBRARY_PATH=f"{torch_lib}:/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK") and "AddressSanitizer" not in out.stderr, out.stdout + out.stderr
