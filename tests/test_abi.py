"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports
every symbol include/peanut_hip.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

from peanut_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = set()
    inc = os.path.join(ROOT, "include")
    for fn in os.listdir(inc):
        if not fn.endswith(".h"):
            continue
        src = open(os.path.join(inc, fn)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        for m in re.finditer(r"\b(peanut_[a-z0-9_]+)\s*\(", src):
            names.add(m.group(1))
    return names


def test_library_builds_and_loads():
    path = build.build()
    assert os.path.exists(path)
    lib = _lib.load()
    assert lib.peanut_abi_version() >= 1
    assert lib.peanut_build_arch() == b"gfx950"


def test_every_declared_symbol_is_exported_and_bound():
    build.build()
    lib = ctypes.CDLL(build.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 10
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    # the python binding table covers exactly the declared ABI
    assert set(_lib.SIGNATURES) == declared


def test_code_object_is_gfx950_only():
    """Every device code object bundled in the library targets gfx950 and nothing else (rocPRIM's
    host-side arch-name table also mentions other gfx names as plain strings; those are not code)."""
    build.build()
    blob = open(build.LIB_PATH, "rb").read()
    targets = set(re.findall(rb"hipv4-amdgcn-amd-amdhsa--([a-z0-9]+)", blob))
    assert targets == {b"gfx950"}, targets
    assert b"sm_90" not in blob and b"nvptx" not in blob


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setenv("PEANUT_HIP_LIB", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "_LIB", None)
    with pytest.raises(_lib.PeanutHipError):
        _lib.load()


def test_product_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from peanut_amd.prediction import PEANUT_Prediction_Model
    from peanut_amd.weights import PredCfg, make_seeded_state_dict
    from types import SimpleNamespace
    with pytest.raises(_lib.PeanutHipError):
        PEANUT_Prediction_Model(SimpleNamespace(sem_gpu_id=0), state_dict=make_seeded_state_dict(PredCfg(), 0))


def test_product_package_never_touches_the_oracle_or_the_reference():
    """The oracle is test infrastructure: nothing under peanut_amd/ (nor bench.py outside its cpu_baseline leg, nor
    __graft_entry__ outside smoke()) may import it, and nothing that runs on the GPU box may read /root/reference."""
    import ast
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def oracle_imports(path):
        tree = ast.parse(open(path).read())
        hits = []
        for node in ast.walk(tree):
            if isinstance(node, ast.Import):
                hits += [(node.lineno, a.name) for a in node.names if a.name.split(".")[0] == "oracle"]
            elif isinstance(node, ast.ImportFrom) and (node.module or "").split(".")[0] == "oracle":
                hits.append((node.lineno, node.module))
        return hits

    for path in glob.glob(os.path.join(root, "peanut_amd", "**", "*.py"), recursive=True):
        assert not oracle_imports(path), f"{path} imports the oracle"
        assert "/root/reference" not in open(path).read(), f"{path} mentions /root/reference"
    # bench.py: exactly one import, inside cpu_baseline(); __graft_entry__: inside smoke() (the checker) and build()
    # (which only COMPILES the oracle's C restatement, oracle/fmm_ref.c -- building the checker is not using it)
    for fname, funcs in (("bench.py", ("cpu_baseline",)), ("__graft_entry__.py", ("smoke", "build"))):
        path = os.path.join(root, fname)
        func = " / ".join(funcs)
        tree = ast.parse(open(path).read())
        inside = set()
        for node in ast.walk(tree):
            if isinstance(node, ast.FunctionDef) and node.name in funcs:
                inside |= {n.lineno for n in ast.walk(node) if isinstance(n, (ast.Import, ast.ImportFrom))}
        hits = oracle_imports(path)
        assert hits and all(line in inside for line, _ in hits), f"{fname}: oracle imported outside {func}()"
        assert "/root/reference" not in open(path).read()


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus 2` on a box with fewer devices exits non-zero and prints no result line (here: none)."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300, text=True)
    import torch
    if torch.cuda.device_count() < 2:
        assert r.returncode != 0 and "n_gpus" not in r.stdout


def test_library_carries_the_hash_of_its_sources(monkeypatch):
    """The library answers for the csrc/ + include/ sources next to it: it embeds their content hash, and the binding
    refuses a library whose hash differs (file times do not survive the copy to the GPU box, contents do)."""
    build.build()
    lib = _lib.load()
    assert lib.peanut_source_hash().decode() == build.source_hash()
    assert not build.is_stale()
    monkeypatch.setattr(build, "source_hash", lambda: "0" * 16)          # "the sources changed"
    assert _lib._stale_reason(lib, build.LIB_PATH)
    assert _lib._stale_reason(lib, "/elsewhere/libpeanut_hip.so") == ""   # a library given by path is taken as it is


@pytest.mark.parametrize("precision", ["bf16x6", "bf16x3", "fp16x3"])
def test_weight_pieces_of_the_emulated_modes_match_independent_roundings(precision):
    """Host arithmetic of the weight packer (csrc/rs_common.h), no GPU: the pieces of a layer's weights against torch's
    bfloat16 / numpy's float16 round-to-nearest-even, piece by piece -- including the power-of-two pack scale of the fp16
    mode, fp16 subnormals and the edge of its range."""
    import ctypes as C
    import numpy as np
    import torch
    lib = _lib.load()
    rng = np.random.RandomState(5)
    v = (rng.standard_normal(4096) * np.exp(rng.uniform(-12, 3, 4096))).astype(np.float32)
    v[:8] = [0.0, -0.0, 1.0, -1.0, 3.0e-7, 65504.0 / 2 ** 14, 1.17549435e-38, 0.333333343]
    pieces = np.zeros((3, v.size), np.uint16)
    scale = C.c_float(0)
    _lib.check(lib.peanut_debug_weight_pieces(v.ctypes.data, v.size, _lib.PRECISIONS[precision], pieces.ctypes.data, C.byref(scale)),
               "peanut_debug_weight_pieces")
    if precision == "fp16x3":
        s = np.float32(scale.value)
        m, e = np.frexp(np.abs(v).max())
        assert s == np.float32(2.0) ** (14 - e) and 2.0 ** 13 <= np.abs(v).max() * s < 2.0 ** 14
        r = v * s                                            # exact: power of two, no overflow / underflow here
        for q in range(2):
            h = r.astype(np.float16)                         # numpy: round to nearest even, subnormals kept
            assert np.array_equal(pieces[q], h.view(np.uint16)), f"piece {q}"
            r = r - h.astype(np.float32)                     # exact
        assert not pieces[2].any()
        # what the two pieces leave: <= 2^-23 relative, or fp16's subnormal spacing
        assert bool((np.abs(r) <= np.maximum(np.abs(v * s) * 2.0 ** -22, 2.0 ** -25)).all())
    else:
        assert scale.value == 1.0
        r = torch.from_numpy(v.copy())
        n = 3 if precision == "bf16x6" else 2
        for q in range(n):
            h = r.to(torch.bfloat16)
            assert np.array_equal(pieces[q], h.view(torch.int16).numpy().view(np.uint16)), f"piece {q}"
            r = r - h.float()
        if n == 3:
            assert float(r.abs().max()) == 0.0 or bool((r.abs() <= torch.from_numpy(np.abs(v)) * 2.0 ** -24).all())
        else:
            assert not pieces[2].any()


# ---- Winograd constants (csrc/winograd.hip): host side pinned here, device side by the GPU tests -------------------------------
WINO_POINTS = {4: ["0", "3/4", "-3/4", "3/2", "-3/2"],                     # finite interpolation points, in position order; + infinity
               5: ["0", "1/2", "-1/2", "1", "-1", "3"],
               6: ["0", "1/2", "-1/2", "1", "-1", "2", "-2"]}


def _toom_cook(m, points):
    """F(m, 3) from its interpolation points in exact rationals: A^T [m][n], G [n][3], B^T [n][n] with n = m + 2, the rows of
    B^T normalised to a last coefficient of 1 (the scale goes into G) as csrc/winograd.hip writes them."""
    from fractions import Fraction as Fr
    pts = [Fr(p) for p in points]
    n = m + 2
    assert len(pts) == n - 1

    def poly_mul(a, b):
        out = [Fr(0)] * (len(a) + len(b) - 1)
        for i, x in enumerate(a):
            for j, y in enumerate(b):
                out[i + j] += x * y
        return out
    AT = [[Fr(0)] * n for _ in range(m)]
    G = [[Fr(0)] * 3 for _ in range(n)]
    BT = [[Fr(0)] * n for _ in range(n)]
    full = [Fr(1)]
    for p in pts:
        full = poly_mul(full, [-p, Fr(1)])                     # prod (x - p_j): degree n - 1
    for i, p in enumerate(pts):
        Ni = Fr(1)
        Mi = [Fr(1)]
        for j, q in enumerate(pts):
            if j != i:
                Ni *= p - q
                Mi = poly_mul(Mi, [-q, Fr(1)])                 # prod_{j != i} (x - p_j): degree n - 2
        # Lagrange form: the row of B^T is M_i(x) up to a sign chosen so that its leading coefficient is +1; 1 / N_i goes to G
        lead = Mi[-1]
        for k in range(n - 1):
            BT[i][k] = Mi[k] / lead
        for k in range(m):
            AT[k][i] = p ** k
        for k in range(3):
            G[i][k] = p ** k / Ni * lead
    for k in range(n):
        BT[n - 1][k] = full[k]
    AT[m - 1][n - 1] = Fr(1)
    G[n - 1][2] = Fr(1)
    return AT, G, BT


@pytest.mark.parametrize("tile", [4, 5, 6])
def test_winograd_weight_transform_matches_the_toom_cook_construction(tile):
    """U = G g G^T as the uploader computes it (peanut_debug_wino_weights: host code of csrc/winograd.hip, no GPU) against
    the Toom-Cook construction of F(tile, 3) from the interpolation points the kernels document -- in exact rationals, so the
    comparison is to the last bit of a double rounded once to fp32 -- and the construction itself against the defining
    identity  A^T [(G g) . (B^T d)] = correlation(d, g)  (which ties G to the B^T / A^T the device kernels implement; those
    are held against F.conv2d by the GPU tests)."""
    import numpy as np
    from fractions import Fraction as Fr
    AT, G, BT = _toom_cook(tile, WINO_POINTS[tile])
    n = tile + 2
    # 1-D identity in exact arithmetic on integer data
    rng = np.random.RandomState(tile)
    d = [Fr(int(v)) for v in rng.randint(-9, 10, n)]
    g = [Fr(int(v)) for v in rng.randint(-9, 10, 3)]
    ref = [sum(d[k + j] * g[j] for j in range(3)) for k in range(tile)]
    Gg = [sum(G[i][k] * g[k] for k in range(3)) for i in range(n)]
    Bd = [sum(BT[i][k] * d[k] for k in range(n)) for i in range(n)]
    y = [sum(AT[k][i] * Gg[i] * Bd[i] for i in range(n)) for k in range(tile)]
    assert y == ref, "Toom-Cook construction does not satisfy the Winograd identity"
    # the library's U against G g G^T
    lib = _lib.load()
    cout, cin = 3, 5
    w = rng.standard_normal((cout, cin, 3, 3)).astype(np.float32)
    out = np.zeros((n * n, cout, cin), np.float32)
    _lib.check(lib.peanut_debug_wino_weights(w.ctypes.data, cout, cin, tile, out.ctypes.data), "peanut_debug_wino_weights")
    Gf = np.array([[float(v) for v in row] for row in G])
    want = np.einsum("ia,ocab,lb->iloc", Gf, w.astype(np.float64), Gf).reshape(n * n, cout, cin)
    assert np.abs(out - want.astype(np.float32)).max() <= 2.0 ** -22 * np.abs(want).max()      # a double product rounded once
    assert np.abs(out.astype(np.float64) - want).max() <= 2.0 ** -23 * np.abs(want).max()
    assert lib.peanut_debug_wino_weights(w.ctypes.data, cout, cin, 7, out.ctypes.data) != 0     # unknown tile size: refused


def test_uncounted_asm_loads_are_not_touched_before_their_wait():
    """csrc/conv_pw_ares.hip hides its residual / scale / shift loads from hipcc's wait-count bookkeeping (inline asm): the
    destination registers are valid only after the kernel's own `s_waitcnt vmcnt(0)`.  tools/audit_uncounted_loads.py
    compiles the file to gfx950 assembly and checks that nothing reads, copies or overwrites such a register between its
    load and that wait (hipcc is free to: cdna_hip_programming.md 5.7).  The audit itself is checked on two synthetic
    listings first: a clean one and one with the copy hipcc once inserted ahead of the wait."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("audit_uncounted_loads", os.path.join(root, "tools", "audit_uncounted_loads.py"))
    audit = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(audit)
    clean = """
	;;#ASMSTART
	global_load_dword v164, v[16:17], off
	;;#ASMEND
	s_cbranch_vccnz .LBB0_2
	v_mfma_f32_32x32x2_f32 v[0:15], v20, v21, v[0:15]
.LBB0_2:
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	v_mov_b32_e32 v81, v164
	s_endpgm
"""
    n, findings = audit.audit(clean)
    assert n == 1 and findings == []
    copied = clean.replace("\tv_mfma_f32_32x32x2_f32 v[0:15], v20, v21, v[0:15]\n", "\tv_mov_b32_e32 v81, v164\n")
    assert len(audit.audit(copied)[1]) == 1
    ranged = clean.replace("v[0:15], v20, v21, v[0:15]", "v[0:15], v20, v21, v[160:175]")
    assert len(audit.audit(ranged)[1]) == 1
    escaping = clean.replace("s_cbranch_vccnz .LBB0_2", "s_cbranch_vccnz .LBB0_9")
    assert len(audit.audit(escaping)[1]) == 1
    for src in audit.DEFAULT:                     # conv_pw_ares.hip AND conv_pw256p.hip (both use uncounted asm loads)
        n, findings = audit.audit(audit.assembly(src))
        assert n >= 100 and findings == [], (src, findings[:5])


def test_no_packed_fp32_instruction_reads_a_high_register_into_its_low_half():
    """gfx950, measured in round 6 (profiles/r9r): `v_pk_fma_f32 ... op_sel:[0,1,0]` -- a packed fp32 instruction whose LOW half takes
    an operand from the HIGH register of a pair -- returns wrong low halves (lanes 48-63) while another wave on the same SIMD issues
    fp16 / bf16 MFMAs; 20 different results in 20 forwards of the two-stream head in the emulated modes, exact through `op_sel_hi`
    alone, through no modifier, unpacked, next to fp32 MFMAs or alone.  hipcc emits the form freely when it packs scalar code.  By
    disassembly of the built library:

      * no kernel of the library contains such an instruction -- the ones in which hipcc had produced it are compiled without packed
        fp32 instructions (csrc/common.h, PEANUT_NO_PK_F32: gemm_skinny_kernel, upsample_logits_kernel, the fallback ppm_conv_term_kernel,
        map_finish_kernel, box_post_kernel) or have the expression pinned to scalar registers (the goal solver's discriminant);
      * the one exception is the test hook that exists to show the behaviour (pkfma_canary_kernel, tests/test_conv_gpu.py);
      * the fenced kernels carry no call either (the attribute keeps helpers compiled without it from being inlined: HIP's
        __syncthreads / atomicAdd / make_float4 wrappers are spelled out in them).
    A new occurrence anywhere fails this test before a GPU does."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    build.build()
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(root, "tools", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    assert kr._OP_SEL.search("v_pk_fma_f32 v[0:1], v[0:1], v[104:105], v[16:17] op_sel:[0,1,0]").group(1) == "0,1,0"
    assert kr._OP_SEL.search("v_pk_fma_f32 v[0:1], v[2:3], v[52:53], v[0:1] op_sel_hi:[1,0,1]") is None
    found = kr.risky_packed_fp32()
    others = [(r["file"], r["name"][:70], r["instruction"]) for r in found if "pkfma_canary_kernel(" not in r["name"]]
    assert others == [], others[:8]
    assert any("pkfma_canary_kernel(" in r["name"] for r in found)          # (the scan sees the form where it is meant to be)
    for obj in ("gemm_skinny", "pspnet_aux", "mapping", "rcnn_post", "goal"):
        assert "s_swappc" not in kr.object_disassembly(os.path.join(kr.BUILD, obj + ".o")), obj
    # the fenced kernel carries no packed fp32 instruction at all
    text = kr.object_disassembly(os.path.join(kr.BUILD, "gemm_skinny.o"))
    cur, packed = None, {}
    for ln in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", ln)
        if m:
            cur = m.group(1)
        elif cur and kr._PK_F32.search(ln):
            packed[cur] = packed.get(cur, 0) + 1
    assert not {n: c for n, c in packed.items() if "gemm_skinny_kernel" in n}, packed


def test_hot_kernels_have_no_spilled_vgprs_and_no_scratch():
    """Register budget of the built library, from the code objects' own metadata (tools/kernel_resources.py; the numbers
    `-Rpass-analysis=kernel-resource-usage` prints): no kernel spills a VGPR, none of the convolution / GEMM / Winograd /
    mapping / goal kernels uses scratch memory, and the MFMA kernels keep the occupancy their LDS plan assumes (two
    workgroups' worth of waves per SIMD).  Spilled SGPRs (v_writelane into a spare VGPR: no memory traffic) are bounded per
    kernel so that growth is noticed.  Round 5's conv_pw_glds256wp_kernel<true, 4> (255 VGPRs + 4 spilled, 20 B scratch) is
    gone: the residual variant exists with two blocks per epilogue group only."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    build.build()
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(root, "tools", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    ks = kr.library_kernels()
    assert len(ks) >= 100
    names = [k["name"] for k in ks]
    assert not any("conv_pw_glds256wp_kernel<true, 4>" in n for n in names)
    assert any("conv_pw_glds256wp_kernel<true, 2>" in n for n in names) and any("conv_pw_glds256wp_kernel<false, 4>" in n for n in names)
    spilled = [(k["file"], k["name"], k["vgpr_spill_count"]) for k in ks if k["vgpr_spill_count"]]
    assert spilled == [], spilled
    # scratch: only the detector's per-box post-processing kernel keeps a small per-thread array (rcnn_post.hip: box_post_kernel)
    scratch = [(k["file"], k["name"], k["private_segment_fixed_size"]) for k in ks if k["private_segment_fixed_size"]]
    assert all(f == "rcnn_post.hip" and "box_post_kernel" in n for f, n, _ in scratch), scratch
    for k in ks:
        assert k["sgpr_spill_count"] <= 90, (k["name"], k["sgpr_spill_count"])
        hot = k["file"] in ("conv_pw256p.hip", "conv_pw256wp.hip", "conv_pw_ares.hip", "conv_pw.hip", "gemm_rs.hip") and "reduce" not in k["name"]
        if hot:
            assert k["waves_per_simd"] >= 2, (k["name"], k["vgpr_count"], k.get("agpr_count"))
