"""CPU-only tests of the host side: the C-ABI library loads and exports every symbol
include/taiyaki_amd_flipflop.h declares (no compute calls without a GPU), the index algebra
and synthetic generators agree with the oracle / the reference's documented examples, the
PyTorch layer stack has the reference's shapes and parameter counts, and the operators fail
loudly on CPU tensors."""
import os
import re

import numpy as np
import pytest
import torch

from tests.conftest import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol():
    from taiyaki_amd import _lib
    _lib.build()
    handle = _lib.lib()
    header = open(os.path.join(ROOT, "include", "taiyaki_amd_flipflop.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(tk_\w+|crf_flipflop_\w+|cat_mod_flipflop_\w+)\s*\(", header))
    assert len(declared) >= 12
    # the RCCL entry points live in a library of their own (csrc/rccl_api.cpp): its dynamic symbol
    # table is read with nm -- loading it would pull a second RCCL into a process that has PyTorch's
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", os.path.join(_lib.CSRC, _lib.RCCL_LIBNAME)],
                        capture_output=True, text=True, check=True).stdout
    rccl_exported = set(re.findall(r" T (tk_\w+)", nm))
    assert rccl_exported == set(_lib.RCCL_SIGNATURES)
    for name in declared:
        if name in _lib.RCCL_SIGNATURES:
            continue
        assert hasattr(handle, name), name
        assert name in _lib.SIGNATURES, "binding missing for %s" % name
    assert set(_lib.SIGNATURES) | set(_lib.RCCL_SIGNATURES) == declared
    assert handle.tk_version().startswith(b"taiyaki_amd flipflop gfx950")


def test_workspace_queries_need_no_gpu():
    from taiyaki_amd import _lib
    L = _lib.lib()
    assert L.tk_flipflop_logz_workspace_bytes(4000, 256, 4) > 0
    assert L.tk_flipflop_logz_workspace_bytes(4000, 256, 9) == 0      # nbase not built
    assert L.tk_crf_flipflop_workspace_bytes(40, 800, 128, 480, 1) > \
        L.tk_crf_flipflop_workspace_bytes(40, 800, 128, 480, 0)
    # one traceback byte per lane (8 lanes per read) and step
    assert L.tk_flipflop_viterbi_workspace_bytes(800, 128, 4) == 800 * (128 // 8) * 64


def test_flipflopfings_reference_docstring_examples(oracle_mod):
    """flipflopfings.py:44-47, 70-73 doctests + agreement with the oracle's C version."""
    from taiyaki_amd import flipflopfings as ff
    x = np.array([1, 3, 2, 3, 3, 3, 3, 1, 1])
    np.testing.assert_array_equal(
        ff.flopmask(x), [False, False, False, False, True, False, True, False, True])
    np.testing.assert_array_equal(ff.flipflop_code(x), [1, 3, 2, 3, 7, 3, 7, 1, 5])
    np.testing.assert_array_equal(ff.flipflop_code(x), oracle_mod.flipflop_code(x))
    codes = ff.flipflop_code(np.array([0, 0, 1, 1, 1, 3, 2, 2]))
    move, stay = oracle_mod.flipflop_indices(codes, [len(codes)], 4)
    np.testing.assert_array_equal(ff.stay_indices(codes), stay[:len(codes)])
    np.testing.assert_array_equal(ff.move_indices(codes), move[:len(codes) - 1])
    assert ff.nstate_flipflop(4) == 40 and ff.nbase_flipflop(40) == 4
    with pytest.raises(AssertionError):
        ff.nbase_flipflop(41)
    assert ff.path_to_str(np.array([0, 0, 1, 5, 5, 2])) == "ACCG"


def test_synth_is_deterministic_and_matches_speedtest_rule():
    from taiyaki_amd import synth
    a = synth.crf_case(50, 6, 3)
    b = synth.crf_case(50, 6, 3)
    for k in a:
        np.testing.assert_array_equal(a[k], b[k])
    assert a["scores"].dtype == np.float32 and a["scores"].min() >= -5 and a["scores"].max() < 5
    # c_crf_flipflop.c:813: nblock * (1 + (i - nbatch/2) / (5 nbatch)) / 2
    np.testing.assert_array_equal(synth.speedtest_seqlens(800, 128)[[0, 64, 127]], [360, 400, 439])
    assert a["seqs"].max() < 8 and len(a["seqs"]) == a["seqlens"].sum()
    m = synth.crf_case(50, 6, 3, nmods_per_base=(1, 1, 0, 0))
    assert m["scores"].shape[2] == 46
    np.testing.assert_array_equal(m["can_mods_offsets"], [0, 2, 4, 5, 6])   # layers.py:1495-1497
    bases = m["seqs"] % 4
    assert np.all(m["mod_cats"][bases >= 2] == 0) and m["mod_cats"].max() == 1


def test_model_shapes_and_parameter_counts():
    """(4000,2,1) -> (800,2,40) / (800,2,46); 2,715,280 trainable parameters."""
    from taiyaki_amd import models
    x = torch.randn(4000, 2, 1)
    net = models.mLstm_flipflop()
    assert net(x).shape == (800, 2, 40)
    assert sum(p.numel() for p in net.parameters() if p.requires_grad) == 2715280
    assert sum(p.numel() for p in net.parameters()) == 2720400
    out = models.mLstm_cat_mod_flipflop()(x)
    assert out.shape == (800, 2, 46)
    assert float(out[:, :, :40].detach().abs().max()) <= 5.0
    # per-base log-softmax groups A,(6mA) C,(5mC) G T  (layers.py:1616-1640)
    np.testing.assert_allclose(out[:, :, 40:42].exp().sum(2).detach().numpy(), 1.0, atol=1e-5)
    np.testing.assert_allclose(out[:, :, 44:].detach().numpy(), 0.0, atol=1e-6)
    assert models.mGru_flipflop(size=96)(torch.randn(2000, 2, 1)).shape == (1000, 2, 40)


def test_operators_refuse_cpu_tensors():
    from taiyaki_amd import ctc, decode, layers
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        layers.flipflop_logpartition(torch.zeros(4, 1, 40))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ctc.crf_flipflop_loss(torch.zeros(4, 1, 40), torch.tensor([0, 1]), torch.tensor([2]), 1.0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        decode.flipflop_viterbi(torch.zeros(4, 1, 40))


def test_product_code_never_imports_the_oracle():
    """The package, the C ABI header and the tools never touch oracle/ or the reference tree."""
    for top in ("taiyaki_amd", "include", "tools"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".hip", ".h")):
                    src = open(os.path.join(dirpath, f)).read()
                    assert not re.search(r"^\s*(import oracle|from oracle)", src, flags=re.M), f
                    assert "liboracle" not in src and "/root/reference" not in src, f


@pytest.mark.parametrize("cin,cout,k,stride", [(1, 4, 5, 1), (4, 16, 5, 1), (16, 32, 19, 5), (3, 8, 4, 2)])
def test_convolution_gemm_form_equals_conv1d(cin, cout, k, stride):
    """The unfold+GEMM evaluation of `Convolution` is the reference layer
    (layers.py:816-831: pad, nn.Conv1d, activation) value- and gradient-wise."""
    from taiyaki_amd import layers
    torch.manual_seed(3)
    conv = layers.Convolution(cin, cout, k, stride=stride, fun=layers.swish)
    x = torch.randn(203, 3, cin, requires_grad=True)
    outs = []
    for use_gemm in (True, False):
        conv.use_gemm = use_gemm
        y = conv(x)
        gx, gw, gb = torch.autograd.grad(y.square().mean(), [x, conv.conv.weight, conv.conv.bias])
        outs.append((y.detach(), gx, gw, gb))
    for a, b in zip(*outs):
        assert a.shape == b.shape
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6)


def test_c_abi_rejects_bad_arguments_before_touching_the_gpu():
    """Argument validation happens before any HIP call: NULL pointers, zero sizes and a
    one-sided fwd/traceback pair return TK_ERR_BAD_ARG (1) on a box without a GPU too."""
    from taiyaki_amd import _lib
    L = _lib.lib()
    BAD = 1
    one = 16        # any non-NULL, 16-byte aligned value: never dereferenced on these paths
    assert L.tk_flipflop_logz_dev(None, 10, 2, 4, None, None, None, 0, None, None) == BAD
    assert L.tk_flipflop_viterbi_dev(None, 10, 2, 4, None, None, None, None, 0, None) == BAD
    # fwd without traceback (or the reverse) is refused
    assert L.tk_flipflop_viterbi_dev(one, 10, 2, 4, one, None, one, one, 1 << 20, None) == BAD
    assert L.tk_flipflop_errprobs_dev(None, None, 10, 2, 4, None, None) == BAD
    assert L.tk_flipflop_errprobs_dev(one, one, 0, 2, 4, one, None) == BAD
    assert L.tk_grad_maxabs_clip_dev(None, None, 3, 10, None, None, None) == BAD
    assert L.tk_crf_flipflop_dev(None, 40, 10, 2, None, None, None, None, None, None, 0, 40,
                                 1.0, 1.0, 1.0, None, None, None, 0, None, None, None) == BAD
    assert L.tk_flipflop_build_indices_dev(None, None, 0, 0, 4, None, None, None, None, None, None,
                                           None, None, None, None) == BAD


def test_catmod_producer_layer_matches_reference_golden():
    """`GlobalNormFlipFlopCatMod.forward` (taiyaki/layers.py:1616-1640) value-pinned: the
    reference layer's weights, input and output (generated by importing the reference,
    tests/golden/make_golden_catmod_layer.py) against the restated layer -- output order AYCZGT:
    [5 tanh(40 transition scores), log_softmax{A, 6mA}, log_softmax{C, 5mC}, 0 (G), 0 (T)]."""
    from taiyaki_amd import layers
    gold = load_golden("catmod_layer.npz")
    for name in ("t9n3", "t60n4", "t40n2_sharp"):
        W, b, x, y = (gold[name + "/" + k] for k in ("W", "b", "x", "y"))
        lay = layers.GlobalNormFlipFlopCatMod(W.shape[1], tuple(int(v) for v in gold[name + "/can_nmods"]))
        assert lay.linear.weight.shape == W.shape and lay.nout == y.shape[2] == 46
        lay.load_state_dict({"linear.weight": torch.tensor(W), "linear.bias": torch.tensor(b)}, strict=False)
        got = lay(torch.tensor(x)).detach().numpy()
        np.testing.assert_allclose(got, y, rtol=2e-6, atol=2e-6)
        np.testing.assert_array_equal(lay.can_mods_offsets, gold[name + "/can_mods_offsets"])
        # canonical-only bases (G, T) carry a constant 0 = log(1) in their category slot
        assert np.all(got[:, :, 44:] == 0.0)


def _calculate_loss_like_the_reference(net_outputs, seqs, seqlens, sharpen=1.0):
    """The four lines of bin/train_flipflop.py:161-182 written against the REFERENCE's names."""
    from taiyaki import ctc, layers
    nblk = float(net_outputs.shape[0])
    lossvector = ctc.crf_flipflop_loss(net_outputs, seqs, seqlens, sharpen)
    lossvector = lossvector + layers.flipflop_logpartition(net_outputs) / nblk
    return lossvector.mean()


def test_shim_standalone_resolves_reference_names_to_the_hip_operators():
    """`taiyaki_amd.shim.install()` with no reference package around: a caller written against
    `taiyaki.ctc` / `taiyaki.layers` / `taiyaki.decode` resolves to the HIP operators (which
    refuse CPU tensors: no fallback)."""
    import sys
    from taiyaki_amd import ctc, decode, layers, shim
    assert shim.install(force_standalone=True) == "standalone"
    try:
        import taiyaki
        from taiyaki import ctc as tctc, decode as tdecode, layers as tlayers
        from taiyaki.flipflopfings import flipflop_code, nstate_flipflop
        from taiyaki.maths import RollingMAD
        assert tctc.crf_flipflop_loss is ctc.crf_flipflop_loss
        assert tctc.cat_mod_flipflop_loss is ctc.cat_mod_flipflop_loss
        assert tlayers.flipflop_logpartition is layers.flipflop_logpartition
        assert tdecode.flipflop_viterbi is decode.flipflop_viterbi
        assert tdecode.flipflop_make_trans is decode.flipflop_make_trans
        from taiyaki import decodeutil as tdu                   # bin/basecall.py:10,218
        from taiyaki_amd import decodeutil
        assert tdu.beamsearch is decodeutil.beamsearch and tdu.forward is decodeutil.forward
        assert tdu.backward is decodeutil.backward
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            tdu.beamsearch(np.zeros((5, 40), dtype=np.float32), 0.0, 5, True)
        assert nstate_flipflop(4) == 40 and RollingMAD(3, 0, 5).nparams == 3
        assert list(flipflop_code(np.array([0, 0, 1, 1, 1]), 4)) == [0, 4, 1, 5, 1]
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            _calculate_loss_like_the_reference(torch.zeros(6, 2, 40), torch.tensor([0, 1, 2]), torch.tensor([2, 1]))
        assert getattr(taiyaki, "__taiyaki_amd_shim__", False)
    finally:
        shim.uninstall()
    assert "taiyaki" not in sys.modules or not getattr(sys.modules["taiyaki"], "__taiyaki_amd_shim__", False)


def test_shim_patches_an_importable_reference_package(tmp_path, monkeypatch):
    """With a `taiyaki` package on the path (here a stand-in with the reference's module layout:
    `ctc` extension module, `layers.flipflop_logpartition`, `decode.flipflop_viterbi`), install()
    re-points exactly the hot-path names and uninstall() restores them."""
    import importlib
    import sys
    pkg = tmp_path / "taiyaki"
    (pkg / "ctc").mkdir(parents=True)
    (pkg / "__init__.py").write_text("")
    (pkg / "ctc" / "__init__.py").write_text("def crf_flipflop_loss(*a):\n    return 'cpu-extension'\n")
    (pkg / "layers.py").write_text("def flipflop_logpartition(x, _never_use_cupy=False):\n    return 'torch-loop'\n"
                                   "def other():\n    return 'untouched'\n")
    (pkg / "decode.py").write_text("def flipflop_viterbi(s, _never_use_cupy=False):\n    return 'torch-loop'\n"
                                   "def flipflop_make_trans(s, _never_use_cupy=False):\n    return 'torch-loop'\n")
    monkeypatch.syspath_prepend(str(tmp_path))
    for k in [k for k in sys.modules if k == "taiyaki" or k.startswith("taiyaki.")]:
        monkeypatch.delitem(sys.modules, k)
    importlib.invalidate_caches()
    from taiyaki_amd import ctc, layers, shim
    assert shim.install() == "patched"
    try:
        import taiyaki.layers as tl
        from taiyaki import ctc as tctc
        assert tctc is ctc and tl.flipflop_logpartition is layers.flipflop_logpartition
        assert tl.other() == "untouched"
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            _calculate_loss_like_the_reference(torch.zeros(6, 2, 40), torch.tensor([0, 1, 2]), torch.tensor([2, 1]))
    finally:
        shim.uninstall()
    import taiyaki.layers as tl2
    assert tl2.flipflop_logpartition(None) == "torch-loop"
    for k in [k for k in sys.modules if k == "taiyaki" or k.startswith("taiyaki.")]:
        sys.modules.pop(k, None)


def test_beam_lse_tables_in_the_kernel_are_the_generators():
    """The 162 table entries and the three reduction constants of the beam search's log-sum-exp
    (csrc/beam_kernels.hip: BEAM_TAB, LN2_32_HI / _LO, INV_LN2_32) are exactly what
    tools/beam_lse_tables.py derives from 60-digit decimals -- an edited digit would still pass every
    parity test with a tolerance and quietly lose the bit-for-bit agreement with glibc."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gen = subprocess.run([sys.executable, os.path.join(root, "tools", "beam_lse_tables.py")], capture_output=True,
                         text=True, check=True).stdout
    want = re.findall(r"-?0x[0-9a-f.]+p[+-]\d+", gen)
    src = open(os.path.join(root, "taiyaki_amd", "csrc", "beam_kernels.hip")).read()
    tab = src[src.index("__constant__ double BEAM_TAB"):]
    tab = tab[:tab.index("// tables into LDS")]
    have = re.findall(r"-?0x[0-9a-f.]+p[+-]\d+", tab)
    assert len(want) == 162 + 3 and have == want


def test_beam_lse_host_twin_matches_glibc(tmp_path):
    """tools/beam_lse_check.c is the host twin of the kernel's table-driven expf / log1pf: on every 101st
    float in [0, 17) (11 M arguments; the full every-third sweep is a 16 s run of the same program) it
    must give bit for bit what glibc's double exp / log1p give after rounding to float."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = subprocess.run([sys.executable, os.path.join(root, "tools", "beam_lse_tables.py")], capture_output=True,
                         text=True, check=True).stdout
    (tmp_path / "beam_lse_tables.h").write_text(hdr)
    exe = str(tmp_path / "beam_lse_check")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-DSTRIDE=101", "-I", str(tmp_path), "-o", exe,
                    os.path.join(root, "tools", "beam_lse_check.c"), "-lm"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout
    assert "expf mismatches 0, log1pf mismatches 0" in out, out


def test_shim_offers_every_hot_path_name_of_the_reference_modules():
    """tests/golden/reference_names.json lists the public names of the reference modules the shim
    stands in for (read from their source by make_reference_names.py).  After install() every name
    SURVEY.md section 8 puts on the path resolves -- `taiyaki.ctc` and `taiyaki.decodeutil`, the two
    compiled extensions, completely (ctc.pyx:13-312: the numpy-level cost / grad functions too)."""
    import json
    from taiyaki_amd import shim
    root = os.path.dirname(os.path.abspath(__file__))
    names = json.load(open(os.path.join(root, "golden", "reference_names.json")))
    assert shim.install(force_standalone=True) == "standalone"
    try:
        import importlib
        for mod, d in names.items():
            m = importlib.import_module("taiyaki." + mod)
            missing = [n for n in d["hot_path"] if not hasattr(m, n)]
            assert not missing, "taiyaki.%s lacks %s" % (mod, missing)
        import taiyaki.ctc as tctc
        assert sorted(n for n in names["ctc"]["public"] if not hasattr(tctc, n)) == []
        # (N, 1) like the reference's torch statement of logZ, (N,) for the dispatcher
        import inspect
        import taiyaki.layers as tl
        assert list(inspect.signature(tl.log_partition_flipflop).parameters) == ["scores"]
        assert list(inspect.signature(tctc.crf_flipflop_grad).parameters) == [
            "logprob", "moveidxs", "stayidxs", "seqlen", "pin"]
        assert list(inspect.signature(tctc.cat_mod_flipflop_cost).parameters) == [
            "logprob", "moveidxs", "stayidxs", "modmoveidxs", "modmovefacts", "seqlen", "pin"]
    finally:
        shim.uninstall()


def test_library_exports_exactly_what_the_header_declares():
    """`nm -D` of libtaiyaki_amd_flipflop.so == the functions include/taiyaki_amd_flipflop.h declares
    (minus the RCCL ones, which live in libtaiyaki_amd_rccl.so): no dispatcher, kernel stub or helper
    leaks into the drop-in's symbol namespace (csrc/exports.map)."""
    import re
    import subprocess
    from taiyaki_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "taiyaki_amd_flipflop.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"^(?:const\s+)?[a-z_0-9]+\s+\*?\s*([a-z_0-9]+)\(", hdr, flags=re.M))
    assert {"tk_version", "crf_flipflop_grad", "tk_allreduce_f32_dev", "tk_flipflop_loss_fused_dev"} <= declared

    def exported(path):
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        return {ln.split()[2] for ln in out.splitlines() if len(ln.split()) == 3 and ln.split()[1] in "TDBW"}

    rccl = {n for n in declared if n in _lib.RCCL_SIGNATURES}
    assert exported(_lib.LIBPATH) == declared - rccl
    assert exported(os.path.join(_lib.CSRC, _lib.RCCL_LIBNAME)) == rccl
    assert set(_lib.SIGNATURES) <= declared


def test_release_library_carries_no_lab_switch():
    """The shipped library reads no dispatch switch from the environment (round 4's verdict: 16 getenv calls per
    launch path, one of which -- TK_CRF_NO_FALLBACK -- switched off the exact-or-redo guarantee) and exports no
    tk_lab_* hook; the lab build (-DTK_LAB, what tests and tools switch to with _lib.use_lab) has both."""
    import subprocess
    from taiyaki_amd import _lib
    _lib.build()

    def strings(path):
        return subprocess.run(["strings", "-n", "6", path], capture_output=True, text=True, check=True).stdout

    def dynsyms(path):
        return subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout

    rel, lab = strings(_lib.LIBPATH), strings(_lib.LAB_LIBPATH)
    switches = ["TK_CRF_NO_FALLBACK", "TK_CRF_MODE", "TK_CRF_BK", "TK_CRF_WBIAS", "TK_CRF_KLIP", "TK_CRF_BAND_R", "TK_CRF_FEED",
                "TK_CRF_LATTICE_MB", "TK_CRF_GATE_DUMP", "TK_K1_RING", "TK_K1_NT", "TK_LOGZ_CH", "TK_LOGZ_SPLIT",
                "TK_SIDE_PRIO"]
    for name in switches:
        assert name not in rel, name
        assert name in lab, name
    # the one variable the release build reads (once, into a static): the initial two-queue mode of the fused loss
    import re
    assert set(re.findall(r"\bTK_[A-Z0-9_]{3,}\b", rel)) == {"TK_LOSS_OVERLAP"}
    assert "tk_lab_" not in dynsyms(_lib.LIBPATH)
    assert "tk_lab_crf_band_phase" in dynsyms(_lib.LAB_LIBPATH)
    # no source file of the library calls getenv outside the macro and that one static
    for fn in os.listdir(_lib.CSRC):
        if fn.endswith((".hip", ".h", ".cpp")) and fn != "rccl_api.cpp":
            text = open(os.path.join(_lib.CSRC, fn)).read()
            n = len(re.findall(r"\bgetenv\(", text))
            assert n == {"c_api.hip": 1, "ff_common.h": 1}.get(fn, 0), (fn, n)


def test_numpy_level_ctc_functions_keep_the_cython_layers_contract():
    """ctc.pyx:31-113: typed C-contiguous buffers or an error, finite input or AssertionError,
    a state count that is no flip-flop model is an AssertionError -- all before anything is launched
    (so checkable without a GPU)."""
    from taiyaki_amd import ctc
    lp = np.zeros((5, 2, 40), dtype=np.float32)
    mv = np.zeros(3, dtype=np.uintp)
    st = np.zeros(5, dtype=np.uintp)
    sl = np.array([3, 2], dtype=np.int32)
    with pytest.raises(ValueError, match="dtype mismatch"):
        ctc.crf_flipflop_cost(lp.astype(np.float64), mv, st, sl)
    with pytest.raises(ValueError, match="dtype mismatch"):
        ctc.crf_flipflop_cost(lp, mv.astype(np.int32), st, sl)
    with pytest.raises(ValueError, match="C-contiguous"):
        ctc.crf_flipflop_grad(np.zeros((5, 40, 2), dtype=np.float32).transpose(0, 2, 1), mv, st, sl)
    with pytest.raises(TypeError):
        ctc.crf_flipflop_grad(torch.zeros(5, 2, 40), mv, st, sl)
    bad = lp.copy()
    bad[2, 1, 7] = np.inf
    with pytest.raises(AssertionError, match="Input not finite"):
        ctc.crf_flipflop_grad(bad, mv, st, sl)
    with pytest.raises(AssertionError, match="not valid for flip-flop"):
        ctc.crf_flipflop_cost(np.zeros((5, 2, 41), dtype=np.float32), mv, st, sl)
    with pytest.raises(ValueError, match="stayidxs has 4 entries"):
        ctc.crf_flipflop_cost(lp, mv, st[:4], sl)
    assert ctc.nstate_to_nbase(40) == 4 and ctc.nstate_to_nbase(12) == 2


def test_captured_step_refuses_batches_it_cannot_prove_fit():
    """`GraphedTrainer.load` with a launch sized for `max_seqlen`: a batch proves that it fits by the
    hint its `seqlens` tensor carries, or (seqlens on the host) by being looked at; a longer
    sequence is refused BEFORE anything is copied into the static buffers."""
    from taiyaki_amd import ctc, models, parallel, train
    net = models.mLstm_flipflop(size=8, stride=5)
    tr = train.Trainer(net, parallel.FlatGradArena(net))
    ok = dict(indata=torch.zeros(60, 2, 1), seqs=torch.zeros(14, dtype=torch.int32),
              seqlens=torch.tensor([8, 6], dtype=torch.int32))
    g = train.GraphedTrainer(tr, ok, seq_capacity=2 * 13, max_seqlen=10)
    g.load(ok)                                              # host seqlens, no hint: looked at
    assert g.static["seqlens"].tolist() == [8, 6] and g.static["seqlens"].tk_max_seqlen == 10
    long = dict(ok, seqlens=torch.tensor([12, 2], dtype=torch.int32))
    with pytest.raises(ValueError, match="up to 12 bases for a step captured for at most 10"):
        g.load(long)
    assert g.static["seqlens"].tolist() == [8, 6]           # nothing was copied
    hinted = dict(ok, seqlens=ctc.set_max_seqlen(torch.tensor([8, 6], dtype=torch.int32), 11))
    with pytest.raises(ValueError, match="up to 11"):
        g.load(hinted)
    # without a captured bound nothing is checked (the launch is sized for nblk + 1)
    train.GraphedTrainer(tr, ok, seq_capacity=2 * 13).load(long)


def test_crf_workspace_is_sized_for_what_runs():
    """`tk_crf_flipflop_workspace_bytes` (no GPU needed): the checkpoint columns of the band path at
    its block length + an EIGHTH of the batch's worth of log-domain slots (round 3: the whole batch:
    12.2 GB at T = 4000 / N = 256, 203 MB at the train step's shape); a sharpened call keeps more
    columns (shorter blocks) and says so through the `_sharp` query; beyond the linear path's range the
    log-domain kernel's full set."""
    from taiyaki_amd import _lib
    L = _lib.lib()
    MB = 1 << 20
    step = L.tk_crf_flipflop_workspace_bytes(40, 800, 128, 533, 1)
    rowk = L.tk_crf_flipflop_workspace_bytes(40, 4000, 256, 2198, 1)
    # (round 5: 116 MB at the step's shape -- the query bounds the plain CRF's two-cells-per-lane layout, padded to 640 cells
    # per read, and cat-mod's one-cell layout with its third instance array, whichever is larger; round 4: 106 MB)
    # (round 6: + the retry launch's 4-step layout for a sixteenth of the batch: 116 -> 128 MB, 4.7 -> 5.1 GB; a cost-only call
    # carries it too -- 12 MB at the step's shape; the log-domain form's checkpoint columns: one set per workgroup of the tail
    # launch -- a sixteenth of the batch, 16 waves wide: 140 MB / 4.6 GB)
    assert step < 144 * MB and rowk < 5200 * MB, (step / MB, rowk / MB)
    assert L.tk_crf_flipflop_workspace_bytes_sharp(40, 800, 128, 533, 1, 1.0) == step
    s2 = L.tk_crf_flipflop_workspace_bytes_sharp(40, 800, 128, 533, 1, 2.0)
    assert step < s2 < 2 * step + 64 * MB
    s9 = L.tk_crf_flipflop_workspace_bytes_sharp(40, 800, 128, 533, 1, 9.0)       # log-domain kernel on every read
    assert 0 < s9 < 200 * MB
    assert L.tk_crf_flipflop_workspace_bytes(40, 800, 128, 533, 0) < 16 * MB     # cost only: the gate, no checkpoint column
    # cost-only calls keep no checkpoint column in either kernel: the same small size whatever the factor
    assert L.tk_crf_flipflop_workspace_bytes_sharp(40, 800, 128, 533, 0, 9.0) < 16 * MB


def test_gate_count_is_read_unsigned_and_taken_out_exactly():
    """Round-4 advisor finding: the status words are int32 tensors, the kernels add their counts into bits 8-31 --
    2^23 or more read back negative; and the count was cleared with an AND of whatever the word held by then.  Read
    as uint32; take out exactly what was read (flags and later counts stay).  Round 6: two counts of 12 bits -- reads
    redone in the log domain (bits 8-19) and reads retried on the linear path (bits 20-31)."""
    from taiyaki_amd import _lib
    was = _lib.is_strict()
    try:
        _lib.set_strict(False)
        t = _lib.status_word(torch.device("cpu"))
        t.fill_(-(1 << 31) + (7 << 20) + (5 << 8) + 2)  # 2^11 + 7 retried, 5 redone, flag 2 (gradients not finite)
        assert _lib.take_gate_count() == 5 == _lib.last_gate_count() and _lib.last_retry_count() == (1 << 11) + 7
        assert int(t.item()) == 2
        t.add_((3 << 8) + (4 << 20))
        assert _lib.take_gate_count() == 3 and _lib.last_retry_count() == 4
        with pytest.raises(AssertionError, match="Gradients not finite"):
            _lib.raise_if_nonfinite()
        assert int(t.item()) == 0
    finally:
        _lib._deferred.pop(("cpu", None), None)
        _lib.set_strict(was)


def test_bulk_length_hint_host_logic():
    """Round 6: `tk_seq_labels.bulk_seqlen` -- a length all but a sixteenth of the batch (at least one read) stay below -- picks the CRF
    launch's block configuration where `max_seqlen` only sizes it.  Host side: computed from lengths that live on the host, taken from
    the hint where the tensor carries one, unknown (0) otherwise; the ctypes struct follows the header's seven fields."""
    import ctypes
    from taiyaki_amd import _lib, ctc
    assert ctc.bulk_of([]) == 0 and ctc.bulk_of([7]) == 7
    assert ctc.bulk_of([400, 410, 744, 420]) == 420                     # one long read of four set aside
    lens = np.array([430] * 120 + [744] * 8)
    assert ctc.bulk_of(lens) == 430                                     # 8 of 128 = a sixteenth: set aside
    assert ctc.bulk_of(np.array([430] * 119 + [744] * 9)) == 744        # one more: the bulk itself is long
    t = torch.tensor([400, 410, 744, 420])
    assert ctc._bulk_seqlen(t) == 420 and ctc._max_seqlen(t) == 744
    assert ctc._bulk_seqlen(ctc.set_max_seqlen(torch.tensor([400, 410, 744, 420]), 744)) == 0          # a maximum, no bulk: unknown
    assert ctc._bulk_seqlen(ctc.set_max_seqlen(torch.tensor([400, 410, 744, 420]), 744, bulk=450)) == 450
    assert ctc._bulk_seqlen(torch.zeros(0, dtype=torch.int64)) == 0
    assert ctypes.sizeof(_lib.SeqLabels) == 7 * ctypes.sizeof(ctypes.c_void_p)
    hdr = open(os.path.join(ROOT, "include", "taiyaki_amd_flipflop.h")).read()
    body = hdr[hdr.index("typedef struct tk_seq_labels {"):hdr.index("} tk_seq_labels;")]
    assert [f for f in ("seqs;", "total_len;", "nbase;", "mod_cats;", "can_mods_offsets;", "mod_cat_weights;", "bulk_seqlen;")
            if f not in body] == []
    assert "TK_STATUS_RETRIED_SHIFT 20" in hdr and "TK_STATUS_GATED_SHIFT 8" in hdr and _lib._RETRIED_SHIFT == 20 and _lib._GATED_SHIFT == 8
