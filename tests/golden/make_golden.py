#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the GENUINE reference (this container only).

    python tests/golden/make_golden.py [--skip-fullsize | --only-fullsize NAME ...]

What it does (nothing from /root/reference is copied into the repo):
 1. copies /root/reference/taiyaki into a scratch dir under /tmp and builds the
    reference's own Cython extension ``taiyaki.ctc.ctc`` there with two
    build-only adaptations for Cython 3 / numpy 2 (no algorithmic change):
    ``include_path`` so ``cimport libctc`` resolves, and ``<size_t*>`` casts of the
    ``uintp`` index buffers;
 2. imports ``taiyaki.ctc``, ``taiyaki.layers``, ``taiyaki.decode`` from the scratch
    copy and evaluates them on the inputs of tests/golden/cases.py (regenerated
    from taiyaki_amd.synth) and on the reference's own unit-test vectors;
 3. writes the OUTPUTS (and a few small inputs) as .npz fixtures.
The fixtures are data: inputs and expected outputs only.
"""
import argparse
import os
import re
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"
SCRATCH = "/tmp/taiyaki_refbuild"

SETUP = '''
from setuptools import setup, Extension
from Cython.Build import cythonize
import numpy as np
ext = Extension("taiyaki.ctc.ctc",
    sources=["taiyaki/ctc/ctc.pyx", "taiyaki/ctc/c_crf_flipflop.c",
             "taiyaki/ctc/c_cat_mod_flipflop.c"],
    include_dirs=[np.get_include(), "taiyaki/ctc"],
    extra_compile_args=["-O3", "-fopenmp", "-std=c11", "-mavx2", "-D_GNU_SOURCE"],
    extra_link_args=["-fopenmp"])
setup(name="refctc", ext_modules=cythonize([ext], include_path=["taiyaki/ctc"],
                                           language_level=3))
'''


def build_reference():
    so = [f for f in (os.listdir(os.path.join(SCRATCH, "taiyaki", "ctc"))
                      if os.path.isdir(os.path.join(SCRATCH, "taiyaki", "ctc")) else [])
          if f.startswith("ctc.") and f.endswith(".so")]
    if not so:
        shutil.rmtree(SCRATCH, ignore_errors=True)
        os.makedirs(SCRATCH)
        shutil.copytree(os.path.join(REF, "taiyaki"), os.path.join(SCRATCH, "taiyaki"))
        pyx = os.path.join(SCRATCH, "taiyaki", "ctc", "ctc.pyx")
        src = open(pyx).read()
        for name in ("moveidxs", "stayidxs", "modmoveidxs"):
            src = src.replace("&%s[0]" % name, "<size_t*>&%s[0]" % name)
        open(pyx, "w").write(src)
        open(os.path.join(SCRATCH, "setup_ctc.py"), "w").write(SETUP)
        subprocess.run([sys.executable, "setup_ctc.py", "build_ext", "--inplace"],
                       cwd=SCRATCH, check=True, stdout=subprocess.DEVNULL)
    sys.path.insert(0, SCRATCH)


def c_array(text, name):
    """Numeric initialiser of a C array called `name` (test DATA of the embedded
    known-answer harnesses)."""
    m = re.search(name + r"\[\d*\]\s*=\s*\{(.*?)\};", text, re.S)
    body = re.sub(r"//[^\n]*", "", m.group(1))
    return np.array([float(x) for x in re.findall(r"[-+]?\d*\.?\d+(?:[eE][-+]?\d+)?", body)])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-fullsize", action="store_true")
    ap.add_argument("--only-fullsize", nargs="+", default=None, metavar="NAME",
                    help="compute only these cases.FULLSIZE entries and merge them into fullsize.npz")
    args = ap.parse_args()

    build_reference()
    import torch
    from taiyaki import ctc, decode, layers
    from taiyaki.constants import SMALL_VAL
    from tests.golden import cases

    torch.set_num_threads(8)
    # (--only-fullsize leaves the small fixtures as they are)
    save_small = (lambda *a, **k: None) if args.only_fullsize else np.savez_compressed

    def t(x, dtype=None):
        return torch.tensor(np.asarray(x), dtype=dtype)

    def run_crf(inp, sharp):
        x = t(inp["scores"]).requires_grad_()
        loss = ctc.crf_flipflop_loss(x, t(inp["seqs"]), t(inp["seqlens"]), sharp)
        loss.sum().backward()
        return loss.detach().numpy(), x.grad.numpy()

    def run_catmod(inp, sharp):
        x = t(inp["scores"]).requires_grad_()
        loss = ctc.cat_mod_flipflop_loss(
            x, t(inp["seqs"]), t(inp["seqlens"]), t(inp["mod_cats"]),
            inp["can_mods_offsets"], inp["mod_cat_weights"], sharp)
        loss.sum().backward()
        return loss.detach().numpy(), x.grad.numpy()

    def run_logz(scores):
        x = t(scores).requires_grad_()
        lz = layers.flipflop_logpartition(x)
        lz.sum().backward()
        return lz.detach().numpy(), x.grad.numpy()

    def store_grad(out, prefix, grad):
        """full tensor when small, else checksums + strided sample"""
        if grad.size <= 20000:
            out[prefix] = grad
        else:
            for k, v in cases.grad_checksums(grad).items():
                out[prefix + "_" + k] = v

    # ---- small full-tensor cases -------------------------------------------
    out = {}
    for name, spec in cases.CRF_SMALL.items():
        loss, grad = run_crf(cases.crf_inputs(spec), spec["sharp"])
        out[name + "/loss"] = loss
        store_grad(out, name + "/grad", grad)
        xc = t(cases.crf_inputs(spec)["scores"])
        inp = cases.crf_inputs(spec)
        out[name + "/loss_nograd"] = ctc.crf_flipflop_loss(
            xc, t(inp["seqs"]), t(inp["seqlens"]), spec["sharp"]).numpy()
    save_small(os.path.join(HERE, "crf_small.npz"), **out)

    out = {}
    for name, spec in cases.CATMOD_SMALL.items():
        loss, grad = run_catmod(cases.crf_inputs(spec, cases.NMODS), spec["sharp"])
        out[name + "/loss"] = loss
        store_grad(out, name + "/grad", grad)
    save_small(os.path.join(HERE, "catmod_small.npz"), **out)

    out = {}
    for name, spec in cases.LOGZ_SMALL.items():
        sc = cases.logz_inputs(spec)
        lz, grad = run_logz(sc)
        fwd, tb, path = decode.flipflop_viterbi(t(sc))
        trans = decode.flipflop_make_trans(t(sc))
        big = sc.size > 20000
        out[name + "/logz"] = lz
        out[name + "/path"] = path.numpy().astype(np.int8)
        out[name + "/fwd_last"] = fwd[-1].numpy()
        if big:
            cs = cases.grad_checksums(grad)
            for k, v in cs.items():
                out[name + "/grad_" + k] = v
            out[name + "/tb_sum"] = tb.numpy().sum(axis=(0, 2))
        else:
            out[name + "/grad"] = grad
            out[name + "/trans"] = trans.numpy()
            out[name + "/fwd"] = fwd.numpy()
            out[name + "/tb"] = tb.numpy().astype(np.int8)
    save_small(os.path.join(HERE, "logz_small.npz"), **out)

    # ---- the reference's own known-answer vectors ---------------------------
    ka = {}
    # test/unit/test_decodeutil.py:16-18 -> 27.16876983642578
    np.random.seed(0xdeadbeef)
    w = np.random.randn(12, 40).astype("f4")
    ka["decodeutil/weights"] = w
    ka["decodeutil/expt_score"] = np.float64(27.16876983642578)
    ka["decodeutil/tensor_score"] = np.float64(
        float(layers.log_partition_flipflop(t(w).unsqueeze(1)).detach()))
    # test/unit/test_decode.py:20-32
    dsc = np.array([
        [[0, 1, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0]],
        [[0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0]],
        [[0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0]],
        [[0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0]],
        [[0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0]],
        [[0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]],
        [[1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]]], dtype="f4")
    ka["decode/scores"] = dsc
    ka["decode/expected_path"] = np.array([1, 0, 2, 2, 1, 1, 0, 0], dtype=np.int64)
    f, tb, p = decode.flipflop_viterbi(t(dsc))
    ka["decode/fwd"], ka["decode/tb"], ka["decode/path"] = f.numpy(), tb.numpy(), p.numpy()
    ka["decode/trans"] = decode.flipflop_make_trans(t(dsc)).numpy()
    # all-zero scores pin the tie rule
    z = np.zeros((5, 2, 40), dtype="f4")
    f, tb, p = decode.flipflop_viterbi(t(z))
    ka["ties/fwd"], ka["ties/tb"], ka["ties/path"] = f.numpy(), tb.numpy(), p.numpy()
    # test/unit/test_ctc_loss.py:39-103: two paths of probability 1/2 each
    nb, nblocks = 4, 4
    outputs = torch.zeros(nblocks, 1, 40, dtype=torch.float)
    paths = {"015": [0, 0, 1, 5, 5], "237": [2, 2, 3, 7, 7]}
    weights = {"015": [1.0, 1.0, 0.5, 1.0], "237": [1.0, 0.5, 1.0, 1.0]}

    def tcode(fr, to):
        return to * 2 * nb + fr if to < nb else 2 * nb * nb + fr
    for k in paths:
        for b in range(nblocks):
            outputs[b, 0, tcode(paths[k][b], paths[k][b + 1])] = weights[k][b]
    outputs = layers.global_norm_flipflop(torch.log(outputs + SMALL_VAL)).detach()
    ka["ctcloss/outputs"] = outputs.numpy()
    ka["ctcloss/logpart"] = np.float64(float(layers.log_partition_flipflop(outputs).detach()))
    for name, seq, prob in (("015", [0, 1, 5], 0.5), ("237", [2, 3, 7], 0.5),
                            ("510", [5, 1, 0], 0.0)):
        x = outputs.clone().detach().requires_grad_()
        lv = ctc.crf_flipflop_loss(x, t(seq), t([3]), 1.0)
        lv.sum().backward()
        ka["ctcloss/%s_seq" % name] = np.array(seq, dtype=np.int64)
        ka["ctcloss/%s_prob" % name] = np.float64(prob)
        ka["ctcloss/%s_loss" % name] = lv.detach().numpy()
        ka["ctcloss/%s_grad" % name] = x.grad.numpy()
    # embedded C harness data: c_crf_flipflop.c:520-695, c_cat_mod_flipflop.c:589-794
    src = open(os.path.join(REF, "taiyaki/ctc/c_crf_flipflop.c")).read()
    src = src[src.index("#ifdef CRF_TWOSTATE_TEST"):]
    ka["ccrf/logprob"] = np.log(c_array(src, "test_logprob1").astype("f4")).reshape(7, 2, 40)
    ka["ccrf/move"] = c_array(src, "test_move1").astype(np.int64)[:10]
    ka["ccrf/stay"] = c_array(src, "test_stay1").astype(np.int64)
    ka["ccrf/seq"] = c_array(src, "test_seq1").astype(np.int64)
    ka["ccrf/seqlen"] = c_array(src, "test_seqlen1").astype(np.int32)
    ka["ccrf/score"] = np.array([-2.378088, -2.378088])
    src = open(os.path.join(REF, "taiyaki/ctc/c_cat_mod_flipflop.c")).read()
    src = src[src.index("#ifdef CAT_MOD_FLIPFLOP_TEST"):]
    ka["ccm/logprob"] = np.log(c_array(src, "test_logprob1").astype("f4")).reshape(7, 2, 45)
    ka["ccm/move"] = c_array(src, "test_move1").astype(np.int64)[:10]
    ka["ccm/stay"] = c_array(src, "test_stay1").astype(np.int64)
    ka["ccm/modmoveidx"] = c_array(src, "test_modmoveidx1").astype(np.int64)[:10]
    ka["ccm/modmovefact"] = c_array(src, "test_modmovefact1").astype("f4")[:10]
    ka["ccm/seqlen"] = c_array(src, "test_seqlen1").astype(np.int32)
    ka["ccm/score"] = np.array([-52.354622, -195.435257])
    save_small(os.path.join(HERE, "known_answers.npz"), **ka)

    # ---- full-size BASELINE configs: scalars + checksums only ---------------
    if not args.skip_fullsize:
        fs = {}
        if args.only_fullsize:      # add / refresh the named configurations, keep the others' numbers
            fs = dict(np.load(os.path.join(HERE, "fullsize.npz")))
        for name, spec in cases.FULLSIZE.items():
            if args.only_fullsize and name not in args.only_fullsize:
                continue
            inp = synth_case(spec)
            if spec["mods"] is None:
                loss, grad = run_crf(inp, 1.0)
                sc40 = inp["scores"]
            else:
                loss, grad = run_catmod(inp, 1.0)
                sc40 = np.ascontiguousarray(inp["scores"][:, :, :40])
            fs[name + "/loss"] = loss
            for k, v in cases.grad_checksums(grad).items():
                fs[name + "/grad_" + k] = v
            lz, lgrad = run_logz(sc40)
            fs[name + "/logz"] = lz
            for k, v in cases.grad_checksums(lgrad).items():
                fs[name + "/lgrad_" + k] = v
            _, _, path = decode.flipflop_viterbi(t(sc40))
            p = path.numpy()
            fs[name + "/path_hash"] = (p.astype(np.int64) *
                                       (1 + np.arange(p.shape[0])[:, None] % 1009)
                                       ).sum(axis=0)
            # calculate_loss assembly (bin/train_flipflop.py:172-182)
            fs[name + "/lossvector"] = loss + lz / spec["T"]
            print(name, "loss[0..2]", loss[:3], "logz[0..2]", lz[:3], flush=True)
        np.savez_compressed(os.path.join(HERE, "fullsize.npz"), **fs)


def synth_case(spec):
    from tests.golden import cases
    return cases.crf_inputs(dict(T=spec["T"], N=spec["N"], seed=spec["seed"], lsm=spec.get("lsm")),
                            spec["mods"])


if __name__ == "__main__":
    main()
