"""Lab: which reads does the linear band path disown, and are cost-only calls right on them (round 5: a cost-only
call used to believe a finite forward score; wide cat-mod alphabets leave the linear range).  GPU box, lab build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from taiyaki_amd import _lib, ctc, synth
import oracle
from tests import parity
_lib.use_lab(True)
os.environ["TK_CRF_GATE_DUMP"] = "1"
dev = torch.device("cuda:0")
for mods, scale in (((2, 2, 1, 0), 0.2), ((3, 2, 2, 1), 0.2), ((5, 5, 4, 4), 0.2), ((1, 1, 0, 0), 0.2), ((1, 1, 0, 0), 1.0)):
    T, N = 400, 24
    seqlens = synth.realistic_seqlens(T, N, 17000, T * 5, 9.0)
    inp = synth.crf_case(T, N, 13, seqlens=seqlens, nmods_per_base=mods)
    synth.normalise_mod_columns(inp, logit_scale=scale)
    x = torch.from_numpy(inp["scores"]).to(dev)
    seqs, sl = torch.from_numpy(inp["seqs"]), torch.from_numpy(inp["seqlens"])
    extra = (torch.from_numpy(inp["mod_cats"]), inp["can_mods_offsets"], inp["mod_cat_weights"])
    oloss, _ = parity.oracle_crf(oracle, inp, 1.0, want_grad=False)
    print("== mods", mods, "scale", scale, flush=True)
    c, g = ctc._run(x, seqs, sl, 1.0, 1.0, 1.0, 40, True, *extra); torch.cuda.synchronize()
    print("grad call: gated", ctc.last_gate_count(), "max cost err", np.abs(c.cpu().numpy() - oloss).max(), flush=True)
    c0, _ = ctc._run(x, seqs, sl, 1.0, 1.0, 1.0, 40, False, *extra); torch.cuda.synchronize()
    err = np.abs(c0.cpu().numpy() - oloss)
    print("cost-only call: gated", ctc.last_gate_count(), "max cost err", err.max(), "bad reads", np.nonzero(err > 1e-5)[0].tolist(), flush=True)
