#!/usr/bin/env python
"""Golden vectors for the cat-mod PRODUCER layer and the cat-mod train-step assembly, from the
GENUINE reference imported in this container (tests/golden/make_golden.py builds it in /tmp).

    python tests/golden/make_golden_catmod_layer.py

Pins SURVEY 8 rows a11 (`GlobalNormFlipFlopCatMod.forward`, taiyaki/layers.py:1616-1640, output
alphabet order AYCZGT, `can_mods_offsets` 1495-1497) and, for cat-mod, a17 (`calculate_loss`,
bin/train_flipflop.py:161-182): the reference layer gets its weights through `load_state_dict`
from a counter-based array (no RNG involved), runs forward, and its output goes through the
reference's `cat_mod_flipflop_loss` + `flipflop_logpartition / nblk`; the loss is back-propagated
to the layer's input and weights.  Stored: inputs, weights, outputs, lossvector, gradients.
The fixture is data only.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import make_golden  # noqa: E402


def main():
    make_golden.build_reference()
    import torch
    from taiyaki import alphabet, ctc, layers
    from taiyaki_amd import synth

    torch.set_num_threads(4)
    out = {}
    for name, (T, N, insize, seed, sharp) in dict(t9n3=(9, 3, 12, 5, 1.0), t60n4=(60, 4, 16, 6, 1.0),
                                                  t40n2_sharp=(40, 2, 8, 7, 2.0)).items():
        ai = alphabet.AlphabetInfo("ACGTZY", "ACGTCA", ["5mC", "6mA"], do_reorder=False)
        lay = layers.GlobalNormFlipFlopCatMod(insize, ai)
        W = (synth.uniform01(seed, 21, lay.size * insize).reshape(lay.size, insize) - 0.5).astype(np.float32)
        b = (synth.uniform01(seed, 22, lay.size) - 0.5).astype(np.float32)
        lay.load_state_dict({"linear.weight": torch.tensor(W), "linear.bias": torch.tensor(b)})
        x = torch.tensor((4 * synth.uniform01(seed, 23, T * N * insize) - 2).reshape(T, N, insize).astype(np.float32),
                         requires_grad=True)
        y = lay(x)
        # labels: alphabet order ACGTZY -> canonical base + modification category
        seqlens = np.array([max(1, (T * (n + 2)) // (2 * N + 3)) for n in range(N)], dtype=np.int32)
        labels = synth.randint(seed, 24, int(seqlens.sum()), 6)
        can = lay.can_labels[labels]
        mods = lay.mod_labels[labels]
        off = np.concatenate([[0], np.cumsum(seqlens)])
        seqs = np.concatenate([synth.flipflop_code(can[off[n]:off[n + 1]]) for n in range(N)])
        mcw = np.full(6, 8.0, dtype=np.float32)             # ones * mod_factor (train_flipflop.py:167-170)
        lossvector = ctc.cat_mod_flipflop_loss(y, torch.tensor(seqs), torch.tensor(seqlens), torch.tensor(mods),
                                               lay.can_mods_offsets, mcw, sharp)
        ntrans = y.shape[2] - int(lay.can_mods_offsets[-1])
        lossvector = lossvector + layers.flipflop_logpartition(y[:, :, :ntrans]) / float(T)
        lossvector.mean().backward()
        out.update({name + "/W": W, name + "/b": b, name + "/x": x.detach().numpy(), name + "/y": y.detach().numpy(),
                    name + "/seqs": seqs.astype(np.int32), name + "/seqlens": seqlens,
                    name + "/mod_cats": mods.astype(np.int32), name + "/mod_cat_weights": mcw,
                    name + "/sharp": np.float32(sharp), name + "/lossvector": lossvector.detach().numpy(),
                    name + "/dx": x.grad.numpy(), name + "/dW": lay.linear.weight.grad.numpy(),
                    name + "/db": lay.linear.bias.grad.numpy(),
                    name + "/can_mods_offsets": lay.can_mods_offsets, name + "/can_nmods": lay.can_nmods,
                    name + "/can_labels": lay.can_labels, name + "/mod_labels": lay.mod_labels,
                    name + "/output_alphabet": np.array(lay.output_alphabet)})
        print(name, "y", tuple(y.shape), "alphabet", lay.output_alphabet, "offsets", lay.can_mods_offsets,
              "lossvector", lossvector.detach().numpy()[:3])
    np.savez_compressed(os.path.join(HERE, "catmod_layer.npz"), **out)


if __name__ == "__main__":
    main()
