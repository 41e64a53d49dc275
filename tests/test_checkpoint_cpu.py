"""Host logic of switch_nerf_amd.checkpoint that needs no GPU: the parameter order torch.optim.Adam's state is addressed by
must equal the reference modules' named_parameters() order - recovered from the golden files, whose gsum__<name> entries
were written while iterating nerf.named_parameters() of the imported reference (oracle/gen_golden.py)."""
import os

import numpy as np

from switch_nerf_amd import checkpoint

G = os.path.join(os.path.dirname(__file__), "golden")


def _golden_order(name, prefix=""):
    g = np.load(os.path.join(G, name))
    tag = "gsum__" + prefix
    return [k[len(tag):] for k in g.files if k.startswith(tag) and (prefix or not k.startswith("gsum__bg__"))]


def test_param_order_matches_the_reference_modules():
    moe = _golden_order("render_train_unbalanced.npz")
    assert len(moe) == 32
    shuffled = sorted(moe, key=lambda k: hash(k) % 977)
    assert checkpoint.param_order(shuffled) == moe
    assert checkpoint.param_order(["module." + k for k in shuffled]) == ["module." + k for k in moe]
    dense = _golden_order("dense_nerf_train.npz")
    assert len(dense) == 25
    assert checkpoint.param_order(sorted(dense)) == dense
    bg = _golden_order("bg_train_coarse.npz", "bg__")
    assert checkpoint.param_order(sorted(bg)) == bg and bg == dense
