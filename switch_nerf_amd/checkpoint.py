"""Checkpoint key layouts of the reference.

The reference trains with `moe_expert_type: expertmlp` - one stacked tensor per expert layer,
`layers.0.experts.0.weights.{l}` [E, in, out] and `layers.0.experts.0.bias.{l}` [E, 1, out]
(/root/reference/switch_nerf/modules/tutel_moe_ext/tutel_moe_layer_nobatch.py:853-870) - saves `model_state_dict` from a
DDP-wrapped module (keys prefixed `module.`, runner.py:2799-2818), and evaluates with per-expert modules after
`convert_to_seqexperts` (models/model_utils.py:12-28): `layers.0.experts.0.experts.{e}.layers.{l}.weight` [out, in] (the
transpose of the stacked slice) and `.bias` [out].  SwitchNeRF.load_state_dict accepts any of these through `to_expertmlp`.
Pure tensor shuffling: no GPU, no library.
"""
from __future__ import annotations

import re
from typing import Dict

import numpy as np
import torch

_SEQ = re.compile(r"^(?P<pre>.*layers\.(?P<moe>\d+)\.experts\.0\.)experts\.(?P<e>\d+)\.layers\.(?P<l>\d+)\.(?P<kind>weight|bias)$")
_MLP = re.compile(r"^(?P<pre>.*layers\.(?P<moe>\d+)\.experts\.0\.)(?P<kind>weights|bias)\.(?P<l>\d+)$")


def _t(v):
    return v if torch.is_tensor(v) else torch.from_numpy(np.asarray(v))


def strip_module_prefix(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """consume_prefix_in_state_dict_if_present(sd, 'module.') (model_utils.py:147)."""
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}


def to_seqexperts(sd: Dict[str, torch.Tensor], prefix: str = "") -> Dict[str, torch.Tensor]:
    """expertmlp layout -> per-expert modules, exactly the reference's convert_to_seqexperts (which also writes the
    `module.` prefix: pass prefix='module.' to reproduce its keys)."""
    out = {}
    for k, v in sd.items():
        m = _MLP.match(k)
        if not m:
            out[k] = _t(v)
            continue
        v = _t(v)
        base = m.group("pre")
        base = base[len("module."):] if base.startswith("module.") else base
        for e in range(v.shape[0]):
            key = f"{prefix}{base}experts.{e}.layers.{m.group('l')}."
            if m.group("kind") == "weights":
                out[key + "weight"] = v[e].t().contiguous()
            else:
                out[key + "bias"] = v[e].reshape(-1)
    return out


def to_expertmlp(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Any of the reference's layouts (with or without the DDP prefix) -> the expertmlp layout SwitchNeRF stores."""
    sd = strip_module_prefix(sd)
    out, groups = {}, {}
    for k, v in sd.items():
        m = _SEQ.match(k)
        if not m:
            out[k] = _t(v)
            continue
        groups.setdefault((m.group("pre"), m.group("kind"), int(m.group("l"))), {})[int(m.group("e"))] = _t(v)
    for (pre, kind, l), per in groups.items():
        E = max(per) + 1
        assert sorted(per) == list(range(E)), f"missing experts for {pre}{kind}.{l}"
        if kind == "weight":
            out[f"{pre}weights.{l}"] = torch.stack([per[e].t() for e in range(E)], 0).contiguous()     # [E, in, out]
        else:
            out[f"{pre}bias.{l}"] = torch.stack([per[e].reshape(1, -1) for e in range(E)], 0).contiguous()  # [E, 1, out]
    return out
