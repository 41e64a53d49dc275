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


# ---------------------------------------------------------------------------------------------------------------------
# Whole checkpoints in the reference's file format (Runner._save_checkpoint, runner.py:2799-2818): a torch.save'd dict
#   {'model_state_dict', ['bg_model_state_dict'], 'optimizers': {'nerf': Adam.state_dict(), ['bg_nerf': ...]}, 'iteration',
#    'dataset_index', ...}
# so that either side can resume the other's run (parameters AND Adam moments).  torch.optim.Adam addresses its state by the
# position of a parameter in module.parameters(): the registration order of the reference's modules, restated below
# (pinned by tests/test_checkpoint_cpu.py against the key order of the golden files, which were written in named_parameters() order).
# ---------------------------------------------------------------------------------------------------------------------
def _rank(key: str, dense: bool):
    """Sort key reproducing named_parameters() order of the reference's NeRFMoE (models/nerf_moe.py:250-316: the `layers`
    ModuleDict is filled 0 (experts: weights, then bias; then the gate), 1, 2, xyz, sigma, color, moe_external_gate,
    gate_input_norm; embedding_a is registered last) and NeRF (models/nerf.py:88-138: xyz_encodings, embedding_a,
    xyz_encoding_final, dir_a_encoding, sigma, rgb)."""
    k = key[len("module."):] if key.startswith("module.") else key
    wb = 0 if k.endswith("weight") else 1
    if dense:
        m = re.match(r"^xyz_encodings\.(\d+)\.0\.(weight|bias)$", k)
        if m:
            return (0, int(m.group(1)), wb)
        for i, pre in enumerate(("embedding_a.", "xyz_encoding_final.", "dir_a_encoding.0.", "sigma.", "rgb.")):
            if k.startswith(pre):
                return (1 + i, 0, wb)
        raise KeyError(f"unknown parameter key {key}")
    m = re.match(r"^layers\.0\.experts\.0\.(weights|bias)\.(\d+)$", k)
    if m:
        return (0, 0 if m.group(1) == "weights" else 1, int(m.group(2)))
    if k.startswith("layers.0.gates."):
        return (0, 2, 0)
    m = re.match(r"^layers\.(\w+)\.fcs\.(\d+)\.(weight|bias)$", k)
    if m:
        blk = {"1": 1, "2": 2, "xyz": 3, "sigma": 4, "color": 5, "moe_external_gate": 6}[m.group(1)]
        return (blk, int(m.group(2)), wb)
    if k.startswith("layers.gate_input_norm."):
        return (7, 0, wb)
    if k == "embedding_a.weight":
        return (8, 0, 0)
    if k.startswith("embedding_xyz."):          # hash-grid table (no reference counterpart): after everything else
        return (9, 0, 0)
    raise KeyError(f"unknown parameter key {key}")


def param_order(keys):
    """The keys in the order of the reference module's parameters()."""
    keys = list(keys)
    dense = any(k.replace("module.", "").startswith("xyz_encodings.") for k in keys)
    return sorted(keys, key=lambda k: _rank(k, dense))


def adam_state_dict(model) -> dict:
    """torch.optim.Adam.state_dict() of the reference's optimizer over model.parameters() (runner.py:485-488) holding this
    model's moments."""
    p = model._to_ref_layout(model.p)
    m = model._to_ref_layout(model._views(model.m))
    v = model._to_ref_layout(model._views(model.v))
    order = param_order(p.keys())
    state = {}
    if model.step_count > 0:
        for i, k in enumerate(order):
            state[i] = {"step": torch.tensor(float(model.step_count)), "exp_avg": m[k].detach().cpu(), "exp_avg_sq": v[k].detach().cpu()}
    # 'initial_lr' = the undecayed base rate: the reference rebuilds ExponentialLR(optimizer, last_epoch=iteration - 1) on resume
    # (runner.py:505-510), which raises KeyError without it; 'lr' = the rate of the current iteration
    group = dict(lr=model.lr, initial_lr=getattr(model, "base_lr", model.lr), betas=(0.9, 0.999), eps=1e-8, weight_decay=0,
                 amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                 params=list(range(len(order))))
    return {"state": state, "param_groups": [group]}


def load_adam_state_dict(model, osd: dict) -> None:
    order = param_order(model._to_ref_layout(model.p).keys())
    st = osd.get("state", {})
    grp = osd["param_groups"][0] if osd.get("param_groups") else {}
    model.lr = float(grp.get("lr", model.lr))
    model.base_lr = float(grp.get("initial_lr", getattr(model, "base_lr", model.lr)))
    if not st:
        model.m.zero_(); model.v.zero_(); model.step_count = 0
        return
    m = {k: st[i]["exp_avg"] for i, k in enumerate(order)}
    v = {k: st[i]["exp_avg_sq"] for i, k in enumerate(order)}
    model._load_ref_layout(m, model._views(model.m))
    model._load_ref_layout(v, model._views(model.v))
    model.step_count = int(float(st[0]["step"]))


def exponential_lr(base_lr: float, iteration: int, lr_decay_factor: float = 0.1, train_iterations: int = 500000) -> float:
    """The rate torch's ExponentialLR(gamma = lr_decay_factor ** (1 / train_iterations)) holds after `iteration` scheduler
    steps (runner.py:505-512): base_lr * lr_decay_factor ** (iteration / train_iterations).  SwitchNeRF.set_iteration applies it."""
    return float(base_lr) * float(lr_decay_factor) ** (float(iteration) / float(train_iterations))


def save_checkpoint(path, nerf, bg_nerf=None, iteration: int = 0, dataset_index: int = 0, module_prefix: bool = True) -> dict:
    """Writes (and returns) the reference's checkpoint dict; module_prefix reproduces the DDP-wrapped key names."""
    pre = "module." if module_prefix else ""
    if getattr(nerf, "ep", None) is not None:      # expert-parallel run: a rank holds trained weights only for the experts it owns;
        nerf.gather_expert_shards()                # collective - every rank must call save_checkpoint (any rank may then write the file)
    scaler = getattr(nerf, "loss_scaler", None)
    ck = {"model_state_dict": {pre + k: v.detach().cpu() for k, v in nerf.state_dict().items()},
          "optimizers": {"nerf": adam_state_dict(nerf)}, "iteration": int(iteration), "dataset_index": int(dataset_index),
          "scaler": scaler.state_dict() if scaler is not None else {}, "torch_random_state": torch.get_rng_state()}
    if bg_nerf is not None:
        ck["bg_model_state_dict"] = {pre + k: v.detach().cpu() for k, v in bg_nerf.state_dict().items()}
        ck["optimizers"]["bg_nerf"] = adam_state_dict(bg_nerf)
    if path is not None:
        torch.save(ck, path)
    return ck


def load_checkpoint(path_or_dict, nerf, bg_nerf=None) -> int:
    """Restores parameters and Adam moments from a checkpoint written by either side; returns its iteration."""
    # weights_only=False: the reference's Runner._save_checkpoint (runner.py:2799-2818) also pickles numpy / python RNG states
    # ('np_random_state', 'random_state'), which torch >= 2.6's default weights_only=True refuses; these are the user's own files
    ck = path_or_dict if isinstance(path_or_dict, dict) else torch.load(path_or_dict, map_location="cpu", weights_only=False)
    nerf.load_state_dict(ck["model_state_dict"])
    if "optimizers" in ck and "nerf" in ck["optimizers"]:
        load_adam_state_dict(nerf, ck["optimizers"]["nerf"])
    if getattr(nerf, "loss_scaler", None) is not None and ck.get("scaler"):     # GradScaler state (runner.py:2806): the scale survives a resume
        nerf.loss_scaler.load_state_dict(ck["scaler"])
        if getattr(nerf, "_ls_dev", None) is not None:
            nerf._loss_scale_tensor()
    if bg_nerf is not None:
        bg_nerf.load_state_dict(ck["bg_model_state_dict"])
        if "optimizers" in ck and "bg_nerf" in ck["optimizers"]:
            load_adam_state_dict(bg_nerf, ck["optimizers"]["bg_nerf"])
    return int(ck.get("iteration", 0))
