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


class _StubModel:
    """The attributes checkpoint.py touches, over three reference-named parameters on the CPU (the real models need the GPU)."""
    def __init__(self):
        import torch
        g = torch.Generator().manual_seed(3)
        self.keys = ["layers.xyz.fcs.0.weight", "layers.xyz.fcs.0.bias", "embedding_a.weight"]
        shapes = [(4, 3), (4,), (2, 5)]
        self.p = {k: torch.randn(s, generator=g) for k, s in zip(self.keys, shapes)}
        self.m = {k: torch.randn(s, generator=g) for k, s in zip(self.keys, shapes)}
        self.v = {k: torch.rand(s, generator=g) for k, s in zip(self.keys, shapes)}
        self.step_count, self.lr, self.base_lr = 7, 4e-4, 5e-4

    def _to_ref_layout(self, d):
        return {k: v.clone() for k, v in d.items()}

    def _views(self, flat):
        return flat

    def _load_ref_layout(self, sd, dst):
        for k in self.keys:
            dst[k].copy_(sd[k])

    def state_dict(self):
        return self._to_ref_layout(self.p)

    def load_state_dict(self, sd):
        sd = {k.replace("module.", ""): v for k, v in sd.items()}
        self._load_ref_layout(sd, self.p)


def test_load_checkpoint_written_by_the_reference_runner(tmp_path):
    """Runner._save_checkpoint (runner.py:2799-2818) also pickles numpy / python RNG states, a GradScaler state and a dataset
    state: the file must load (torch >= 2.6 defaults to weights_only=True, which refuses them)."""
    import random
    import torch
    m = _StubModel()
    ck = checkpoint.save_checkpoint(None, m, iteration=12, dataset_index=3)
    ck.update(np_random_state=np.random.get_state(), random_state=random.getstate(), scaler={"scale": 65536.0, "_growth_tracker": 0},
              dataset_state="chunk-000123")
    path = tmp_path / "12.pt"
    torch.save(ck, path)
    m2 = _StubModel()
    for k in m2.keys:
        m2.p[k].zero_(); m2.m[k].zero_(); m2.v[k].zero_()
    m2.step_count, m2.lr, m2.base_lr = 0, 1.0, 1.0
    assert checkpoint.load_checkpoint(str(path), m2) == 12
    for k in m.keys:
        assert torch.equal(m2.p[k], m.p[k]) and torch.equal(m2.m[k], m.m[k]) and torch.equal(m2.v[k], m.v[k])
    assert m2.step_count == 7 and m2.lr == 4e-4 and m2.base_lr == 5e-4


def test_adam_state_resumes_under_the_reference_scheduler():
    """The reference rebuilds ExponentialLR(optimizer, last_epoch=iteration - 1) after optimizer.load_state_dict
    (runner.py:496-512); that needs 'initial_lr' in the param group."""
    import torch
    from torch.optim.lr_scheduler import ExponentialLR
    m = _StubModel()
    it, total, decay = 40, 1000, 0.1
    m.lr = checkpoint.exponential_lr(m.base_lr, it - 1, decay, total)            # the rate the scheduler left behind
    osd = checkpoint.adam_state_dict(m)
    assert osd["param_groups"][0]["initial_lr"] == 5e-4 and osd["param_groups"][0]["lr"] == m.lr
    order = checkpoint.param_order(m.keys)
    params = [torch.nn.Parameter(m.p[k].clone()) for k in order]
    opt = torch.optim.Adam(params, lr=5e-4)
    opt.load_state_dict(osd)
    sch = ExponentialLR(opt, gamma=decay ** (1 / total), last_epoch=it - 1)       # raised KeyError('initial_lr') before
    want = checkpoint.exponential_lr(5e-4, it, decay, total)
    assert abs(sch.get_last_lr()[0] - want) <= 5e-3 * want      # (whether the constructor takes one decay step varies with the torch version)
    assert torch.equal(opt.state[params[0]]["exp_avg"], m.m[order[0]])


def test_loss_scaler_follows_gradscaler_schedule():
    """model.LossScaler against torch's GradScaler bookkeeping (runner.py:483, 679-693): same scale after the same sequence of clean /
    overflowing steps (GradScaler itself needs a CUDA device for its tensors - the schedule is restated from its documented contract:
    backoff on inf, growth after growth_interval consecutive clean steps, the streak restarts after either)."""
    from switch_nerf_amd.model import LossScaler
    s = LossScaler(init_scale=1024.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=3)
    seq = [False, False, True, False, False, False, False, True, True, False, False, False]
    scale, good, want = 1024.0, 0, []
    for inf in seq:
        if inf:
            scale, good = scale * 0.5, 0
        else:
            good += 1
            if good == 3:
                scale, good = scale * 2.0, 0
        want.append(scale)
    got = []
    for inf in seq:
        s.update(inf)
        got.append(s.scale)
    assert got == want and s.skipped == 3
    sd = s.state_dict()
    assert sd["scale"] == want[-1] and sd["growth_interval"] == 3 and sd["_growth_tracker"] == 0
    d = LossScaler()
    assert (d.scale, d.growth_factor, d.backoff_factor, d.growth_interval) == (65536.0, 2.0, 0.5, 2000)      # torch's defaults
