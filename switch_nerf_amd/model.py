"""Host-side model of the Switch-NeRF train hot path on top of libswn_hip.so.

Mirrors, for the building.yaml wiring (NeRFMoE, external gate + LayerNorm, top-1 expertmlp, batched capacity path):
  * models/nerf_moe.py:103-455  NeRFMoE            -> SwitchNeRF (parameters, state_dict layout, forward)
  * rendering.py:15-494         render_rays        -> SwitchNeRF.render / rendering.render_rays
  * runner.py:1077-1123, 646-686 training step      -> SwitchNeRF.train_step (loss, backward, Adam)
(paths relative to /root/reference/switch_nerf/).  All arithmetic on points runs in HIP kernels through the C ABI;
torch only owns the device buffers, the stream, and a handful of per-ray (N_rays x 75) tensors.

MI355X-first design notes
  * the whole ray batch (e.g. 8192 x 256 = 2M points, ~30 GB of bf16 activations) is processed in ONE pass: the
    reference's model chunks (rendering.py:354) survive only as routing *segments* (capacity / ranking / l_aux are
    per segment), so every kernel is launched once per step over all segments - 288 GB of HBM makes chunking for
    memory unnecessary;
  * dispatch and the backward scatter never materialise: the expert chain gathers its input rows through the
    routing permutation, and the front backward gathers the expert input-gradient rows through tok2row;
  * the direction / appearance part of layer "2" is constant along a ray: it is folded into a per-ray bias
    (N_rays x 128) instead of being concatenated to every point (331 -> 256 input features on the hot GEMM);
  * master weights are fp32 in [in, out] layout in one flat buffer (one Adam launch, one gradient all-reduce);
    compute copies (bf16 or fp32) in [out, in] (forward) and [in, out] (backward-data) are refreshed after Adam.
"""
from __future__ import annotations

import contextlib
import os
import math
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib, ops

BUILDING = dict(model_dim=256, num_experts=8, expert_layers=7, skips=(3,), pos_xyz_dim=12, pos_dir_dim=4,
                appearance_dim=48, appearance_count=10, gate_hidden=256, gate_layers=2, layer2_out=128)


def _ceil_to(x, m):
    return (x + m - 1) // m * m


_EXP_RECORD_STREAM = os.environ.get("SWN_EXP_RECORD_STREAM") == "1"      # see SwitchNeRF._join_side_outputs (read once, experiment only)


def resolve_kernel_switches(env=None) -> dict:
    """The kernel-selection switches of a model, resolved ONCE (SwitchNeRF.__init__ reads the process environment through this; no
    forward / backward looks at os.environ).  Defaults = the shipped kernel set; the SWN_* variables are experiment / test knobs:
      front_geom (SWN_FRONT_GEOM, 7)      dense front chains: 7 / 6 = the persistent 256-row geometry, 1 = the 64-row kernels
      chain_geom (SWN_CHAIN_GEOM, 7)      expert chains: include/swn.h swn_chain_desc.geometry (1, 2, 4, 5, 6, 7)
      tail_geom (SWN_TAIL_GEOM, 1)        the separate tail backward chain: 6 / 7 = persistent geometry (measured slower: off)
      fused_tail (SWN_FUSED_TAIL, on)     dense tail + heads inside the expert forward launch (chain_big.hip tag 7)
      fused_tail_bwd (SWN_FUSED_TAIL_BWD, on)   tail backward + combine backward in front of the expert backward chain (tag 8)
      fused_dwsig (SWN_FUSED_DWSIG, on)   the sigma head's weight gradient from tag 8's combine pass
      fused_heads (SWN_NO_FUSED_HEADS unset)    sigma / colour heads inside the tail forward chain (swn.h: heads_raw)
      overlap (SWN_NO_OVERLAP != 1)       side-stream overlap of the expert weight gradients and the per-ray launches"""
    e = os.environ if env is None else env
    return dict(front_geom=int(e.get("SWN_FRONT_GEOM", "7")), chain_geom=int(e.get("SWN_CHAIN_GEOM", "7")),
                tail_geom=int(e.get("SWN_TAIL_GEOM", "1")), fused_tail=e.get("SWN_FUSED_TAIL", "1") != "0",
                fused_tail_bwd=e.get("SWN_FUSED_TAIL_BWD", "1") != "0", fused_dwsig=e.get("SWN_FUSED_DWSIG", "1") != "0",
                fused_heads=e.get("SWN_NO_FUSED_HEADS") is None, overlap=e.get("SWN_NO_OVERLAP", "0") != "1")


def c_esz(dtype) -> int:
    return 4 if dtype == torch.float32 else 2

class LossScaler:
    """torch.cuda.amp.GradScaler's contract (the reference trains fp16 with it: runner.py:483 `GradScaler(enabled=hparams.amp)`,
    :679 `scaler.scale(loss).backward()`, scaler.step / update): the loss gradient is multiplied by `scale`, the optimizer step
    is skipped when a gradient is not finite (scale *= backoff_factor), and after growth_interval clean steps scale *= growth_factor.
    Same defaults as torch."""

    def __init__(self, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        self.scale, self.growth_factor, self.backoff_factor, self.growth_interval = float(init_scale), growth_factor, backoff_factor, growth_interval
        self._good = 0
        self.skipped = 0

    def update(self, found_inf: bool):
        if found_inf:
            self.scale *= self.backoff_factor
            self._good = 0
            self.skipped += 1
        else:
            self._good += 1
            if self._good >= self.growth_interval:
                self.scale *= self.growth_factor
                self._good = 0

    def state_dict(self):
        return {"scale": self.scale, "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": self._good}

    def load_state_dict(self, sd):
        """GradScaler.load_state_dict's keys (the reference saves scaler.state_dict() under 'scaler', runner.py:2806)."""
        if not sd:
            return
        self.scale = float(sd.get("scale", self.scale))
        self.growth_factor = float(sd.get("growth_factor", self.growth_factor))
        self.backoff_factor = float(sd.get("backoff_factor", self.backoff_factor))
        self.growth_interval = int(sd.get("growth_interval", self.growth_interval))
        self._good = int(sd.get("_growth_tracker", 0))


class SwitchNeRF:
    def __init__(self, cfg: dict = BUILDING, dtype=torch.bfloat16, device="cuda", capacity_factor=1.0,
                 batch_prioritized=True, moe_l_aux_wt=5e-4, lr=5e-4, seed=0, gate_noise=-1.0, kernel_switches=None):
        self.cfg, self.dtype, self.dev = dict(cfg), dtype, torch.device(device)
        # the kernel selection is resolved HERE, once (environment knobs included); set_kernel_switches() changes it afterwards
        self.sw = resolve_kernel_switches()
        self._env_overrides = {k: v for k, v in sorted(os.environ.items()) if k.startswith("SWN_")}
        if kernel_switches:
            self.set_kernel_switches(**kernel_switches)
        # the 16-bit compute type selects the build of the library: bfloat16 (amp_use_bfloat16) or IEEE half (the reference's default
        # autocast dtype; BASELINE configs[4] "fp16 MFMA") - see _lib.use_half.  fp16 trains with loss scaling like the reference.
        if dtype == torch.float16:
            _lib.use_half("f16")
        elif dtype == torch.bfloat16:
            _lib.use_half("bf16")
        self.loss_scaler = LossScaler() if dtype == torch.float16 else None
        self.cf, self.bpr, self.wt, self.lr = capacity_factor, batch_prioritized, moe_l_aux_wt, lr
        # --gate_noise (opts.py:208, default -1 = off): a TRAINING forward adds gate_noise * randn / E to the router's logits
        # (tutel_moe_layer_nobatch.py:119-122).  gate_noise_draw: a [P, E] tensor to use as that draw (tests), else drawn per forward.
        self.gate_noise, self.gate_noise_draw = float(gate_noise), None
        self.base_lr = lr             # undecayed rate ('initial_lr' of the reference's ExponentialLR); self.lr = the current rate
        spec = self._configure(cfg)
        self.spec, off = {}, 0
        self.n_dense = None           # elements of the flat buffer before the first expert parameter (all of it for dense models)
        for i, (name, shape) in enumerate(spec):
            if i == getattr(self, "_n_dense_spec", len(spec)):
                self.n_dense = off
            n = int(np.prod(shape))
            self.spec[name] = (off, shape)
            off += _ceil_to(n, 64)
        self.n_flat = off
        if self.n_dense is None:
            self.n_dense = off
        z = lambda: torch.zeros(self.n_flat, dtype=torch.float32, device=self.dev)
        self.flat, self.grad, self.m, self.v = z(), z(), z(), z()
        self.p = {k: self.flat[o:o + int(np.prod(s))].view(s) for k, (o, s) in self.spec.items()}
        self.g = {k: self.grad[o:o + int(np.prod(s))].view(s) for k, (o, s) in self.spec.items()}
        self.step_count = 0
        # compute copies
        self.wf: Dict[str, torch.Tensor] = {}
        self.wb: Dict[str, torch.Tensor] = {}
        self._init_random(seed)
        self._bufs = {}
        self.profile = False          # bench.py: record HIP events around the major launches
        self.events: Dict[str, list] = {}
        # side HIP stream: the HBM-bound expert weight-gradient GEMMs overlap with the rest of the backward pass
        self.side = torch.cuda.Stream(device=self.dev) if self.dev.type == "cuda" else None
        self.overlap = self.sw["overlap"]      # side-stream overlap of the expert weight gradients
        self._kernel_sel = {}         # kernel_set(): what the last forward / backward selected
        self.ep = None                # parallel.ExpertParallel: experts sharded over ranks, tokens exchanged (set_expert_parallel)
        self.expert_wgrad_splits = 0         # row splits of the legacy expert weight-gradient launch (0 = heuristic)

    def _configure(self, cfg):
        """Sets the network dimensions and returns the flat parameter layout [(name, shape)]; Linear weights are [in, out]."""
        assert tuple(cfg["skips"]) == (3,) or len(cfg["skips"]) <= 1, "one skip connection supported"
        M, E, L, G, H2 = cfg["model_dim"], cfg["num_experts"], cfg["expert_layers"], cfg["gate_hidden"], cfg["layer2_out"]
        assert G == M, "external gate width must equal model_dim (building.yaml)"
        self.hash = cfg.get("hash")                  # multiresolution hash-grid input encoding (ops.hash_encode_fwd) or None
        self.in_xyz = 3 + 6 * cfg["pos_xyz_dim"] if self.hash is None else 2 * self.hash["n_levels"]
        self.in_dir = 3 + 6 * cfg["pos_dir_dim"]
        self.KP = _ceil_to(self.in_xyz, 64)          # padded PE width (chain K granularity)
        self.DP = _ceil_to(self.in_dir, 8)
        self.n_ray_feat = self.in_dir + cfg["appearance_dim"]
        spec = [("xyz.w", (self.KP, M)), ("xyz.b", (M,)), ("gate0.w", (M, G)), ("gate0.b", (G,)),
                ("gate1.w", (G, G)), ("gate1.b", (G,)), ("ln.w", (G,)), ("ln.b", (G,)), ("wg", (E, G))]
        spec += [("l1.w", (M, M)), ("l1.b", (M,)), ("l2h.w", (M, H2)), ("l2r.w", (self.n_ray_feat, H2)), ("l2.b", (H2,)),
                 ("sigma.w", (M,)), ("sigma.b", (1,)), ("color.w", (3, H2)), ("color.b", (3,)),
                 ("emb", (cfg["appearance_count"], cfg["appearance_dim"]))]
        # the expert parameters come LAST in the flat buffer: under expert parallelism only the dense prefix is all-reduced
        # (the reference keeps the experts out of DDP, models/nerf_moe.py:139, 1037-1039); hash table: see below
        self._expert_spec = []
        for l in range(L):
            self._expert_spec += [(f"exp{l}.w", (E, M, M)), (f"exp{l}.b", (E, M))]
        self.L, self.M, self.E, self.G, self.H2 = L, M, E, G, H2
        self._chain_weights = ["xyz", "gate0", "gate1", "l1", "l2h"] + [f"exp{l}" for l in range(L)]
        self._fwd_only_weights = {"xyz"}              # first layer: no input gradient, no transposed copy
        if self.hash is not None:                     # trainable encoding: the first layer's input gradient feeds the table
            spec.append(("hash.table", (self.hash["n_levels"], 1 << self.hash["log2_table"], 2)))
            self._fwd_only_weights = set()
        self._n_dense_spec = len(spec)
        return spec + self._expert_spec

    @contextlib.contextmanager
    def _timed(self, name):
        if not self.profile:
            yield
            return
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        yield
        b.record()
        self.events.setdefault(name, []).append((a, b))

    # ------------------------------------------------------------------------------------------ parameters
    def _init_random(self, seed):
        """torch.nn.Linear-style init (U(-1/sqrt(in), 1/sqrt(in))) drawn in the reference's state_dict layout."""
        import sys, os
        g = torch.Generator().manual_seed(seed)
        cfg = self.cfg

        def lin(out_f, in_f):
            b = 1.0 / math.sqrt(in_f)
            return ((torch.rand(out_f, in_f, generator=g) * 2 - 1) * b), ((torch.rand(out_f, generator=g) * 2 - 1) * b)
        sd = {}
        M, E, L, G, H2 = self.M, self.E, self.L, self.G, self.H2
        sd["layers.xyz.fcs.0.weight"], sd["layers.xyz.fcs.0.bias"] = lin(M, self.in_xyz)
        for l in range(L):
            w = torch.empty(E, M, M)
            b = torch.empty(E, 1, M)
            for e in range(E):
                wt, bt = lin(M, M)
                w[e], b[e, 0] = wt.t(), bt
            sd[f"layers.0.experts.0.weights.{l}"], sd[f"layers.0.experts.0.bias.{l}"] = w, b
        sd["layers.0.gates.0.wg.weight"] = lin(E, G)[0]
        sd["layers.1.fcs.0.weight"], sd["layers.1.fcs.0.bias"] = lin(M, M)
        sd["layers.2.fcs.0.weight"], sd["layers.2.fcs.0.bias"] = lin(H2, M + self.n_ray_feat)
        sd["layers.sigma.fcs.0.weight"], sd["layers.sigma.fcs.0.bias"] = lin(1, M)
        sd["layers.color.fcs.0.weight"], sd["layers.color.fcs.0.bias"] = lin(3, H2)
        sd["layers.moe_external_gate.fcs.0.weight"], sd["layers.moe_external_gate.fcs.0.bias"] = lin(G, M)
        sd["layers.moe_external_gate.fcs.1.weight"], sd["layers.moe_external_gate.fcs.1.bias"] = lin(G, G)
        sd["layers.gate_input_norm.weight"], sd["layers.gate_input_norm.bias"] = torch.ones(G), torch.zeros(G)
        sd["embedding_a.weight"] = torch.randn(cfg["appearance_count"], cfg["appearance_dim"], generator=g)
        if self.hash is not None:         # instant-NGP initialisation: U(-1e-4, 1e-4)
            sd["embedding_xyz.table"] = (torch.rand(self.spec["hash.table"][1], generator=g) * 2 - 1) * 1e-4
        self.load_state_dict(sd)

    def load_state_dict(self, sd):
        """Accepts the reference's NeRFMoE.state_dict() key layout - as saved (`module.` prefix of
        the DDP wrapper, expertmlp stacking) or after the reference's convert_to_seqexperts (per-expert modules,
        models/model_utils.py:12-28); values torch or numpy.  See checkpoint.py."""
        from . import checkpoint
        self._load_ref_layout(checkpoint.to_expertmlp(sd), self.p)
        self.refresh_compute_copies()

    def _views(self, flat):
        """name -> view dict over a flat buffer laid out like the parameters (gradients, Adam moments)."""
        return {k: flat[o:o + int(np.prod(s))].view(s) for k, (o, s) in self.spec.items()}

    def _load_ref_layout(self, sd, p):
        """Inverse of _to_ref_layout: tensors keyed like the reference's state_dict -> the flat-buffer views `p`."""
        def t(k):
            v = sd[k]
            v = torch.from_numpy(np.asarray(v)) if not torch.is_tensor(v) else v
            return v.detach().to(torch.float32).to(self.dev)
        M = self.M
        with torch.no_grad():
            p["xyz.w"].zero_()
            p["xyz.w"][: self.in_xyz] = t("layers.xyz.fcs.0.weight").t()
            p["xyz.b"].copy_(t("layers.xyz.fcs.0.bias"))
            p["gate0.w"].copy_(t("layers.moe_external_gate.fcs.0.weight").t())
            p["gate0.b"].copy_(t("layers.moe_external_gate.fcs.0.bias"))
            p["gate1.w"].copy_(t("layers.moe_external_gate.fcs.1.weight").t())
            p["gate1.b"].copy_(t("layers.moe_external_gate.fcs.1.bias"))
            p["ln.w"].copy_(t("layers.gate_input_norm.weight"))
            p["ln.b"].copy_(t("layers.gate_input_norm.bias"))
            p["wg"].copy_(t("layers.0.gates.0.wg.weight"))
            for l in range(self.L):
                p[f"exp{l}.w"].copy_(t(f"layers.0.experts.0.weights.{l}"))
                p[f"exp{l}.b"].copy_(t(f"layers.0.experts.0.bias.{l}").view(self.E, M))
            p["l1.w"].copy_(t("layers.1.fcs.0.weight").t())
            p["l1.b"].copy_(t("layers.1.fcs.0.bias"))
            w2 = t("layers.2.fcs.0.weight")
            p["l2h.w"].copy_(w2[:, :M].t())
            p["l2r.w"].copy_(w2[:, M:].t())
            p["l2.b"].copy_(t("layers.2.fcs.0.bias"))
            p["sigma.w"].copy_(t("layers.sigma.fcs.0.weight").view(-1))
            p["sigma.b"].copy_(t("layers.sigma.fcs.0.bias"))
            p["color.w"].copy_(t("layers.color.fcs.0.weight"))
            p["color.b"].copy_(t("layers.color.fcs.0.bias"))
            p["emb"].copy_(t("embedding_a.weight"))
            if self.hash is not None:
                p["hash.table"].copy_(t("embedding_xyz.table"))

    def _to_ref_layout(self, d):
        M = self.M
        out = {}
        out["layers.xyz.fcs.0.weight"] = d["xyz.w"][: self.in_xyz].t().contiguous()
        out["layers.xyz.fcs.0.bias"] = d["xyz.b"].clone()
        out["layers.moe_external_gate.fcs.0.weight"] = d["gate0.w"].t().contiguous()
        out["layers.moe_external_gate.fcs.0.bias"] = d["gate0.b"].clone()
        out["layers.moe_external_gate.fcs.1.weight"] = d["gate1.w"].t().contiguous()
        out["layers.moe_external_gate.fcs.1.bias"] = d["gate1.b"].clone()
        out["layers.gate_input_norm.weight"] = d["ln.w"].clone()
        out["layers.gate_input_norm.bias"] = d["ln.b"].clone()
        out["layers.0.gates.0.wg.weight"] = d["wg"].clone()
        for l in range(self.L):
            out[f"layers.0.experts.0.weights.{l}"] = d[f"exp{l}.w"].clone()
            out[f"layers.0.experts.0.bias.{l}"] = d[f"exp{l}.b"].view(self.E, 1, M).clone()
        out["layers.1.fcs.0.weight"] = d["l1.w"].t().contiguous()
        out["layers.1.fcs.0.bias"] = d["l1.b"].clone()
        out["layers.2.fcs.0.weight"] = torch.cat([d["l2h.w"].t(), d["l2r.w"].t()], 1).contiguous()
        out["layers.2.fcs.0.bias"] = d["l2.b"].clone()
        out["layers.sigma.fcs.0.weight"] = d["sigma.w"].view(1, -1).clone()
        out["layers.sigma.fcs.0.bias"] = d["sigma.b"].clone()
        out["layers.color.fcs.0.weight"] = d["color.w"].clone()
        out["layers.color.fcs.0.bias"] = d["color.b"].clone()
        out["embedding_a.weight"] = d["emb"].clone()
        if self.hash is not None:
            out["embedding_xyz.table"] = d["hash.table"].clone()
        return out

    def state_dict(self, layout="expertmlp", prefix=""):
        """Parameters in the reference's key layout (so checkpoints are interchangeable): layout "expertmlp" (what the
        reference trains and saves) or "seqexperts" (what its evaluation loads after convert_to_seqexperts)."""
        sd = self._to_ref_layout(self.p)
        if layout == "seqexperts":
            from . import checkpoint
            return checkpoint.to_seqexperts(sd, prefix)
        assert layout == "expertmlp"
        return {prefix + k: v for k, v in sd.items()}

    def grad_dict(self):
        """Gradients in the reference's key layout (for parity tests)."""
        return self._to_ref_layout(self.g)

    def refresh_compute_copies(self):
        """fp32 master [in, out] -> MFMA-fragment-major compute copies: forward (N=out, K=in), backward (N=in, K=out)."""
        pairs = []
        for n in self._chain_weights:
            w = self.p[n + ".w"]
            w3 = w if w.dim() == 3 else w.unsqueeze(0)
            if n not in self.wf:
                self.wf[n] = ops.pack_weights(w3, self.dtype, True)
                if n not in self._fwd_only_weights:
                    self.wb[n] = ops.pack_weights(w3, self.dtype, False)
            else:
                pairs.append((w3, self.wf[n], True))
                if n in self.wb:
                    pairs.append((w3, self.wb[n], False))
        # the front chains on the persistent 256-row geometry (chain_big.hip, geometry 7) run every layer with K = 256: the first layer's
        # 128-row weight (75 -> 256, PE padded to 128) is packed zero-padded to 256 rows
        if self._front_big():
            w3 = self.p["xyz.w"].unsqueeze(0)
            if "xyz_pad" not in self.wf:
                self.wf["xyz_pad"] = ops.pack_weights_padded(w3, self.dtype, True, 256)
            else:
                pairs.append((w3, self.wf["xyz_pad"], True, 256))
        if self._tail_big() or self._tail_fused() or "l2h_pad" in self.wb:      # the tail BACKWARD layers likewise: dh2 (128 features) under the backward-data copy of layer "2" padded in K
            w3 = self.p["l2h.w"].unsqueeze(0)
            if "l2h_pad" not in self.wb:
                self.wb["l2h_pad"] = ops.pack_weights_padded(w3, self.dtype, False, 0, 256)
            else:
                pairs.append((w3, self.wb["l2h_pad"], False, 0, 256))
        if self._tail_fused() or "l2h_pad" in self.wf:    # the tail folded into the expert forward chain: layer "2" (256 -> 128) zero-padded to 256 outputs
            w3 = self.p["l2h.w"].unsqueeze(0)
            if "l2h_pad" not in self.wf:
                self.wf["l2h_pad"] = ops.pack_weights_padded(w3, self.dtype, True, 0, 256)
            else:
                pairs.append((w3, self.wf["l2h_pad"], True, 0, 256))
        if pairs:
            ops.repack_weights_batched(pairs)      # one launch (was 23 of ~5 us each: a tenth of the step at 1024 rays per GPU)
        if self._flat_param is not None:           # the copies now match the master weights as of this version of flat_param
            self._packed_version = self._flat_param._version

    def kernel_set(self) -> dict:
        """The kernel selection the LAST forward / backward of this model resolved to (what the SWN_* environment switches and the
        shapes added up to): expert chain geometry (7 = persistent phase-shifted 256-row tiles, accumulators start at the bias), front
        chain geometry, the dense tail inside the expert forward launch (tag 7), its backward + combine backward in front of the expert
        backward chain (tag 8), the sigma head's weight gradient from that launch's combine pass, side-stream overlap of the expert
        weight gradients, expert parallelism, and every SWN_* variable that is set.  bench.py prints it in its line's `config`;
        tests/test_fullsize_gpu.py pins the default set."""
        sel = dict(self._kernel_sel)
        sel["wgrad_overlap"] = bool(self.overlap)
        sel["expert_parallel"] = int(self.ep.world) if self.ep is not None else 0
        sel["env_overrides"] = dict(self._env_overrides)          # (the SWN_* variables that were set when the model was built)
        return sel

    def set_kernel_switches(self, **kw):
        """Change kernel-selection switches of this model (names: resolve_kernel_switches); returns the previous values of the ones
        changed.  Takes effect with the next forward; compute copies a newly selected kernel needs are packed on demand."""
        prev = {}
        for k, v in kw.items():
            if k not in self.sw:
                raise KeyError(f"unknown kernel switch {k!r} (known: {sorted(self.sw)})")
            prev[k] = self.sw[k]
            self.sw[k] = type(self.sw[k])(v)
        if "overlap" in kw:
            self.overlap = self.sw["overlap"]
        return prev

    DEFAULT_KERNEL_SET = dict(geom=7, front_geom=7, tail_fused=True, fused_backward=True, comb_dwsig=True, wgrad_overlap=True)

    def _front_big(self) -> bool:
        """The dense front chains (PE -> xyz -> gate MLP, and their backward) on the persistent 256-row geometry: 256-feature layers over
        a 128-feature encoding in a 16-bit compute dtype (building.yaml).  SWN_FRONT_GEOM=1 keeps them on the 64-row kernels."""
        return ("xyz.w" in self.spec and self.M == 256 and self.G == 256 and self.KP == 128 and self.dtype != torch.float32
                and self.hash is None and self.sw["front_geom"] != 1)

    def _tail_big(self) -> bool:
        """The tail backward chain (dh2 -> dh1 -> dy with the combine backward in its write-out) on the persistent 256-row geometry:
        built and bit-exact (tests/test_kernels_gpu.py::test_chain_fused_combine_backward), but OFF by default - measured 0.23 ms per
        step SLOWER than the 64-row kernel (14.42 against 14.19 ms on one box): a two-layer chain is 5 phases, and the staging phase
        with the combine backward in it (three more operand streams behind the write-out) is the longest of them.  SWN_TAIL_GEOM=7
        turns it on (profiles/r04_experiments.md)."""
        return ("l2h.w" in self.spec and "l1.w" in self.spec and self.M == 256 and self.H2 == 128 and self.dtype != torch.float32
                and self.sw["tail_geom"] in (6, 7))

    def _tail_fused(self) -> bool:
        """The dense tail (gate scaling + ReLU, Linear "1", Linear "2" + per-ray bias, sigma / colour heads) inside the expert forward
        launch (swn_chain_desc.tail_first, chain_big.hip tag 7): 256-feature experts, 128-feature layer "2", 16-bit compute dtype,
        experts local.  SWN_FUSED_TAIL=0 keeps the 64-row tail chain as its own launch."""
        return ("l2h.w" in self.spec and "l1.w" in self.spec and self.M == 256 and self.H2 == 128 and self.dtype != torch.float32
                and self.sw["fused_heads"] and self.L + 2 <= 12 and self.sw["fused_tail"])

    def set_expert_parallel(self, ep):
        """Shard the experts over the ranks of `ep` (parallel.ExpertParallel) and exchange the dispatched rows instead of
        computing every expert locally.  Ownership is sharded like the reference's (models/nerf_moe.py:139, 1037-1039: expert
        parameters are kept out of DDP): a rank computes the gradients of ITS experts from the rows of all ranks, only the dense
        prefix of the flat gradient buffer is all-reduced, and Adam moves only the owner's copy of an expert (the other ranks'
        copies see an exactly-zero gradient: their weights and moments never change and are never read).  The buffers keep
        their full size (3.7 M expert parameters); gather_expert_shards() refreshes every rank's copy from the owners before a
        checkpoint / evaluation without expert parallelism."""
        assert ep is None or ep.E == self.E
        if ep is None and self.ep is not None:       # leaving expert parallelism: every rank needs every expert's trained weights
            self.gather_expert_shards()
        self.ep = ep

    def gather_expert_shards(self, include_optimizer=True):
        """Every expert's parameters (and Adam moments) from its owner rank to all ranks (expert-parallel runs only): called by
        checkpoint.save_checkpoint and before leaving expert parallelism, so a checkpoint written by ANY rank holds every expert's
        trained weights.  The expert block is the contiguous tail of the flat buffers: per buffer, the owners' slices are packed
        rank-major, exchanged with ONE all_gather, and scattered back per tensor."""
        import torch.distributed as dist
        ep = self.ep
        if ep is None or ep.local:
            return
        W, El, r = ep.world, ep.El, ep.rank
        bufs = [self.flat] + ([self.m, self.v] if include_optimizer else [])
        spans = []                                   # (offset of the tensor in the flat buffer, elements per expert)
        for name, _shape in self._expert_spec:
            off, shape = self.spec[name]
            spans.append((off, int(np.prod(shape)) // self.E))
        per_rank = sum(El * per for _o, per in spans)
        for b in bufs:
            mine = torch.cat([b[off + r * El * per: off + (r + 1) * El * per] for off, per in spans])
            parts = [torch.empty_like(mine) for _ in range(W)]
            dist.all_gather(parts, mine, group=ep.group)       # (the list form: every backend implements it, gloo included)
            full = torch.stack(parts, 0)
            q = 0
            for off, per in spans:
                b[off: off + self.E * per].view(W, El * per).copy_(full[:, q:q + El * per])
                q += El * per
        self.refresh_compute_copies()

    def _local_experts(self, t, per_expert_leading=True):
        """Slice of a per-expert tensor / packed weight stream that belongs to this rank's experts."""
        if self.ep is None:
            return t
        El, r = self.ep.El, self.ep.rank
        if t.dim() == 1:                     # packed weights: [E * per]
            per = t.numel() // self.E
            v = t[r * El * per:(r + 1) * El * per]
            v.swn_nk = t.swn_nk
            return v
        return t[r * El:(r + 1) * El]

    # ------------------------------------------------------------------------------------------ buffers
    _saving = True          # False inside an inference forward: the chains skip the activation saves and ReLU masks
    hash = None             # multiresolution hash-grid input encoding (cfg["hash"]) or None
    _grow_bufs = False      # True: one buffer per name, grown to the largest row count seen (row counts that vary per step)

    def _buf(self, name, shape, dtype):
        if self._grow_bufs:
            key = (name, tuple(shape[1:]), dtype)
            b = self._bufs.get(key)
            if b is None or b.shape[0] < shape[0]:
                b = torch.empty(shape, dtype=dtype, device=self.dev)
                self._bufs[key] = b
            return b[: shape[0]]
        key = (name, tuple(shape), dtype)
        b = self._bufs.get(key)
        if b is None:
            b = torch.empty(shape, dtype=dtype, device=self.dev)
            self._bufs[key] = b
        return b

    def _linspace(self, n):
        """torch.linspace(0, 1, n) computed on the host like the reference's CPU path, cached on the device (no host-to-device copy
        inside a step: a copy from pageable memory cannot be captured into a hipGraph)."""
        key = ("linspace", int(n))
        t = self._bufs.get(key)
        if t is None:
            t = torch.linspace(0, 1, int(n), dtype=torch.float32).to(self.dev)
            self._bufs[key] = t
        return t

    # ------------------------------------------------------------------------------------------ forward
    def forward_rays(self, rays, image_indices, n_samples, seg_tokens, perturb=0.0, perturb_rand=None, sigma_noise=None,
                     training=True, routing_override=None, no_batch=False, z_in=None, pe_dir=None, tag="c",
                     want_weights=False, composite=True):
        """One pass of render_rays' _inference (rendering.py:277-494) over N rays x n_samples points: sampling (or the
        caller's depths z_in for the fine pass), positional encoding, the network, and (optionally) compositing.
        Returns a context dict holding every tensor the backward needs and the rendered results."""
        o, dt, dev = ops, self.dtype, self.dev
        N, S = rays.shape[0], n_samples
        self._sync_compute_copies()       # (an optimizer outside this class may have stepped flat_param since the last forward)
        if self.hash is not None:         # hash-grid encoding of the sample positions (BASELINE configs[4])
            z = z_in if z_in is not None else o.sample_z(rays, self._linspace(S), perturb_rand,
                                                         perturb, S)
            pe = o.hash_encode_fwd(rays, z, self.p["hash.table"], self.hash, dt, self.KP)
            if pe_dir is None:
                pe_dir = self._dir_pe(rays)
        elif z_in is None:
            t_steps = self._linspace(S)
            z, pe, pe_dir = o.sample_pe(rays, t_steps, perturb_rand, perturb, S, self.cfg["pos_xyz_dim"],
                                        self.cfg["pos_dir_dim"], dt, self.KP, self.DP)
        else:
            z = z_in
            pe = o.pe_from_z(rays, z_in, self.cfg["pos_xyz_dim"], dt, self.KP)      # xyz_fine_fn, rendering.py:103
            if pe_dir is None:
                pe_dir = self._dir_pe(rays)
        self._saving = bool(training)      # inference: no activation saves / ReLU masks (nothing will run backward)
        try:
            c = self._net_forward(pe, pe_dir, image_indices, N, S, seg_tokens, sigma_noise, routing_override, no_batch, tag)
        finally:
            self._saving = True
        c["z"], c["rays"] = z, rays
        if composite:
            c["rgb"], c["depth"], c["depth_variance"], c["weights"] = o.composite_fwd(c["raw"], c["z"], want_weights=want_weights)
        return c

    def _dir_pe(self, rays):
        """PE of the ray directions only (per ray)."""
        N = rays.shape[0]
        l_xyz = 0 if self.hash is not None else self.cfg["pos_xyz_dim"]      # (the position encoding of this call is discarded)
        _, _, pe_dir = ops.sample_pe(rays, torch.zeros(1, device=self.dev), None, 0.0, 1, l_xyz,
                                     self.cfg["pos_dir_dim"], self.dtype, self.KP, self.DP)
        return pe_dir

    def _net_forward(self, pe, pe_dir, image_indices, N, S, seg_tokens, sigma_noise, routing_override, no_batch, tag):
        """NeRFMoE.forward over the N*S points whose encodings are in `pe`, evaluated in model chunks of seg_tokens points like
        the reference's loop (rendering.py:354-383): routing, capacity and l_aux are per chunk.  A ragged last chunk
        (N*S not a multiple of seg_tokens) is routed on its own with its own capacity, exactly as the reference does - in
        evaluation and in training (backward_net walks the two parts)."""
        P = N * S
        seg_tokens = min(seg_tokens, P)
        if P % seg_tokens == 0:
            return self._net_forward_rows(pe, pe_dir, image_indices, N, S, seg_tokens, sigma_noise, routing_override, no_batch, tag, None)
        main = (P // seg_tokens) * seg_tokens
        a = self._net_forward_rows(pe, pe_dir, image_indices, N, S, seg_tokens, sigma_noise, routing_override, no_batch, tag, (0, main))
        b = self._net_forward_rows(pe, pe_dir, image_indices, N, S, P - main, sigma_noise, routing_override, no_batch, tag + "t", (main, P))
        c = dict(N=N, S=S, P=P, n_seg=a["n_seg"] + 1, seg_tokens=seg_tokens, tag=tag, image_indices=image_indices, ragged=True,
                 parts=(a, b), pe=pe, pe_dir=pe_dir)
        for k in ("raw", "l_aux", "idx", "gmax"):
            c[k] = torch.cat([a[k], b[k]], 0)
        return c

    def _join_side_outputs(self, ev, *tensors):
        """The launch stream waits for work issued on the side stream.  Allocation invariant: tensors allocated on the side
        stream (ray_feat, c_ray) are consumed on the launch stream, tensors allocated on the launch stream (dc_ray, dh2, dsig in
        backward_net_a) are read on the side stream - safe because EVERY side-stream section starts with side.wait_event(<launch-stream
        event>) and ends in an event the launch stream waits for before the tensors' last use, and the readers' inputs are kept alive in the
        returned state until that join: no block can be handed to the other stream's allocations while it is still in use.
        (`Tensor.record_stream` is deliberately NOT used: these tensors live in a captured graph's private pool - blocks with recorded
        streams are freed through deferred events when the graph is dropped; a bench.py run that creates and drops two graphed steps died
        with SIGSEGV once in a while with it, round 6.)"""
        torch.cuda.current_stream().wait_event(ev)
        if _EXP_RECORD_STREAM:      # (experiment only: scripts/experiments/graph_drop_stress.py reproduces the crash with it)
            for t in tensors:
                if t is not None and t.is_cuda:
                    t.record_stream(torch.cuda.current_stream())

    def _net_forward_rows(self, pe, pe_dir, image_indices, N, S, seg_tokens, sigma_noise, routing_override, no_batch, tag, row_range):
        """One or more whole model chunks: front chain, gate, routing, expert chain, tail chain, heads -> c["raw"] [P, 4].
        row_range = None: all N*S points; (r0, r1): that row range of the ray-major point grid (ragged evaluation)."""
        o, dt, dev = ops, self.dtype, self.dev
        P = N * S
        if row_range is not None:
            r0, r1 = row_range
            P = r1 - r0
            pe = pe[r0:r1]
            sigma_noise = None if sigma_noise is None else sigma_noise[r0:r1]
            routing_override = None if routing_override is None else routing_override[r0:r1]
        M, E, L, G, H2 = self.M, self.E, self.L, self.G, self.H2
        assert P % seg_tokens == 0, "points must be a multiple of the segment (model chunk) size"
        n_seg = P // seg_tokens
        cap = int(self.cf * ((seg_tokens + E - 1) // E))     # tutel_fast_dispatch.py:211
        if no_batch:        # eval path (apply_on_expert_fn_nobatch): nothing is dropped == a capacity nothing exceeds
            cap = seg_tokens
        c = dict(N=N, S=S, P=P, n_seg=n_seg, cap=cap, seg_tokens=seg_tokens, tag=tag, image_indices=image_indices)
        c["pe"], c["pe_dir"] = pe, pe_dir
        # the per-ray half of layer "2" (gather + a 75 x 128 GEMM per ray: one small launch) only needs the direction encoding: on the side
        # stream it runs under the front chain / the router instead of between the routing and the expert launch
        ray_feat_ev = None
        if (self.overlap and self.side is not None and not self.profile and self.ep is None and row_range is None and self._tail_fused()
                and "l2r.w" in self.p):
            ev0 = torch.cuda.Event()
            ev0.record()
            with torch.cuda.stream(self.side):
                self.side.wait_event(ev0)
                c["ray_feat"], c["c_ray"] = ops.ray_feat_fwd(pe_dir, self.in_dir, self.p["emb"], image_indices.contiguous(), self.p["l2r.w"],
                                                             self.p["l2.b"])
                ray_feat_ev = torch.cuda.Event()
                ray_feat_ev.record()
        _b = lambda name, shape, dtype: self._buf(tag + ":" + name, shape, dtype)
        # ---- front chain: PE -> xyz -> gate MLP
        c["h0"] = _b("h0", (P, M), dt)
        c["a1"] = _b("a1", (P, G), dt)
        c["g"] = _b("g", (P, G), dt)
        c["m_a1"] = _b("m_a1", (o.chain_mask_words(dt, 1, P, max(M, G, self.KP)),), torch.int32)
        sv = self._saving
        c["no_grad"] = not sv
        # (the 64-row kernels re-stream the 320 KB of weights from L2 for every 64 rows - 11.8 GB of L2 reads per 2M-point launch against
        #  0.5 GB of input, the CU's L2 -> L1 path is what bounds them - the persistent 256-row geometry a quarter of that)
        c["front_geom"] = self.sw["front_geom"] if self._front_big() else 1
        big = c["front_geom"] >= 6
        o.mlp_chain(c["pe"], [o.Layer(self.wf["xyz_pad" if big else "xyz"], self.p["xyz.b"].view(1, M), save=c["h0"]),
                              o.Layer(self.wf["gate0"], self.p["gate0.b"].view(1, G), relu=1, mask=c["m_a1"] if sv else None,
                                      save=c["a1"] if sv else None),
                              o.Layer(self.wf["gate1"], self.p["gate1.b"].view(1, G))], c["g"], tag=3, geometry=c["front_geom"] if big else 0,
                    x_features=self.KP if big else 0)
        # ---- gate + routing
        gnoise = None
        if sv and self.gate_noise > 0:       # (training only, like `self.training and self.gate_noise > 0`)
            if self.gate_noise_draw is not None:      # a supplied draw (tests): [N * S, E] of this pass's point grid; a row range takes its rows
                draw = self.gate_noise_draw.to(dev, torch.float32)
                if draw.numel() != N * S * E:
                    raise ValueError(f"gate_noise_draw holds {draw.numel()} values, this pass needs [{N * S}, {E}] (one row per point)")
                draw = draw.reshape(N * S, E)
                gnoise = (draw if row_range is None else draw[row_range[0]:row_range[1]]).contiguous()
            else:
                gnoise = torch.randn(P, E, device=dev, dtype=torch.float32)
        c["gates"], c["idx"], c["gmax"], c["stats"] = o.gate_fwd(c["g"], self.p["ln.w"], self.p["ln.b"], self.p["wg"], noise=gnoise,
                                                                 noise_scale=self.gate_noise / E)
        if routing_override is not None:     # tests: inject the oracle's expert choice (near-tie robustness)
            c["idx"] = routing_override.to(dev).int().contiguous()
            c["gmax"] = c["gates"].gather(1, c["idx"].long()[:, None])[:, 0].contiguous()
        packed = bool(no_batch) and not sv and self.ep is None      # (see below: the no-batch row layout of the inference forward)
        # expert chains (forward here, backward-data in backward_net - the pair shares its ReLU mask layout): the 256-row geometry
        # with phase-shifted row groups as a persistent launch (chain_big.hip, geometry 7 = geometry 4 walking a tile queue) for 256-feature
        # experts in a 16-bit compute dtype once a group holds at least one full tile.  The chain_geom switch picks another one (1: the
        # 64-row kernels, 2 / 4 / 5 / 6: see include/swn.h) - tests.
        c["geom"] = self.sw["chain_geom"] if (M == 256 and dt != torch.float32 and cap >= 256) else 1
        # the tail inside the expert launch (local experts, standard row space, whole point grid): the expert output never reaches memory
        c["tail_fused"] = (self._tail_fused() and self.ep is None and c["geom"] == 7 and row_range is None
                           and P * M * c_esz(dt) < (1 << 32) - 64)
        # expert parallelism with the tail on the expert's rank (ep_owner.py): the same fused launches, on a received token space
        from . import ep_owner
        owner = ep_owner.eligible(self, c, no_batch, row_range)
        # (the fused tail's list of dropped tokens comes out of the routing launch)
        want_drops = (c["tail_fused"] or owner) and not packed
        routed = o.route_top1(c["idx"], c["gmax"], c["gates"], seg_tokens, E, cap, self.bpr, want_perm=not packed, want_drops=want_drops)
        c["loc"], c["counts"], c["perm"], c["tok2row"], c["l_aux"] = routed[:5]
        if want_drops:
            c["drop_begin"], c["dropped"] = routed[5], routed[6]
        # ---- expert chain (gathers its rows through perm; ragged groups = (segment, expert))
        rows = n_seg * E * cap
        ng = n_seg * E
        # evaluation without token dropping (apply_on_expert_fn_nobatch, tutel_moe_layer_nobatch.py:237-352): the reference packs the
        # rows contiguously per expert (expert_locations_begin, tutel_fast_dispatch_nobatch.py:24-36).  Same layout here for the
        # inference forward: P rows instead of n_seg * E * seg_tokens, groups addressed through their first row.
        group_begin = None
        if packed:
            rows = P
            group_begin, c["perm"], c["tok2row"] = o.route_pack(c["idx"], c["loc"], c["counts"], seg_tokens, E)
            c["group_begin"] = group_begin
        c["rows"], c["ng"] = rows, ng
        c["counts_flat"] = c["counts"].view(-1)
        c["saves"] = [_b(f"save{l}", (rows, M), dt) if sv else None for l in range(L - 1)]
        nw = max(o.chain_mask_words(dt, ng, cap, M), n_seg * o.chain_mask_words(dt, E, cap, M))    # (expert parallel: one launch per segment)
        c["masks"] = [_b(f"mask{l}", (nw,), torch.int32) if sv else None for l in range(L - 1)]
        skips = set(self.cfg["skips"])
        layers = [o.Layer(self._local_experts(self.wf[f"exp{l}"]), self._local_experts(self.p[f"exp{l}.b"]),
                          relu=1 if l < L - 1 else 0, skip=(l in skips), save=c["saves"][l] if (sv and l < L - 1) else None,
                          mask=c["masks"][l] if (sv and l < L - 1) else None) for l in range(L)]
        c["eo"] = None if c["tail_fused"] else _b("eo", (rows, M), dt)
        self._kernel_sel.update(geom=c["geom"], front_geom=c["front_geom"], tail_fused=bool(c["tail_fused"]))
        if no_batch and not sv and self.ep is not None:
            # evaluation without token dropping under expert parallelism (tutel_moe_layer_nobatch.py:308-335): the packed rows of a
            # segment, expert-major = (destination rank, local expert), travel with UNEQUAL splits (the reference's list_all_to_all);
            # the owner runs its experts on the (source rank, local expert) groups it received and sends the rows back
            ep = self.ep
            _gb, perm_p, c["tok2row"] = o.route_pack(c["idx"], c["loc"], c["counts"], seg_tokens, E)
            c["row_of_tok"] = c["tok2row"]
            c["eo"] = _b("eo_packed", (P, M), dt)
            for s_ in range(n_seg):
                rs = slice(s_ * seg_tokens, (s_ + 1) * seg_tokens)          # every segment holds exactly seg_tokens packed rows
                recv, rc = ep.all_to_all_ragged(o.gather_rows(c["h0"], perm_p[rs]), c["counts"][s_].contiguous())
                out = torch.empty_like(recv)
                if recv.shape[0]:
                    gb = (torch.cumsum(rc, 0) - rc).to(torch.int32)
                    o.mlp_chain(recv, layers, out, n_groups=ep.world * ep.El, n_wsets=ep.El, group_stride=seg_tokens, group_rows=rc,
                                group_rows_clamp=seg_tokens, tag=1, geometry=c["geom"], group_begin=gb)
                back, _ = ep.all_to_all_ragged(out, rc, recv_counts=c["counts"][s_].contiguous())
                c["eo"][rs] = back
        elif self.ep is None and c["tail_fused"]:
            # ---- experts + tail in ONE launch (chain_big.hip, tag 7): behind the last expert layer a row is scaled by its gate value and
            # ReLU'd (the decoded MoE output y), runs through layer "1" and layer "2" (+ the per-ray bias below) and the two heads; y, h1
            # and h2 are saved in TOKEN order for the backward (a training forward), raw is written in token order; the tokens no expert
            # kept enter at layer "1" as zero rows (swn_route_dropped).  Replaces: the expert output's round trip through memory, the
            # 64-row tail chain (which re-streams its 192 KiB of weights from L2 for every 64 rows) and its launch.
            c["row_of_tok"] = c["tok2row"]
            if ray_feat_ev is not None:
                self._join_side_outputs(ray_feat_ev, c["ray_feat"], c["c_ray"])
            else:
                c["ray_feat"], c["c_ray"] = o.ray_feat_fwd(pe_dir, self.in_dir, self.p["emb"], image_indices.contiguous(), self.p["l2r.w"], self.p["l2.b"])
            if "dropped" not in c:
                c["drop_begin"], c["dropped"] = o.route_dropped(c["idx"], c["loc"], c["counts"], seg_tokens, E, cap)
            if "l2h_pad" not in self.wf:      # (SWN_FUSED_TAIL switched on after the compute copies were made)
                self.wf["l2h_pad"] = o.pack_weights_padded(self.p["l2h.w"].unsqueeze(0), dt, True, 0, 256)
            c["y"] = _b("y", (P, M), dt) if sv else None
            c["h1"] = _b("h1", (P, M), dt) if sv else None
            c["h2"] = _b("h2", (P, H2), dt) if sv else None
            c["raw"] = torch.empty(P, 4, dtype=torch.float32, device=dev)
            heads = (self.p["sigma.w"], self.p["sigma.b"], self.p["color.w"], self.p["color.b"], sigma_noise, c["raw"])

            def run_experts(save=sv):
                lys = [o.Layer(ly.w, ly.b, relu=ly.relu, skip=ly.skip, save=ly.save if save else None, mask=ly.mask if save else None)
                       for ly in layers]
                lys[-1].save = c["y"] if save else None
                lys += [o.Layer(self.wf["l1"], self.p["l1.b"].view(1, M), save=c["h1"] if save else None),
                        o.Layer(self.wf["l2h_pad"], None, relu=1, rowbias=c["c_ray"], rows_per_bias=S)]
                o.mlp_chain(c["h0"], lys, c["h2"] if save else None, n_groups=ng, n_wsets=E, group_stride=cap, group_rows=c["counts_flat"],
                            group_rows_clamp=cap, x_gather=c["perm"].view(-1), tag=7, geometry=7, heads=heads, group_begin=group_begin,
                            tail=(L, c["gmax"], c["drop_begin"], c["dropped"], H2))
            with self._timed("expert_fwd"):
                run_experts()
            if self.profile:      # bench.py: the launch again on this step's live buffers, its save-free form, and the expert layers ALONE
                def expert_gemm():    # (the grouped GEMM without the tail: tag 1, into a scratch output - what round 1-3's figure measured)
                    o.mlp_chain(c["h0"], [o.Layer(ly.w, ly.b, relu=ly.relu, skip=ly.skip) for ly in layers], _b("eo_probe", (rows, M), dt),
                                n_groups=ng, n_wsets=E, group_stride=cap, group_rows=c["counts_flat"], group_rows_clamp=cap,
                                x_gather=c["perm"].view(-1), tag=1, geometry=7, group_begin=group_begin)
                c["_relaunch"] = {"expert_fwd": run_experts, "expert_fwd_nosave": lambda: run_experts(False), "expert_gemm_nosave": expert_gemm}
            return c
        elif self.ep is None:
            c["row_of_tok"] = c["tok2row"]

            def run_experts(lys=layers):
                o.mlp_chain(c["h0"], lys, c["eo"], n_groups=ng, n_wsets=E, group_stride=cap, group_rows=c["counts_flat"],
                            group_rows_clamp=cap, x_gather=c["perm"].view(-1), tag=1, geometry=c["geom"], group_begin=group_begin)
            with self._timed("expert_fwd"):
                run_experts()
            if self.profile:      # bench.py: the same launch again, back to back, on this step's live buffers (and its save-free form)
                c["_relaunch"] = {"expert_fwd": run_experts,
                                  "expert_fwd_nosave": lambda: run_experts([o.Layer(ly.w, ly.b, relu=ly.relu, skip=ly.skip) for ly in layers])}
        else:
            # expert parallel, pipelined per routing segment (parallel.ExpertParallel), KEPT ROWS ONLY: the kept rows of segment s, packed in
            # (expert, slot) order = payload order (destination rank, local expert, slot) -> unequal-split all-to-all on the side stream ->
            # the local experts run on the (source rank, local expert) groups (first rows from a device prefix sum) -> all-to-all back into
            # the packed row space the combine gathers from.  The dispatch of segment s + 1 and the return of segment s - 1 travel while
            # the experts work on segment s.  One host read of the counts per forward pass (ExpertParallel.plan).
            if owner:
                return ep_owner.forward(self, c, pe_dir, image_indices, sigma_noise, sv)
            ep = self.ep
            W, El = ep.world, ep.El
            ngs = W * El
            recv_counts = ep.exchange_counts(c["counts"], cap, self.side)()            # [n_seg, W * E_local], the expert kernels' group order
            c["ep_counts"] = recv_counts
            c["ep_padded"] = ep.use_padded(E * cap * M * c_esz(dt))
            if c["ep_padded"]:
                # the reference's layout (tutel_moe_layer_nobatch.py:157): every (expert, capacity slot) of a segment travels - empty slots
                # as zero rows - with EQUAL splits.  The row spaces are the standard ones (perm / tok2row of swn_route_top1, group g at row
                # g * cap), nothing is read on the host: the step can be captured into a hipGraph, collectives included.
                perm_p, c["row_of_tok"] = c["perm"].view(-1), c["tok2row"]
                seg_rows = E * cap
                pl = dict(in_splits=[[El * cap] * W] * n_seg, out_splits=[[El * cap] * W] * n_seg,
                          send_off=[s_ * seg_rows for s_ in range(n_seg + 1)], recv_off=[s_ * seg_rows for s_ in range(n_seg + 1)])
                key = ("ep_begin_padded", n_seg * ngs, cap)
                if key not in self._bufs:
                    self._bufs[key] = (torch.arange(n_seg * ngs, device=dev, dtype=torch.int32) * cap).contiguous()
                c["ep_begin"] = self._bufs[key]
            else:
                kept = c["counts"].clamp(max=cap)
                idx_kept = torch.where(c["loc"] < cap, c["idx"], torch.full_like(c["idx"], -1))
                _gb, perm_p, c["row_of_tok"] = o.route_pack(idx_kept, c["loc"], kept, seg_tokens, E)     # packed row space of this rank's rows
                pl = ep.plan(kept, recv_counts)
                flat_rc = recv_counts.reshape(-1)
                c["ep_begin"] = (torch.cumsum(flat_rc, 0, dtype=torch.int32) - flat_rc).contiguous()      # first row of every received group
            c["ep_perm"], c["ep_plan"] = perm_p, pl
            so, ro = pl["send_off"], pl["recv_off"]
            xr = _b("ep_x", (rows, M), dt)                                  # received rows of all segments (also the first layer's
            send = xr if ep.local else _b("ep_send_x", (rows, M), dt)         # weight-gradient operand)
            eo_r = c["eo"] if ep.local else _b("ep_eo", (rows, M), dt)        # expert outputs in the received row space
            wseg = o.chain_mask_words(dt, E, cap, M)
            c["ep_mask_words"] = wseg

            def issue(s_):
                o.gather_rows(c["h0"], perm_p[so[s_]:so[s_ + 1]], send[so[s_]:so[s_ + 1]])
                return ep.all_to_all_v(send[so[s_]:so[s_ + 1]], pl["in_splits"][s_], xr[ro[s_]:ro[s_ + 1]], pl["out_splits"][s_], self.side)
            with self._timed("expert_fwd"):
                pend = issue(0)
                returns = []
                for s_ in range(n_seg):
                    nxt = issue(s_ + 1) if s_ + 1 < n_seg else None
                    pend()
                    layers_s = [o.Layer(ly.w, ly.b, relu=ly.relu, skip=ly.skip, save=ly.save,
                                        mask=None if ly.mask is None else ly.mask[s_ * wseg:(s_ + 1) * wseg]) for ly in layers]
                    if ro[s_ + 1] > ro[s_]:
                        o.mlp_chain(xr, layers_s, eo_r, n_groups=ngs, n_wsets=El, group_stride=cap, group_rows=recv_counts[s_],
                                    group_rows_clamp=cap, tag=1, geometry=c["geom"], group_begin=c["ep_begin"][s_ * ngs:(s_ + 1) * ngs])
                    returns.append(ep.all_to_all_v(eo_r[ro[s_]:ro[s_ + 1]], pl["out_splits"][s_], c["eo"][so[s_]:so[s_ + 1]],
                                                   pl["in_splits"][s_], self.side))
                    pend = nxt
                for wait in returns:
                    wait()
            c["ep_x"] = xr
        # ---- per-ray part of layer "2": [PE(dir), appearance embedding] @ W2r + b2   (N_rays x 75, host-side torch)
        # (one launch: swn_ray_feat_fwd - was cat / embedding lookup / addmm in torch)
        if ray_feat_ev is not None:        # (issued on the side stream above; this branch: the fused tail was not taken after all)
            self._join_side_outputs(ray_feat_ev, c["ray_feat"], c["c_ray"])
        else:
            c["ray_feat"], c["c_ray"] = o.ray_feat_fwd(pe_dir, self.in_dir, self.p["emb"], image_indices.contiguous(), self.p["l2r.w"], self.p["l2.b"])
        # ---- tail chain.  Its input load IS the combine: rows gathered from the expert output through tok2row, scaled by
        # the gate value, ReLU'd (dropped tokens -> zero rows) and saved as y;  then layer "1" -> layer "2" (+ per-ray bias)
        # The sigma / colour heads run inside that launch (swn.h: heads_raw): sigma from the staged y tile, colour from the h2 tile.  A
        # training forward still writes y, h1 and h2 (the backward reads them); an inference forward writes nothing but raw.
        fused = self.sw["fused_heads"] and M in (256, 512) and H2 in (128, 256) and M * c_esz(dt) <= 1024      # (rows of at most 1 KiB)
        keep = sv or not fused
        c["y"] = _b("y", (P, M), dt) if keep else None
        c["h1"] = _b("h1", (P, M), dt) if sv else None
        c["h2"] = _b("h2", (P, H2), dt) if keep else None
        c["raw"] = torch.empty(P, 4, dtype=torch.float32, device=dev) if fused else None
        rowbias, rpb = c["c_ray"], S
        if row_range is not None:  # a row range of the point grid: the rows' rays through an explicit per-row gather
            rowbias, rpb = c["c_ray"].index_select(0, torch.arange(r0, r1, device=dev) // S), 1
            c["ragged"], c["row0"] = True, r0
        o.mlp_chain(c["eo"], [o.Layer(self.wf["l1"], self.p["l1.b"].view(1, M), save=c["h1"] if sv else None),
                              o.Layer(self.wf["l2h"], None, relu=1, rowbias=rowbias, rows_per_bias=rpb)], c["h2"],
                    group_stride=P, x_gather=c["row_of_tok"], x_save=c["y"], x_scale=c["gmax"], x_relu=True, tag=4,
                    heads=(self.p["sigma.w"], self.p["sigma.b"], self.p["color.w"], self.p["color.b"], sigma_noise, c["raw"]) if fused else None)
        if not fused:      # ---- heads as their own launch (SWN_NO_FUSED_HEADS=1: rounds 1-2)
            c["raw"] = o.heads_fwd(c["y"], c["h2"], self.p["sigma.w"], self.p["sigma.b"], self.p["color.w"], self.p["color.b"],
                                   sigma_noise)
        return c

    # ------------------------------------------------------------------------------------------ backward
    def backward(self, c, d_rgb, d_laux):
        """Accumulates parameter gradients into self.grad given dL/d rgb [N,3] and dL/d l_aux[seg] (scalar each)."""
        d_raw = ops.composite_bwd(c["raw"], c["z"], d_rgb)
        self.backward_net(c, d_raw, d_laux)

    def backward_net(self, c, d_raw, d_laux):
        """Backward of _net_forward given dL/d raw [N*S, 4] and dL/d l_aux[seg]; accumulates into self.grad.
        = backward_net_a (heads, tail, expert chain, expert weight gradients: after it the EXPERT block of the flat gradient - 93 % of
        its bytes - is final) + backward_net_b (router, front chain, dense weight gradients).  A data-parallel step replayed from
        graphs cuts there: the all-reduce of the expert block travels while the second half runs (graph.GraphedTrainStep)."""
        if c.get("parts") is not None:
            # a ragged last model chunk (rendering.py:354-383 trains any batch size): the whole chunks and the short last chunk were
            # routed as two contexts with their own capacities; each runs its own backward on its rows of dL/d raw and its chunks'
            # share of the l_aux gradient
            a, b = c["parts"]
            if self.hash is not None:      # the parts leave dL/d encoding in their rows of one [P, KP] buffer; the table's gradient is
                d_enc = self._buf(c["tag"] + ":d_enc", (c["P"], self.KP), self.dtype)       # one launch over the whole rays behind them
                a["d_enc_out"], b["d_enc_out"] = d_enc[: a["P"]], d_enc[a["P"]:]
            self.backward_net(a, d_raw[: a["P"]], d_laux[: a["n_seg"]].contiguous())
            self.backward_net(b, d_raw[a["P"]:], d_laux[a["n_seg"]:].contiguous())
            if self.hash is not None:
                ops.hash_encode_bwd(c["rays"], c["z"], d_enc, self.hash, self.g["hash.table"])
            return
        self.backward_net_b(self.backward_net_a(c, d_raw, d_laux))

    def backward_net_a(self, c, d_raw, d_laux, join_side=False):
        """First half of backward_net (see there); returns the state backward_net_b continues from.  join_side: the launch stream
        waits for the expert weight gradients before returning (they run on the side stream otherwise and are joined at the end of
        the second half)."""
        o, dt = ops, self.dtype
        if c.get("no_grad"):
            raise RuntimeError("this context comes from an inference forward (training=False): nothing was saved for the backward")
        assert c.get("parts") is None, "ragged contexts go through backward_net"
        if c.get("ep_owner") is not None:      # expert parallelism with the tail on the expert's rank
            from . import ep_owner
            return ep_owner.backward_a(self, c, d_raw, d_laux)
        N, S, P, n_seg, cap, seg_tokens = c["N"], c["S"], c["P"], c["n_seg"], c["cap"], c["seg_tokens"]
        M, E, L, G, H2 = self.M, self.E, self.L, self.G, self.H2
        g = self.g
        rows, ng = c["rows"], c["ng"]
        _b = lambda name, shape, dtype: self._buf(c["tag"] + ":" + name, shape, dtype)
        # the tail's two backward layers and the combine backward in FRONT of the expert backward chain, one launch (chain_big.hip, tag 8):
        # pairs with the fused forward (its list of dropped tokens); SWN_FUSED_TAIL_BWD=0 keeps the two launches
        fused_bwd = bool(c.get("tail_fused")) and self.ep is None and self.sw["fused_tail_bwd"]
        # ... which also forms the sigma head's weight gradient where it reads y anyway (swn_chain_desc.comb_dwsig): the heads' backward
        # launch then runs without y - 512 bytes per point less (SWN_FUSED_DWSIG=0: from y in the heads' launch, as before)
        fused_dws = fused_bwd and M == 256 and self.sw["fused_dwsig"]
        y_heads = None if fused_dws else c["y"]
        self._kernel_sel.update(fused_backward=fused_bwd, comb_dwsig=fused_dws)
        # per-ray bias gradient (the column sums of a ray's dh2 rows: from the heads' launch) and the tiny per-ray GEMM's parameters
        if c.get("ragged"):      # a row range of the point grid: rays may be cut at either end - per-ray sums through the rows' ray index
            dh2, dsig = o.heads_bwd(y_heads, c["h2"], self.p["color.w"], c["raw"], d_raw, g["sigma.w"], g["sigma.b"], g["color.w"],
                                    g["color.b"])
            ray_of_row = torch.arange(c["row0"], c["row0"] + P, device=self.dev) // S
            dc_ray = torch.zeros(N, H2, dtype=torch.float32, device=self.dev).index_add_(0, ray_of_row, dh2.float())
        else:
            dh2, dsig, dc_ray = o.heads_bwd(y_heads, c["h2"], self.p["color.w"], c["raw"], d_raw, g["sigma.w"], g["sigma.b"], g["color.w"],
                                            g["color.b"], rows_per_group=S)
        def ray_level():
            if dc_ray.shape[1] in (64, 128, 256) and c["ray_feat"].shape[1] <= 256:      # split over the rays + ordered reduce (one launch)
                o.ray_feat_wgrad(c["ray_feat"], dc_ray, g["l2r.w"], g["l2.b"])
            else:
                g["l2r.w"].addmm_(c["ray_feat"].t(), dc_ray)
                g["l2.b"].add_(dc_ray.sum(0))
            d_feat_emb = dc_ray @ self.p["l2r.w"][self.in_dir:].t()
            o.emb_grad(d_feat_emb, c["image_indices"].contiguous(), g["emb"])        # (rays added in order: torch's index_add_ uses atomics)
        # The per-ray work (layer "2"'s ray half and the appearance embedding: three small launches + a reduce, ~0.1 ms in which 10-30
        # workgroups hold the chip) goes to the SIDE stream: it only needs dc_ray and writes gradient slices nobody else touches, so it
        # runs under the expert backward launch that follows (whose resident workgroups pick the few late CUs up through the tile queue);
        # joined with the expert weight gradients (same stream) or at the end of this half.  SWN_NO_OVERLAP=1: on the launch stream.
        side_small = None
        if self.overlap and self.side is not None and not self.profile and not c.get("ragged") and self.ep is None:
            ready_small = torch.cuda.Event()
            ready_small.record()
            with torch.cuda.stream(self.side):
                self.side.wait_event(ready_small)
                ray_level()
                side_small = torch.cuda.Event()
                side_small.record()
        else:
            ray_level()
        # tail backward chain: dh2 -> dh1 -> dy
        # ... with the combine backward (the sigma head's rank-1 term, the ReLU mask of y, the gate gradient, the gate scaling) applied
        # in the write-out of the last layer: dy itself never reaches memory
        dh1 = _b("dh1", (P, M), dt)
        dout = None if fused_bwd else _b("dy", (P, M), dt)
        if fused_bwd:
            dgmax = _b("dgmax", (P,), torch.float32)
            if "l2h_pad" not in self.wb:
                self.wb["l2h_pad"] = o.pack_weights_padded(self.p["l2h.w"].unsqueeze(0), dt, False, 0, 256)
        elif M * dout.element_size() <= 1024:      # (a row's 16-byte chunks must fit one wavefront: everything but fp32 rows of 512)
            dgmax = _b("dgmax", (P,), torch.float32)
            tg = self.sw["tail_geom"] if self._tail_big() else 0
            o.mlp_chain(dh2, [o.Layer(self.wb["l2h_pad" if tg >= 6 else "l2h"], None, save=dh1), o.Layer(self.wb["l1"], None)], dout, tag=5,
                        combine=(c["y"], dsig, self.p["sigma.w"], c["gmax"], dgmax), geometry=tg, x_features=H2 if tg >= 6 else 0)
        else:
            o.mlp_chain(dh2, [o.Layer(self.wb["l2h"], None, save=dh1), o.Layer(self.wb["l1"], None)], dout, tag=5)
            dout, dgmax = o.combine_bwd(dout, c["y"], dsig, self.p["sigma.w"], c["gmax"])
        nsp = max(1, min(256, P // 1024))          # row splits of the dense weight-gradient GEMMs: one workgroup per CU also at the
                                                   # per-GPU batch of an 8-GPU run (262144 points)
        ep = self.ep
        if ep is not None:      # the rows of the first segment travel to their experts while the tail's weight gradients run
            pl, perm_p = c["ep_plan"], c["ep_perm"]
            so, ro = pl["send_off"], pl["recv_off"]
            dr = _b("ep_d", (rows, M), dt)
            dsend = dr if ep.local else _b("ep_send_d", (rows, M), dt)

            def issue_b(s_):
                o.gather_rows(dout, perm_p[so[s_]:so[s_ + 1]], dsend[so[s_]:so[s_ + 1]])
                return ep.all_to_all_v(dsend[so[s_]:so[s_ + 1]], pl["in_splits"][s_], dr[ro[s_]:ro[s_ + 1]], pl["out_splits"][s_], self.side)
            pend = issue_b(0)
        # The weight gradients of the two tail layers: at small batches (the per-GPU share of a strong-scaling run) they go out together
        # with the front layers' at the end - one balanced launch + one reduction for all five dense layers (2.39 against 2.42 ms per
        # step at 1024 rays); at the full batch two launches are faster (15.15 against 15.30 ms: the tail operands are the most recently
        # written tensors when their launch follows the tail backward directly).  Measured on one box, scripts/ab_env.sh.
        tail_jobs = [(c["h1"], dh2, g["l2h.w"].view(1, M, H2), None), (c["y"], dh1, g["l1.w"].view(1, M, M), g["l1.b"].view(1, M))]
        if P > (1 << 19) and not fused_bwd:      # (fused backward: dh1 comes out of the expert launch - the tail's weight gradients follow it)
            self._dense_wgrads(tail_jobs, nsp)
            tail_jobs = []
        # expert backward chain
        dz = [_b(f"dz{l}", (rows, M), dt) for l in range(L - 1)]     # dz[L-1] = dout through perm: never materialised
        dx = _b("dx", (rows, M), dt)
        skip_l = list(self.cfg["skips"])[0] if len(self.cfg["skips"]) else None
        bl = []
        for i in range(L):
            l = L - 1 - i
            bl.append(o.Layer(self._local_experts(self.wb[f"exp{l}"]), None, relu=2 if l > 0 else 0,
                              mask=c["masks"][l - 1] if l > 0 else None, save=dz[l - 1] if l > 0 else None))
        n_loc = E if ep is None else ep.El
        grp_rows = c["counts_flat"] if ep is None else c["ep_counts"]
        if ep is None and fused_bwd:
            perm = c["perm"].view(-1)
            x_first, dz_last = c["h0"], _b("dz_last", (rows, M), dt)      # the last expert layer's dZ in the row space (the combine's output)

            dws_dst = [g["sigma.w"].view(-1) if fused_dws else None]

            def run_expert_bwd():
                o.mlp_chain(dh2, [o.Layer(self.wb["l2h_pad"], None, save=dh1), o.Layer(self.wb["l1"], None, save=dz_last)] + bl, dx, n_groups=ng,
                            n_wsets=n_loc, group_stride=cap, group_rows=grp_rows, group_rows_clamp=cap, x_gather=perm,
                            y_add=dz[skip_l] if skip_l is not None else None, tag=8, geometry=7, x_features=H2,
                            combine=(c["y"], dsig, self.p["sigma.w"], c["gmax"], dgmax, dws_dst[0]), head=(2, c["drop_begin"], c["dropped"]),
                            group_begin=c.get("group_begin"))
            with self._timed("expert_bwd"):
                run_expert_bwd()
            if self.profile and "_relaunch" in c:
                if fused_dws:      # (a relaunch for timing adds into a scratch vector, not into the gradient)
                    dws_dst[0] = torch.zeros(M, dtype=torch.float32, device=self.dev)
                c["_relaunch"]["expert_bwd"] = run_expert_bwd
            if P > (1 << 19):
                self._dense_wgrads(tail_jobs, nsp)
                tail_jobs = []
        elif ep is None:
            perm = c["perm"].view(-1)
            x_first, dz_last = c["h0"], dout            # read through the routing permutation
            def run_expert_bwd():
                o.mlp_chain(dz_last, bl, dx, n_groups=ng, n_wsets=n_loc, group_stride=cap, group_rows=grp_rows,
                            group_rows_clamp=cap, x_gather=perm, y_add=dz[skip_l] if skip_l is not None else None, tag=2,
                            geometry=c["geom"])
            with self._timed("expert_bwd"):
                run_expert_bwd()
            if self.profile and "_relaunch" in c:
                c["_relaunch"]["expert_bwd"] = run_expert_bwd
        else:
            # per segment like the forward pass: dispatch of segment s + 1 and return of segment s - 1 overlap the chain of segment s;
            # the input gradients come home into the packed row space (dx) that the front backward chain gathers from
            perm = None
            x_first, dz_last = c["ep_x"], dr                     # the received rows, already in group order
            grp_rows = c["ep_counts"].reshape(-1)
            ngs = ep.world * ep.El
            dx_r = dx if ep.local else _b("ep_dx", (rows, M), dt)
            wseg = c["ep_mask_words"]
            returns = []
            with self._timed("expert_bwd"):
                for s_ in range(n_seg):
                    nxt = issue_b(s_ + 1) if s_ + 1 < n_seg else None
                    pend()
                    bl_s = [o.Layer(ly.w, None, relu=ly.relu, save=ly.save,
                                    mask=None if ly.mask is None else ly.mask[s_ * wseg:(s_ + 1) * wseg]) for ly in bl]
                    if ro[s_ + 1] > ro[s_]:
                        o.mlp_chain(dr, bl_s, dx_r, n_groups=ngs, n_wsets=n_loc, group_stride=cap, group_rows=c["ep_counts"][s_],
                                    group_rows_clamp=cap, y_add=dz[skip_l] if skip_l is not None else None, tag=2, geometry=c["geom"],
                                    group_begin=c["ep_begin"][s_ * ngs:(s_ + 1) * ngs])
                    returns.append(ep.all_to_all_v(dx_r[ro[s_]:ro[s_ + 1]], pl["out_splits"][s_], dx[so[s_]:so[s_ + 1]], pl["in_splits"][s_],
                                                   self.side))
                    pend = nxt

        def expert_wgrads():
            # all layers in ONE launch: layer 0 reads its input rows, layer L-1 its dZ rows, through the routing permutation
            items = []
            for l in range(L):
                a = x_first if l == 0 else c["saves"][l - 1]
                bz = dz_last if l == L - 1 else dz[l]
                items.append((a, bz, self._local_experts(g[f"exp{l}.w"]), self._local_experts(g[f"exp{l}.b"]),
                              perm if l == 0 else None, perm if (l == L - 1 and not fused_bwd) else None))
            if ep is not None:      # received rows are packed: groups through their first rows (any width: wgrad_multi cuts 512-feature
                o.wgrad_multi(items, n_groups=ng, n_wsets=n_loc, group_stride=cap, group_rows=grp_rows, group_rows_clamp=cap, tag=1,
                              group_begin=c["ep_begin"])      # operands into 256-column GEMMs of the same launch)
                return
            for i0 in range(0, L, 8):
                o.wgrad_batched(items[i0:i0 + 8], n_groups=ng, n_wsets=n_loc, group_stride=cap, group_rows=grp_rows,
                                group_rows_clamp=cap, n_splits=self.expert_wgrad_splits or max(1, min(256 // ng, cap // 2048)), tag=1)
        side_done = None
        if self.overlap and self.side is not None and not self.profile and not join_side:
            # independent of everything that follows (they only read the saved activations / dZ and write their own
            # gradient slices): run them on the side stream, join before Adam
            ready = torch.cuda.Event()
            ready.record()
            with torch.cuda.stream(self.side):
                self.side.wait_event(ready)
                expert_wgrads()
                side_done = torch.cuda.Event()
                side_done.record()
        else:
            with self._timed("expert_wgrad"):
                expert_wgrads()
            if self.profile and "_relaunch" in c:
                c["_relaunch"]["expert_wgrad"] = expert_wgrads        # (accumulates into the gradient buffer again: timing only)
        if side_small is not None and side_done is None:      # (no expert weight gradients behind it on the side stream: join here)
            torch.cuda.current_stream().wait_event(side_small)
        return dict(c=c, d_laux=d_laux, dgmax=dgmax, dx=dx, dout=dout, returns=returns if ep is not None else None, tail_jobs=tail_jobs,
                    nsp=nsp, side_done=side_done, keep=(dc_ray, dh2, dsig))      # (keep: tensors the side stream reads stay allocated to the join)

    def backward_net_b(self, st):
        """Second half of backward_net: router backward, front backward chain, dense weight gradients (+ the hash table's)."""
        o, dt = ops, self.dtype
        c, d_laux, dgmax, dx, returns, tail_jobs, nsp, side_done = (st[k] for k in ("c", "d_laux", "dgmax", "dx", "returns", "tail_jobs",
                                                                                   "nsp", "side_done"))
        P, seg_tokens = c["P"], c["seg_tokens"]
        M, E, G = self.M, self.E, self.G
        g, ep = self.g, self.ep
        _b = lambda name, shape, dtype: self._buf(c["tag"] + ":" + name, shape, dtype)
        # gate backward (softmax / router / LayerNorm) including the l_aux term
        coef = (d_laux * (E / float(seg_tokens * seg_tokens))).to(torch.float32).contiguous()
        dg = o.gate_bwd(c["g"], self.p["ln.w"], self.p["ln.b"], self.p["wg"], c["gates"], c["idx"], dgmax, c["stats"],
                        c["counts"], coef, seg_tokens, g["wg"], g["ln.w"], g["ln.b"])
        # front backward chain: dg -> d(a1) -> d(h0), adding the expert path's input gradient through tok2row
        dza1 = _b("dza1", (P, G), dt)
        dh0 = _b("dh0", (P, M), dt)
        if ep is not None:      # (the input gradients travelled home under the expert weight gradients / the router backward)
            for wait in returns:
                wait()
        o.mlp_chain(dg, [o.Layer(self.wb["gate1"], None, relu=2, mask=c["m_a1"], save=dza1), o.Layer(self.wb["gate0"], None)],
                    dh0, y_add=dx, y_add_gather=c["row_of_tok"], tag=6, geometry=c["front_geom"] if c["front_geom"] >= 6 else 0)
        self._dense_wgrads(tail_jobs + [(c["a1"], dg, g["gate1.w"].view(1, G, G), g["gate1.b"].view(1, G)),
                                        (c["h0"], dza1, g["gate0.w"].view(1, M, G), g["gate0.b"].view(1, G)),
                                        (c["pe"], dh0, g["xyz.w"].view(1, self.KP, M), g["xyz.b"].view(1, M))], nsp)
        if self.hash is not None:          # dL/d encoding = dh0 W_xyz^T, scattered into the hash table's gradient
            d_enc = c.get("d_enc_out")     # (a part of a ragged batch: its rows of the batch's buffer - backward_net scatters them)
            part = d_enc is not None
            if not part:
                d_enc = _b("d_enc", (P, self.KP), dt)
            o.mlp_chain(dh0, [o.Layer(self.wb["xyz"], None)], d_enc, tag=0)
            if not part:
                o.hash_encode_bwd(c["rays"], c["z"], d_enc, self.hash, g["hash.table"])
        if side_done is not None:
            torch.cuda.current_stream().wait_event(side_done)

    def _dense_wgrads(self, jobs, n_splits):
        """Weight gradients of dense layers that are ready together, (a, dz, dw, db) each: ONE balanced launch and one reduction
        (swn_wgrad_multi) for layers of up to 256 features, per-layer block launches for wider ones."""
        if all(a.shape[1] <= 256 and b.shape[1] <= 256 for a, b, _w, _b in jobs):
            ops.wgrad_multi([(a, b, dw, db, None, None) for a, b, dw, db in jobs])
        else:
            for a, b, dw, db in jobs:
                ops.wgrad(a, b, dw, db, n_splits=n_splits)

    # ------------------------------------------------------------------------------------------ training step
    def train_step(self, rgbs, rays, image_indices, n_samples, seg_tokens, perturb=1.0, perturb_rand=None,
                   sigma_noise=None, optimizer_step=True, routing_override=None, grad_allreduce=None, fine_samples=0,
                   fine_u=None, sigma_noise_fine=None):
        """Runner._training_step + loss assembly + backward + Adam (runner.py:1077-1123, 646-686) = grad_step + apply_step."""
        res = self.grad_step(rgbs, rays, image_indices, n_samples, seg_tokens, perturb, perturb_rand, sigma_noise, routing_override,
                             fine_samples, fine_u, sigma_noise_fine)
        self.apply_step(grad_allreduce, optimizer_step)
        return res

    def grad_step(self, rgbs, rays, image_indices, n_samples, seg_tokens, perturb=1.0, perturb_rand=None, sigma_noise=None,
                  routing_override=None, fine_samples=0, fine_u=None, sigma_noise_fine=None, split=False):
        """Forward + loss + backward of one training step: fills self.grad (zeroed first) and returns the metrics.  Nothing here
        depends on host state that changes from step to step, so the whole launch sequence can be captured into a hipGraph
        (graph.GraphedTrainStep).  split=True (plain step only) stops after backward_net_a - the expert block of the gradient is final,
        the launch stream has waited for it - and returns the second half's state as res["bwd_b"] (finish with backward_net_b).
        fine_samples > 0 adds the hierarchical pass (rendering.py:236-268): importance-sample fine depths from the coarse
        weights (detached), evaluate the network on them, sort-merge with the coarse samples, composite the union;
        loss = mse(rgb_fine) + wt * (mean(gate_loss_fine) + mean(gate_loss_coarse)) / 2."""
        N = rays.shape[0]
        self.grad.zero_()
        fine = fine_samples > 0
        if not fine:
            c = out = self.forward_rays(rays, image_indices, n_samples, seg_tokens, perturb, perturb_rand, sigma_noise, True,
                                        routing_override)
        else:
            c, cf, out = self.forward_hier(rays, image_indices, n_samples, fine_samples, seg_tokens, perturb, perturb_rand,
                                           fine_u, sigma_noise, sigma_noise_fine, routing_override)
        # loss (F.mse_loss runner.py:1099, + wt * gate loss :646-651, :1104-1111), psnr and the gradient seeds in ONE launch
        # (swn_step_loss); scaler.scale(loss).backward() (runner.py:679): the scale is read from a DEVICE scalar (kept equal to
        # loss_scaler.scale by _unscale_ok), so a captured step (graph.GraphedTrainStep) follows the scale as it adapts between replays
        ls = self._loss_scale_tensor() if self.loss_scaler is not None else None
        out4, d_rgb, d_la, d_lb = ops.step_loss(out["rgb"], rgbs.to(torch.float32).contiguous(), cf["l_aux"] if fine else c["l_aux"],
                                                c["l_aux"] if fine else None, self.wt, ls)
        photo, gate_loss, loss, psnr = out4[0], out4[1], out4[2], out4[3]
        bwd_b = None
        if split:
            if fine or c.get("parts") is not None or type(self).backward_net is not SwitchNeRF.backward_net:
                raise ValueError("grad_step(split=True): the plain single-context step of a SwitchNeRF only")
            bwd_b = self.backward_net_a(c, ops.composite_bwd(c["raw"], c["z"], d_rgb), d_la, join_side=True)
        elif not fine:
            self.backward(c, d_rgb, d_la)
        else:
            d_raw_m = ops.composite_bwd(out["raw"], out["z"], d_rgb)
            d_raw_f, d_raw_c = ops.unmerge_grad(d_raw_m, out["order"], fine_samples, n_samples)
            self.backward_net(cf, d_raw_f, d_la)
            self.backward_net(c, d_raw_c, d_lb)
        res = dict(loss=loss, photo_loss=photo, gate_loss=gate_loss, psnr=psnr,
                   depth_variance=out["depth_variance"].mean(), ctx=c, rgb=out["rgb"], depth=out["depth"])
        if fine:
            res["ctx_fine"] = cf
        if split:
            res["bwd_b"] = bwd_b
        return res

    def apply_step(self, grad_allreduce=None, optimizer_step=True, refresh=None):
        """Gradient all-reduce (N > 1) + Adam (runner.py:486, 686) + refresh of the compute copies of the weights (`refresh`: a
        replacement for refresh_compute_copies, e.g. the replay of its captured graph)."""
        scale = 1.0
        if grad_allreduce is not None:
            scale = grad_allreduce(self._allreduce_view())
        if optimizer_step and self._unscale_ok():
            if self.loss_scaler is not None:
                scale /= self._applied_loss_scale
            self.step_count += 1
            ops.adam_step(self.flat, self.grad, self.m, self.v, None, self.step_count, self.lr, grad_scale=scale)
            (refresh or self.refresh_compute_copies)()

    _ls_dev = None
    _ls_dev_val = None

    def _loss_scale_tensor(self):
        """The current loss scale as a device scalar (see grad_step), refreshed whenever the host value has changed since the last
        call (after _unscale_ok adapted it, after a checkpoint load, after a caller set loss_scaler.scale)."""
        v = float(self.loss_scaler.scale)
        if self._ls_dev is None:
            self._ls_dev = torch.full((1,), v, dtype=torch.float32, device=self.dev)
        elif self._ls_dev_val != v:
            self._ls_dev.fill_(v)
        self._ls_dev_val = v
        return self._ls_dev

    def _found_inf(self) -> bool:
        """GradScaler's found_inf for THIS model's gradient (one device-to-host flag, like torch's): records the scale the backward
        used (`_applied_loss_scale`, what the Adam step divides by).  Under expert parallelism the expert part of the gradient is
        rank-local (never all-reduced), so the flag is agreed over the ranks (MAX): every rank skips - or takes - the same steps and
        holds the same scale."""
        self._applied_loss_scale = self._ls_dev_val if self._ls_dev_val is not None else self.loss_scaler.scale   # what the backward used
        bad = (~torch.isfinite(self.grad).all()).to(torch.float32).view(1)
        if self.ep is not None and not self.ep.local:
            import torch.distributed as dist
            dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=self.ep.group)
        return bool(bad.item() > 0)

    def _unscale_ok(self) -> bool:
        """GradScaler.step / update (runner.py:686-690): with loss scaling on, skip the optimizer step when a gradient is not finite
        and adapt the scale.  Always True without loss scaling."""
        if self.loss_scaler is None:
            return True
        found_inf = self._found_inf()
        self.loss_scaler.update(found_inf)
        self._loss_scale_tensor()           # the device copy follows (a captured step reads it on its next replay)
        return not found_inf

    def _allreduce_view(self):
        """What the data-parallel all-reduce sums: the whole flat gradient, or - experts sharded over the ranks - its dense
        prefix (a local expert's gradient already holds the contributions of every rank's rows)."""
        if self.ep is not None and not self.ep.local:
            return self.grad[: self.n_dense]
        return self.grad

    def forward_hier(self, rays, image_indices, n_samples, fine_samples, seg_tokens, perturb=0.0, perturb_rand=None,
                     fine_u=None, sigma_noise=None, sigma_noise_fine=None, routing_override=None, no_batch=False, training=True):
        """_get_results with fine_samples > 0 and no cascade (rendering.py:199-274): coarse pass (weights only, its raw
        outputs kept), importance sampling of the fine depths from the detached coarse weights, fine pass on those
        depths, sort-merge of both sample sets (:419-433) and compositing of the union.
        Returns (coarse ctx, fine ctx, merged results {raw, z, order, rgb, depth, depth_variance})."""
        N = rays.shape[0]
        c = self.forward_rays(rays, image_indices, n_samples, seg_tokens, perturb, perturb_rand, sigma_noise, training,
                              routing_override, no_batch=no_batch, want_weights=True)
        if fine_u is None:                                                # det = (perturb == 0): linspace, else rand (:605-609)
            fine_u = (self._linspace(fine_samples).expand(N, fine_samples).contiguous() if perturb == 0
                      else torch.rand(N, fine_samples, device=self.dev))
        z_fine = ops.sample_pdf(c["z"], c["weights"], fine_u, fine_samples)
        seg_f = min(seg_tokens, N * fine_samples)
        cf = self.forward_rays(rays, image_indices, fine_samples, seg_f, 0.0, None, sigma_noise_fine, training, None,
                               no_batch=no_batch, z_in=z_fine, pe_dir=c["pe_dir"], tag="f", composite=False)
        zm, order, raw_m = ops.merge_samples(z_fine, c["z"], cf["raw"], c["raw"])
        out = dict(raw=raw_m, z=zm, order=order, z_fine=z_fine)
        out["rgb"], out["depth"], out["depth_variance"], _ = ops.composite_fwd(raw_m, zm)
        return c, cf, out

    # ------------------------------------------------------------------------------------------ mip path
    def forward_level_mip(self, rays, radii, image_indices, z, seg_tokens, sigma_noise=None, no_batch=False, tag="c",
                          want_weights=False, rgb_padding=0.001, pe_dir=None, training=True):
        """One level of rendering_mip._get_results (rendering_mip.py:195-215 / :236-253): the S edges `z` of every ray give
        S - 1 conical frustums; integrated positional encoding (swn_mip_encode) -> the same network (MipNeRFMoE.forward is
        NeRFMoE.forward behind MipEmbedder, nerf_moe.py:675-810) -> compositing at the frustum mid points with the colour
        padding of :383-386."""
        N, S1 = rays.shape[0], z.shape[1] - 1
        self._sync_compute_copies()
        pe = ops.mip_encode(rays, radii, z, self.cfg["pos_xyz_dim"], self.dtype, self.KP)
        if pe_dir is None:
            pe_dir = self._dir_pe(rays)
        seg = min(seg_tokens, N * S1)
        self._saving = bool(training)      # inference: no activation saves / ReLU masks (like forward_rays)
        try:
            c = self._net_forward(pe, pe_dir, image_indices, N, S1, seg, sigma_noise, None, no_batch, tag)
        finally:
            self._saving = True
        c["z_edges"] = z
        c["z"] = (0.5 * (z[:, 1:] + z[:, :-1])).contiguous()
        c["rgb_padding"] = float(rgb_padding)
        c["rgb"], c["depth"], c["depth_variance"], c["weights"] = ops.composite_fwd(c["raw"], c["z"], want_weights=want_weights,
                                                                                    rgb_padding=rgb_padding)
        return c

    def forward_mip(self, rays, radii, image_indices, n_samples, n_fine, seg_tokens, perturb=0.0, perturb_rand=None, fine_u=None,
                    sigma_noise=None, sigma_noise_fine=None, no_batch=False, rgb_padding=0.001, resample_padding=0.01,
                    training=True, fine_randomized=None):
        """rendering_mip.render_rays (rendering_mip.py:133-172): coarse level on n_samples edges, then (n_fine > 0) the fine
        level on n_fine edges resampled from the blurred coarse weights (stop_level_grad: no gradient through the edges).
        fine_u: the U[0,1) tensor [N, n_fine] of sorted_piecewise_constant_pdf1's randomized branch (drawn here if perturb > 0
        and none is given); perturb = 0 -> deterministic.  fine_randomized (default: perturb > 0) decouples the fine level's
        randomisation from the coarse jitter: the reference resamples with randomized=hparams.perturb even in eval mode
        (rendering_mip.py:227) while its coarse jitter is off there (:147).  training=False skips the activation saves."""
        N = rays.shape[0]
        if fine_randomized is None:
            fine_randomized = perturb > 0
        t_steps = self._linspace(n_samples)
        radii = radii.reshape(-1).contiguous()
        z = ops.sample_z(rays, t_steps, perturb_rand, perturb, n_samples)
        c = self.forward_level_mip(rays, radii, image_indices, z, seg_tokens, sigma_noise, no_batch, "c", n_fine > 0, rgb_padding,
                                   training=training)
        if n_fine <= 0:
            return c, None
        if fine_u is None and fine_randomized:
            fine_u = torch.rand(N, n_fine, device=self.dev)
        z_f = ops.mip_resample(z, c["weights"], fine_u if fine_randomized else None, n_fine, resample_padding)
        cf = self.forward_level_mip(rays, radii, image_indices, z_f, seg_tokens, sigma_noise_fine, no_batch, "f", False, rgb_padding,
                                    pe_dir=c["pe_dir"], training=training)
        return c, cf

    def train_step_mip(self, rgbs, rays, radii, image_indices, n_samples, n_fine, seg_tokens, perturb=1.0, perturb_rand=None,
                       fine_u=None, sigma_noise=None, sigma_noise_fine=None, optimizer_step=True, grad_allreduce=None,
                       rgb_padding=0.001, resample_padding=0.01):
        """Runner._training_step_mip (runner.py:1126-1167): loss = (mse(rgb_fine) + mse(rgb_coarse)) / 2
        + wt * (mean(gate_loss_fine) + mean(gate_loss_coarse)) / 2, backward through both levels, Adam."""
        self.grad.zero_()
        c, cf = self.forward_mip(rays, radii, image_indices, n_samples, n_fine, seg_tokens, perturb, perturb_rand, fine_u,
                                 sigma_noise, sigma_noise_fine, False, rgb_padding, resample_padding)
        levels = [c] if cf is None else [cf, c]
        share = 1.0 / len(levels)
        photo, gate_loss = 0.0, 0.0
        ls = self._loss_scale_tensor() if self.loss_scaler is not None else 1.0
        for lv in levels:
            diff = lv["rgb"] - rgbs
            lv["_d_rgb"] = (diff * (2.0 * share / diff.numel()) * ls).contiguous()
            photo = photo + share * (diff * diff).mean()
            gate_loss = gate_loss + share * lv["l_aux"].mean()
        loss = photo + self.wt * gate_loss
        for lv in levels:
            d_raw = ops.composite_bwd(lv["raw"], lv["z"], lv["_d_rgb"], rgb_padding=lv["rgb_padding"])
            d_laux = torch.full((lv["n_seg"],), share * self.wt / lv["n_seg"], dtype=torch.float32, device=self.dev) * ls
            self.backward_net(lv, d_raw, d_laux)
        self.apply_step(grad_allreduce, optimizer_step)
        top = levels[0]
        return dict(loss=loss, photo_loss=photo, gate_loss=gate_loss, psnr=-10.0 * torch.log10(((top["rgb"] - rgbs) ** 2).mean()),
                    depth_variance=top["depth_variance"].mean(), ctx=c, ctx_fine=cf, rgb=top["rgb"], depth=top["depth"])

    def set_iteration(self, iteration: int, lr_decay_factor: float = 0.1, train_iterations: int = 500000):
        """Learning-rate schedule hook = the reference's ExponentialLR(gamma = lr_decay_factor ** (1 / train_iterations)) stepped
        once per iteration (runner.py:505-512, 688-693): the rate the step of `iteration` (0-based) uses."""
        from . import checkpoint
        self.lr = checkpoint.exponential_lr(self.base_lr, iteration, lr_decay_factor, train_iterations)
        return self.lr

    # ------------------------------------------------------------------------------------------ NeRFMoE mirrors
    training = True
    moe_no_batch = False

    def train(self, mode=True):
        self.training = bool(mode)
        return self

    def eval(self):
        return self.train(False)

    def modules(self):
        """nn.Module.modules() stand-in: Runner.set_no_batch (runner.py:946-951) walks it looking for `moe_no_batch`."""
        yield self

    def named_parameters(self):
        """(reference key, tensor) pairs - the parameters in the reference's key layout, for inspection / count_parameters
        (runner.py:196-199).  The trainable leaf an optimizer should be given is `flat_param` (trainable_parameters())."""
        return iter(self._to_ref_layout(self.p).items())

    def parameters(self):
        return (v for _, v in self.named_parameters())

    # ---- torch.autograd bridge (autograd.py): the reference's Runner loop drives the HIP path unchanged
    _flat_param = None
    _packed_version = -1

    @property
    def flat_param(self):
        """The model's whole parameter set as ONE leaf nn.Parameter sharing storage with the flat fp32 master buffer.  Give it to
        torch.optim.Adam / DDP; rendering.render_rays(...) in training mode returns tensors whose backward fills its `.grad`."""
        if self._flat_param is None:
            self._flat_param = torch.nn.Parameter(self.flat, requires_grad=True)
            self._packed_version = self._flat_param._version
        return self._flat_param

    def trainable_parameters(self):
        return [self.flat_param]

    def _sync_compute_copies(self):
        """An optimizer outside this class moved the master weights in place (the Parameter's version counter tells): refresh the
        packed compute copies the chain kernels read."""
        fp = self._flat_param
        if fp is not None and fp._version != self._packed_version:
            self.refresh_compute_copies()
            self._packed_version = fp._version

    def to(self, *_a, **_k):
        return self

    def set_no_batch(self, mode=True):
        """NeRFMoE.set_no_batch (models/nerf_moe.py:315-318): eval path without capacity / token dropping."""
        self.moe_no_batch = bool(mode)

    def __call__(self, x, sigma_only=False, sigma_noise=None):
        """NeRFMoE.forward (models/nerf_moe.py:320-455): x [P, 7] = xyz(3), dir(3), image index(1) ->
        {"outputs": [P,4] (rgb, sigma), "extras": {"moe_loss": [1], "moe_gates": [[P,1]]}}.  Routing (capacity, ranking,
        l_aux) is over the P points of this call, exactly like one model chunk of the reference.  Inference only."""
        expected = 7
        if x.shape[1] != expected:
            raise Exception("Unexpected input shape: {} (expected: {}, xyz_dim: {})".format(x.shape, expected, 3))
        P = x.shape[0]
        xf = x.to(torch.float32)
        rays = torch.cat([xf[:, :6], torch.zeros(P, 2, device=self.dev)], 1).contiguous()   # o = xyz, z = 0 -> sample = xyz
        c = self.forward_rays(rays, xf[:, 6].long().contiguous(), 1, P, 0.0, None,
                              None if sigma_noise is None else sigma_noise.reshape(-1).to(torch.float32).contiguous(),
                              training=self.training, no_batch=self.moe_no_batch)
        return {"outputs": c["raw"], "extras": {"moe_loss": c["l_aux"], "moe_gates": [c["idx"].long().view(P, 1)]}}
