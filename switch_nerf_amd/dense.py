"""The reference's dense (non-MoE) NeRF on the same HIP kernels: BASELINE.json configs[0] (`--no-use_moe`), and the network
the reference also uses as its background model.

Mirrors /root/reference/switch_nerf/models/nerf.py:60-190 (class NeRF; built by models/model_utils.py:90-120 get_nerf with
use_moe = False): `layers` x (Linear + ReLU) of width `layer_dim`, the encoded position concatenated again in front of the
layers named in `skip_layers` (torch.cat([enc, h]), nerf.py:155-156), the sigma head, then the same direction /
appearance tail as NeRFMoE (xyz_encoding_final -> dir_a_encoding + ReLU -> rgb + sigmoid).

DenseNeRF derives from SwitchNeRF: ray sampling, positional encoding, the tail chain, the heads, compositing, the loss
assembly and Adam are the same launches; only the trunk differs - ONE MLP chain launch instead of gate + routing + experts.
The concat-skip is not materialised: Linear(cat([enc, h])) = h W_h + enc W_enc runs as two consecutive K loops on the same
accumulators (chain layer mode skip = 2: the h half, then the chain input re-staged in the LDS tile for the enc half, which
carries the bias / ReLU / mask / save).  The backward is a plain chain through W_h^T (the encoding has no gradient) and the
weight gradient of the skip layer is two GEMMs (h^T dZ, enc^T dZ).

The state_dict layout is the reference NeRF's (xyz_encodings.{i}.0.weight, xyz_encoding_final.weight, dir_a_encoding.0.*,
sigma.*, rgb.*, embedding_a.weight), so checkpoints interchange.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import ops
from .model import SwitchNeRF, _ceil_to

# /root/reference/switch_nerf/opts.py defaults for the non-MoE model (layer_dim 256, 8 layers, skip at 4, appearance 48)
DENSE = dict(layer_dim=256, layers=8, skip_layers=(4,), pos_xyz_dim=12, pos_dir_dim=4, appearance_dim=48,
             appearance_count=1920, xyz_dim=3)


class DenseNeRF(SwitchNeRF):
    def __init__(self, cfg: dict = DENSE, dtype=torch.bfloat16, device="cuda", lr=5e-4, seed=0):
        super().__init__(cfg, dtype, device, capacity_factor=1.0, batch_prioritized=False, moe_l_aux_wt=0.0, lr=lr, seed=seed)

    def _configure(self, cfg):
        W, L, xd = cfg["layer_dim"], cfg["layers"], cfg.get("xyz_dim", 3)
        skips = tuple(cfg["skip_layers"])
        assert len(skips) <= 1 and all(0 < s < L for s in skips), "at most one concat-skip layer, not the first"
        assert W % 64 == 0 and W <= 256, "layer_dim: a multiple of 64 up to 256"
        self.xyz_dim = xd
        self.in_xyz = xd + 2 * xd * cfg["pos_xyz_dim"]
        self.in_dir = 3 + 6 * cfg["pos_dir_dim"]
        self.KP = _ceil_to(self.in_xyz, 64)
        self.DP = _ceil_to(self.in_dir, 8)
        self.n_ray_feat = self.in_dir + cfg["appearance_dim"]
        self.skip_l = skips[0] if skips else None
        M, H2 = W, W // 2
        self.L, self.M, self.E, self.G, self.H2 = L, M, 1, M, H2
        spec = []
        for i in range(L):
            if i == self.skip_l:          # Linear(cat([enc, h])): the enc rows and the h rows as two matrices
                spec += [(f"enc{i}p.w", (self.KP, W)), (f"enc{i}h.w", (W, W)), (f"enc{i}.b", (W,))]
            else:
                spec += [(f"enc{i}.w", (self.KP if i == 0 else W, W)), (f"enc{i}.b", (W,))]
        spec += [("l1.w", (M, M)), ("l1.b", (M,)), ("l2h.w", (M, H2)), ("l2r.w", (self.n_ray_feat, H2)), ("l2.b", (H2,)),
                 ("sigma.w", (M,)), ("sigma.b", (1,)), ("color.w", (3, H2)), ("color.b", (3,)),
                 ("emb", (cfg["appearance_count"], cfg["appearance_dim"]))]
        s = self.skip_l
        self._chain_weights = [f"enc{i}" for i in range(L) if i != s] + ([f"enc{s}p", f"enc{s}h"] if s is not None else []) + ["l1", "l2h"]
        self._fwd_only_weights = {"enc0"} | ({f"enc{s}p"} if s is not None else set())
        return spec

    # ------------------------------------------------------------------------------------------ parameters
    def _init_random(self, seed):
        g = torch.Generator().manual_seed(seed)
        cfg, W = self.cfg, self.M

        def lin(out_f, in_f):
            b = 1.0 / math.sqrt(in_f)
            return ((torch.rand(out_f, in_f, generator=g) * 2 - 1) * b), ((torch.rand(out_f, generator=g) * 2 - 1) * b)
        sd = {}
        for i in range(self.L):
            k = self.in_xyz if i == 0 else (W + self.in_xyz if i == self.skip_l else W)
            sd[f"xyz_encodings.{i}.0.weight"], sd[f"xyz_encodings.{i}.0.bias"] = lin(W, k)
        sd["xyz_encoding_final.weight"], sd["xyz_encoding_final.bias"] = lin(W, W)
        sd["dir_a_encoding.0.weight"], sd["dir_a_encoding.0.bias"] = lin(self.H2, W + self.n_ray_feat)
        sd["sigma.weight"], sd["sigma.bias"] = lin(1, W)
        sd["rgb.weight"], sd["rgb.bias"] = lin(3, self.H2)
        sd["embedding_a.weight"] = torch.randn(cfg["appearance_count"], cfg["appearance_dim"], generator=g)
        self.load_state_dict(sd)

    def load_state_dict(self, sd):
        """Accepts the reference NeRF's state_dict (optionally with the DDP wrapper's `module.` prefix)."""
        from . import checkpoint
        self._load_ref_layout(checkpoint.strip_module_prefix(sd), self.p)
        self.refresh_compute_copies()

    def _load_ref_layout(self, sd, p):
        def t(k):
            v = sd[k]
            v = torch.from_numpy(np.asarray(v)) if not torch.is_tensor(v) else v
            return v.detach().to(torch.float32).to(self.dev)
        W, nx = self.M, self.in_xyz
        with torch.no_grad():
            for i in range(self.L):
                w = t(f"xyz_encodings.{i}.0.weight").t()            # [in, out]
                if i == self.skip_l:                                 # input rows: [enc (nx) | h (W)]  (torch.cat([enc, h]))
                    p[f"enc{i}p.w"].zero_()
                    p[f"enc{i}p.w"][:nx] = w[:nx]
                    p[f"enc{i}h.w"].copy_(w[nx:])
                else:
                    p[f"enc{i}.w"].zero_()
                    p[f"enc{i}.w"][: w.shape[0]] = w
                p[f"enc{i}.b"].copy_(t(f"xyz_encodings.{i}.0.bias"))
            p["l1.w"].copy_(t("xyz_encoding_final.weight").t())
            p["l1.b"].copy_(t("xyz_encoding_final.bias"))
            w2 = t("dir_a_encoding.0.weight")
            p["l2h.w"].copy_(w2[:, :W].t())
            p["l2r.w"].copy_(w2[:, W:].t())
            p["l2.b"].copy_(t("dir_a_encoding.0.bias"))
            p["sigma.w"].copy_(t("sigma.weight").view(-1))
            p["sigma.b"].copy_(t("sigma.bias"))
            p["color.w"].copy_(t("rgb.weight"))
            p["color.b"].copy_(t("rgb.bias"))
            p["emb"].copy_(t("embedding_a.weight"))

    def _to_ref_layout(self, d):
        W, nx = self.M, self.in_xyz
        out = {}
        for i in range(self.L):
            if i == self.skip_l:
                w = torch.cat([d[f"enc{i}p.w"][:nx], d[f"enc{i}h.w"]], 0)
            else:
                w = d[f"enc{i}.w"]
                if i == 0:
                    w = w[:nx]
            out[f"xyz_encodings.{i}.0.weight"] = w.t().contiguous()
            out[f"xyz_encodings.{i}.0.bias"] = d[f"enc{i}.b"].clone()
        out["xyz_encoding_final.weight"] = d["l1.w"].t().contiguous()
        out["xyz_encoding_final.bias"] = d["l1.b"].clone()
        out["dir_a_encoding.0.weight"] = torch.cat([d["l2h.w"].t(), d["l2r.w"].t()], 1).contiguous()
        out["dir_a_encoding.0.bias"] = d["l2.b"].clone()
        out["sigma.weight"] = d["sigma.w"].view(1, -1).clone()
        out["sigma.bias"] = d["sigma.b"].clone()
        out["rgb.weight"] = d["color.w"].clone()
        out["rgb.bias"] = d["color.b"].clone()
        out["embedding_a.weight"] = d["emb"].clone()
        return out

    def state_dict(self, layout="nerf", prefix=""):
        return {prefix + k: v for k, v in self._to_ref_layout(self.p).items()}

    def set_expert_parallel(self, ep):
        raise NotImplementedError("the dense NeRF has no experts to shard")

    # ------------------------------------------------------------------------------------------ forward
    def _net_forward(self, pe, pe_dir, image_indices, N, S, seg_tokens, sigma_noise, routing_override, no_batch, tag):
        """NeRF.forward (nerf.py:143-190) over the N*S points whose encodings are in `pe` -> c["raw"] [N*S, 4]."""
        o, dt = ops, self.dtype
        P = N * S
        W, L, H2, s = self.M, self.L, self.H2, self.skip_l
        c = dict(N=N, S=S, P=P, n_seg=max(1, P // seg_tokens), seg_tokens=seg_tokens, tag=tag, image_indices=image_indices)
        c["pe"], c["pe_dir"] = pe, pe_dir
        _b = lambda name, shape, dtype: self._buf(tag + ":" + name, shape, dtype)
        c["acts"] = [_b(f"act{i}", (P, W), dt) for i in range(L)]          # post-ReLU outputs; acts[L-1] = xyz_ (c["y"])
        mw = o.chain_mask_words(dt, 1, P, max(W, self.KP))
        c["masks"] = [_b(f"mask{i}", (mw,), torch.int32) for i in range(L)]
        sv = self._saving
        c["no_grad"] = not sv
        layers = []
        for i in range(L):
            last = i == L - 1
            kw = dict(relu=1, mask=c["masks"][i] if sv else None, save=c["acts"][i] if (sv and not last) else None)
            bias = self.p[f"enc{i}.b"].view(1, W)
            if i == s:      # h W_h (+ nothing), then enc W_enc + b -> ReLU: two K loops on the same accumulators
                layers += [o.Layer(self.wf[f"enc{i}h"], None, skip=2), o.Layer(self.wf[f"enc{i}p"], bias, **kw)]
            else:
                layers.append(o.Layer(self.wf[f"enc{i}"], bias, **kw))
        with self._timed("trunk_fwd"):
            o.mlp_chain(pe, layers, c["acts"][L - 1], tag=1)
        c["y"] = c["acts"][L - 1]
        # ---- per-ray part of dir_a_encoding: [PE(dir), appearance embedding] @ W2r + b2 (nerf.py:173-181)
        c["ray_feat"], c["c_ray"] = o.ray_feat_fwd(pe_dir, self.in_dir, self.p["emb"], image_indices.contiguous(), self.p["l2r.w"], self.p["l2.b"])
        from .model import c_esz
        fused = self.sw["fused_heads"] and W in (256, 512) and H2 in (128, 256) and W * c_esz(dt) <= 1024      # heads inside the tail chain's launch (swn.h: heads_raw)
        c["h1"] = _b("h1", (P, W), dt) if sv else None
        c["h2"] = _b("h2", (P, H2), dt) if (sv or not fused) else None      # an inference forward writes nothing but raw
        c["raw"] = torch.empty(P, 4, dtype=torch.float32, device=self.dev) if fused else None
        o.mlp_chain(c["y"], [o.Layer(self.wf["l1"], self.p["l1.b"].view(1, W), save=c["h1"] if sv else None),
                             o.Layer(self.wf["l2h"], None, relu=1, rowbias=c["c_ray"], rows_per_bias=S)], c["h2"], tag=4, group_stride=P,
                    heads=(self.p["sigma.w"], self.p["sigma.b"], self.p["color.w"], self.p["color.b"], sigma_noise, c["raw"]) if fused else None)
        if not fused:
            c["raw"] = o.heads_fwd(c["y"], c["h2"], self.p["sigma.w"], self.p["sigma.b"], self.p["color.w"], self.p["color.b"],
                                   sigma_noise)
        c["l_aux"] = torch.zeros(c["n_seg"], dtype=torch.float32, device=self.dev)     # no gate loss (runner.py:1104 guards on use_moe)
        c["idx"] = None
        return c

    # ------------------------------------------------------------------------------------------ backward
    def backward_net(self, c, d_raw, d_laux=None):
        o, dt = ops, self.dtype
        if c.get("no_grad"):
            raise RuntimeError("this context comes from an inference forward (training=False): nothing was saved for the backward")
        S, P = c["S"], c["P"]
        W, L, H2, s = self.M, self.L, self.H2, self.skip_l
        g, acts, masks = self.g, c["acts"], c["masks"]
        _b = lambda name, shape, dtype: self._buf(c["tag"] + ":" + name, shape, dtype)
        dh2, dsig, dc_ray = o.heads_bwd(c["y"], c["h2"], self.p["color.w"], c["raw"], d_raw, g["sigma.w"], g["sigma.b"], g["color.w"],
                                        g["color.b"], rows_per_group=S)
        if dc_ray.shape[1] in (64, 128, 256) and c["ray_feat"].shape[1] <= 256:      # split over the rays + ordered reduce (one launch)
            o.ray_feat_wgrad(c["ray_feat"], dc_ray, g["l2r.w"], g["l2.b"])
        else:
            g["l2r.w"].addmm_(c["ray_feat"].t(), dc_ray)
            g["l2.b"].add_(dc_ray.sum(0))
        o.emb_grad(dc_ray @ self.p["l2r.w"][self.in_dir:].t(), c["image_indices"].contiguous(), g["emb"])
        dh1 = _b("dh1", (P, W), dt)
        nsp = max(1, min(256, P // 1024))
        # d(pre-activation of the last trunk layer) = (dy + dsigma * w_sigma) * (xyz_ > 0): the combine backward with a unit gate
        if getattr(self, "_ones", None) is None or self._ones.numel() < P:
            self._ones = torch.ones(P, dtype=torch.float32, device=self.dev)
        ones = self._ones[:P]
        dz = [_b(f"dz{i}", (P, W), dt) for i in range(L - 1)]
        if W * dh1.element_size() <= 1024:      # ... applied in the write-out of the tail backward chain (swn.h comb_*: swn_combine_bwd's
            dz.append(_b(f"dz{L - 1}", (P, W), dt))                      # arithmetic, value for value): dy never reaches memory
            o.mlp_chain(dh2, [o.Layer(self.wb["l2h"], None, save=dh1), o.Layer(self.wb["l1"], None)], dz[L - 1], tag=5,
                        combine=(c["y"], dsig, self.p["sigma.w"], ones, _b("dgate_unused", (P,), torch.float32)))
        else:
            dy = _b("dy", (P, W), dt)
            o.mlp_chain(dh2, [o.Layer(self.wb["l2h"], None, save=dh1), o.Layer(self.wb["l1"], None)], dy, tag=5)
            dz.append(o.combine_bwd(dy, c["y"], dsig, self.p["sigma.w"], ones)[0])
        o.wgrad(c["h1"], dh2, g["l2h.w"].view(1, W, H2), None, n_splits=nsp)
        o.wgrad(c["y"], dh1, g["l1.w"].view(1, W, W), g["l1.b"].view(1, W), n_splits=nsp)
        with self._timed("trunk_bwd"):
            # dz[L-1] -> ... -> dz[0]: the encoding carries no gradient, so the skip layer is just W_h^T here
            if L > 1:
                bl = [o.Layer(self.wb[f"enc{i}h" if i == s else f"enc{i}"], None, relu=2, mask=masks[i - 1], save=dz[i - 1] if i > 1 else None)
                      for i in range(L - 1, 0, -1)]
                o.mlp_chain(dz[L - 1], bl, dz[0], tag=2)
        with self._timed("trunk_wgrad"):
            same = [(acts[i - 1], dz[i], g[f"enc{i}h.w" if i == s else f"enc{i}.w"].view(1, W, W), g[f"enc{i}.b"].view(1, W), None, None)
                    for i in range(1, L)]
            for i0 in range(0, len(same), 8):          # the W x W layers in one launch per 8
                o.wgrad_batched(same[i0:i0 + 8], n_splits=nsp)
            o.wgrad(c["pe"], dz[0], g["enc0.w"].view(1, self.KP, W), g["enc0.b"].view(1, W), n_splits=nsp)
            if s is not None:
                o.wgrad(c["pe"], dz[s], g[f"enc{s}p.w"].view(1, self.KP, W), None, n_splits=nsp)

    # ------------------------------------------------------------------------------------------ NeRF mirrors
    def set_no_batch(self, mode=True):
        pass

    def __call__(self, x, sigma_only=False, sigma_noise=None):
        """NeRF.forward (nerf.py:143-190): x [P, xyz_dim + 3 + 1] = position, direction, image index -> [P, 4] (rgb, sigma), or
        x [P, xyz_dim] with sigma_only -> [P, 1].  Inference only.  xyz_dim 4 (the background model called on explicit
        inverted-sphere points): the points are encoded here on the host (BackgroundScene encodes them inside
        swn_bg_sample_pe instead)."""
        xd = self.xyz_dim
        expected = xd if sigma_only else xd + 4
        if x.shape[1] != expected:
            raise Exception("Unexpected input shape: {} (expected: {}, xyz_dim: {})".format(x.shape, expected, xd))
        P = x.shape[0]
        xf = x.to(torch.float32)
        if sigma_only:
            xf = torch.cat([xf, torch.zeros(P, 4, device=self.dev)], 1)
            xf[:, xd + 2] = 1.0
        noise = None if sigma_noise is None else sigma_noise.reshape(-1).to(torch.float32).contiguous()
        img = xf[:, xd + 3].long().contiguous()
        if xd == 3:
            rays = torch.cat([xf[:, :6], torch.zeros(P, 2, device=self.dev)], 1).contiguous()
            c = self.forward_rays(rays, img, 1, P, 0.0, None, noise, training=self.training, composite=False)
        else:
            pts = xf[:, :xd]
            f = 2.0 ** torch.arange(self.cfg["pos_xyz_dim"], device=self.dev, dtype=torch.float32)
            ang = pts[:, None, :] * f[:, None]                                             # [P, L, xd]
            enc = torch.cat([pts, torch.stack([torch.sin(ang), torch.cos(ang)], 2).reshape(P, -1)], 1)   # nerf.py:21-26 order
            pe = torch.zeros(P, self.KP, dtype=self.dtype, device=self.dev)
            pe[:, : self.in_xyz] = enc.to(self.dtype)
            rays = torch.cat([torch.zeros(P, 3, device=self.dev), xf[:, xd:xd + 3], torch.zeros(P, 2, device=self.dev)], 1).contiguous()
            self._saving = bool(self.training)
            try:
                c = self._net_forward(pe, self._dir_pe(rays), img, P, 1, P, noise, None, False, "c")
            finally:
                self._saving = True
        return c["raw"][:, 3:4] if sigma_only else c["raw"]
