"""CPU oracle (TEST INFRASTRUCTURE, never on the product path) for the CUT trainer.

Functional PyTorch-CPU fp32 restatement of the reference's GAN_Variant1 hot path.
Parameters live in plain ``dict``s keyed by the reference's state_dict keys
(SURVEY.md §8b); every function cites the reference lines it follows
(paths relative to /root/reference).

Pinned by tests/golden/*.npz, generated from the imported reference by
oracle/make_golden.py.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]

IN_EPS = 1e-5  # nn.InstanceNorm2d default (generator_resnet_attn.py:56,114,126,150)


# ----------------------------------------------------------------------------------------------
# Parameter construction: consumes the global torch RNG in the same order as the reference's
# build_models (train_cutpp.py:88-124): nn.Conv2d / nn.ConvTranspose2d default init, layer by layer.
# ----------------------------------------------------------------------------------------------
def _conv_params(out: Params, key: str, cin: int, cout: int, k: int, transposed: bool = False, bias: bool = True):
    mod = (torch.nn.ConvTranspose2d if transposed else torch.nn.Conv2d)(cin, cout, k, bias=bias)
    out[key + ".weight"] = mod.weight.detach().clone()
    if bias:
        out[key + ".bias"] = mod.bias.detach().clone()


def init_generator(ngf: int = 64, n_blocks: int = 9, n_down: int = 2, in_c: int = 3, out_c: int = 3) -> Params:
    """Keys/layer order of ResNetGenerator.__init__ (generator_resnet_attn.py:104-163)."""
    p: Params = OrderedDict()
    _conv_params(p, "initial.1", in_c, ngf, 7)
    for i in range(n_down):
        _conv_params(p, f"downsample.{3 * i}", ngf * 2**i, ngf * 2 ** (i + 1), 3)
    c = ngf * 2**n_down
    for b in range(n_blocks):
        _conv_params(p, f"res_blocks.{b}.conv_block.1", c, c, 3)
        _conv_params(p, f"res_blocks.{b}.conv_block.5", c, c, 3)
    for i in range(n_down):
        m = 2 ** (n_down - i)
        _conv_params(p, f"upsample.{3 * i}", ngf * m, ngf * m // 2, 3, transposed=True)
    _conv_params(p, "output.1", ngf, out_c, 7)
    return p


def init_discriminator(ndf: int = 64, n_layers: int = 3, num_scales: int = 1, in_c: int = 3) -> Params:
    """Keys/layer order of MultiscaleDiscriminator / PatchGANDiscriminator (discriminator_patchgan.py:26-54,94-97)."""
    p: Params = OrderedDict()
    for s in range(num_scales):
        chans = [in_c, ndf] + [ndf * min(2**n, 8) for n in range(1, n_layers)] + [ndf * min(2**n_layers, 8), 1]
        for li in range(len(chans) - 1):
            _conv_params(p, f"discriminators.{s}.model.{2 * li}", chans[li], chans[li + 1], 4)
    return p


def set_seed(seed: int):
    """seed_dist.py:7-12."""
    import random

    import numpy as np

    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


# ----------------------------------------------------------------------------------------------
# Models
# ----------------------------------------------------------------------------------------------
def _inorm(x: Tensor) -> Tensor:
    return F.instance_norm(x, eps=IN_EPS)


def _rpad(x: Tensor, p: int) -> Tensor:
    return F.pad(x, (p, p, p, p), mode="reflect")


def generator_layers(p: Params, x: Tensor, n_blocks: int = 9, n_down: int = 2, upto_output: bool = True):
    """Walks the generator; returns (image_or_None, [activation after each numbered 'layer']).

    Layer numbering of get_feature_layers (generator_resnet_attn.py:204-233): 0 after ``initial``,
    1..n_down after each downsample ReLU, then one per residual block, then one per upsample ReLU.
    """
    acts: List[Tensor] = []
    h = F.relu(_inorm(F.conv2d(_rpad(x, 3), p["initial.1.weight"], p["initial.1.bias"])))  # :110-116
    acts.append(h)
    for i in range(n_down):  # :124-128
        h = F.relu(_inorm(F.conv2d(h, p[f"downsample.{3*i}.weight"], p[f"downsample.{3*i}.bias"], stride=2, padding=1)))
        acts.append(h)
    for b in range(n_blocks):  # :24-52, :70-71
        k = f"res_blocks.{b}.conv_block."
        t = F.relu(_inorm(F.conv2d(_rpad(h, 1), p[k + "1.weight"], p[k + "1.bias"])))
        t = _inorm(F.conv2d(_rpad(t, 1), p[k + "5.weight"], p[k + "5.bias"]))
        h = h + t
        acts.append(h)
    for i in range(n_down):  # :146-151
        h = F.relu(_inorm(F.conv_transpose2d(h, p[f"upsample.{3*i}.weight"], p[f"upsample.{3*i}.bias"],
                                             stride=2, padding=1, output_padding=1)))
        acts.append(h)
    img = None
    if upto_output:  # :157-162
        img = torch.tanh(F.conv2d(_rpad(h, 3), p["output.1.weight"], p["output.1.bias"]))
    return img, acts


def generator_forward(p: Params, x: Tensor, n_blocks: int = 9, n_down: int = 2) -> Tensor:
    """ResNetGenerator.forward (generator_resnet_attn.py:165-188)."""
    return generator_layers(p, x, n_blocks, n_down)[0]


def generator_features(p: Params, x: Tensor, layer_ids: Sequence[int], n_blocks: int = 9, n_down: int = 2) -> List[Tensor]:
    """ResNetGenerator.get_feature_layers (:190-235): ids outside 0..n_down*2+n_blocks-1 are silently dropped."""
    _, acts = generator_layers(p, x, n_blocks, n_down, upto_output=False)
    return [a for i, a in enumerate(acts) if i in layer_ids]


def discriminator_forward(p: Params, x: Tensor, n_layers: int = 3, num_scales: int = 1) -> List[Tensor]:
    """MultiscaleDiscriminator.forward (discriminator_patchgan.py:102-116) -> list of per-scale logits."""
    outs = []
    for s in range(num_scales):
        if s > 0:
            x = F.avg_pool2d(x, 3, 2, 1, count_include_pad=False)  # :100
        h = x
        nconv = n_layers + 2
        for li in range(nconv):
            k = f"discriminators.{s}.model.{2*li}"
            stride = 2 if li < n_layers else 1
            h = F.conv2d(h, p[k + ".weight"], p[k + ".bias"], stride=stride, padding=1)
            if li < nconv - 1:
                h = F.leaky_relu(h, 0.2)
        outs.append(h)
    return outs


# ----------------------------------------------------------------------------------------------
# Losses
# ----------------------------------------------------------------------------------------------
def d_hinge(real_preds: List[Tensor], fake_preds: List[Tensor]) -> Tensor:
    """discriminator_hinge_loss (adv_hinge.py:6-36)."""
    tot = 0.0
    for r, f in zip(real_preds, fake_preds):
        tot = tot + (F.relu(1.0 - r).mean() + F.relu(1.0 + f).mean()) * 0.5
    return tot / len(real_preds)


def g_hinge(fake_preds: List[Tensor]) -> Tensor:
    """generator_hinge_loss (adv_hinge.py:39-62)."""
    tot = 0.0
    for f in fake_preds:
        tot = tot - f.mean()
    return tot / len(fake_preds)


def patchnce_layer(src: Tensor, tgt: Tensor, ids: Tensor, temperature: float = 0.07) -> Tensor:
    """PatchNCELoss._compute_nce_loss (patchnce_cut.py:42-110) with the patch ids injected.

    ids: (P,) int64 positions in [0, H*W), shared by the whole batch and by src/tgt (:63).
    """
    B, C = src.shape[:2]
    s = src.reshape(B, C, -1).permute(0, 2, 1)[:, ids, :]  # (B,P,C) :56-71
    t = tgt.reshape(B, C, -1).permute(0, 2, 1)[:, ids, :]
    s = F.normalize(s, dim=2, eps=1e-6)  # :78-79
    t = F.normalize(t, dim=2, eps=1e-6)
    logits = torch.bmm(t, s.transpose(1, 2)) / temperature  # :85
    logits = logits.clamp(-50.0, 50.0)  # :88
    P = ids.numel()
    labels = torch.arange(P).repeat(B)
    per_img = F.cross_entropy(logits.reshape(B * P, P), labels, reduction="none").reshape(B, P).mean(1)  # :94
    per_img = torch.where(torch.isfinite(per_img), per_img, torch.zeros_like(per_img))  # :97-99
    return per_img.sum() / B  # :103


def patchnce(src_feats: List[Tensor], tgt_feats: List[Tensor], ids_list: List[Tensor], temperature: float = 0.07) -> Tensor:
    """PatchNCELoss.forward (patchnce_cut.py:25-40): mean over the feature maps actually returned."""
    tot = 0.0
    for s, t, ids in zip(src_feats, tgt_feats, ids_list):
        tot = tot + patchnce_layer(s.detach(), t, ids, temperature)
    return tot / len(src_feats)


# ----------------------------------------------------------------------------------------------
# DiffAugment with injected draws (diffaugment.py:6-60, policy order color -> translation -> cutout)
# ----------------------------------------------------------------------------------------------
def sample_diffaugment(B: int, H: int, W: int, policy: Sequence[str] = ("color", "translation", "cutout"),
                       generator: Optional[torch.Generator] = None) -> Dict[str, Tensor]:
    """Draws in the reference's consumption order and shapes (diffaugment.py:8,15,22,29-30,47-48)."""
    d: Dict[str, Tensor] = {}
    for pol in policy:
        if pol == "color":
            d["brightness"] = torch.rand(B, 1, 1, 1, generator=generator)
            d["saturation"] = torch.rand(B, 1, 1, 1, generator=generator)
            d["contrast"] = torch.rand(B, 1, 1, 1, generator=generator)
        elif pol == "translation":
            sx, sy = int(H * 0.125 + 0.5), int(W * 0.125 + 0.5)
            d["tx"] = torch.randint(-sx, sx + 1, size=[B, 1, 1], generator=generator)
            d["ty"] = torch.randint(-sy, sy + 1, size=[B, 1, 1], generator=generator)
        elif pol in ("cutout", "cutout_light"):
            ratio = 0.5 if pol == "cutout" else 0.2
            ch, cw = int(H * ratio + 0.5), int(W * ratio + 0.5)
            d["cut_h"], d["cut_w"] = torch.tensor(ch), torch.tensor(cw)
            d["cx"] = torch.randint(0, H + (1 - ch % 2), size=[B, 1, 1], generator=generator)
            d["cy"] = torch.randint(0, W + (1 - cw % 2), size=[B, 1, 1], generator=generator)
    return d


def diffaugment(x: Tensor, d: Dict[str, Tensor]) -> Tensor:
    B, C, H, W = x.shape
    if "brightness" in d:
        x = x + (d["brightness"].to(x.dtype) - 0.5)  # :8
        m = x.mean(dim=1, keepdim=True)
        x = (x - m) * (d["saturation"].to(x.dtype) * 2) + m  # :14-15
        m = x.mean(dim=[1, 2, 3], keepdim=True)
        x = (x - m) * (d["contrast"].to(x.dtype) + 0.5) + m  # :21-22
    if "tx" in d:  # :26-41 : out[h,w] = x[h+tx, w+ty] or 0 outside
        hh = torch.arange(H).view(1, H, 1) + d["tx"]
        ww = torch.arange(W).view(1, 1, W) + d["ty"]
        ok = ((hh >= 0) & (hh < H) & (ww >= 0) & (ww < W)).unsqueeze(1).to(x.dtype)
        hh = hh.clamp(0, H - 1).expand(B, H, W)
        ww = ww.clamp(0, W - 1).expand(B, H, W)
        bb = torch.arange(B).view(B, 1, 1).expand(B, H, W)
        x = x.permute(0, 2, 3, 1)[bb, hh, ww].permute(0, 3, 1, 2) * ok
    if "cx" in d:  # :44-60 : zero a (clamped) cut_h x cut_w window centred on (cx, cy)
        ch, cw = int(d["cut_h"]), int(d["cut_w"])
        lo_h = (d["cx"] - ch // 2).clamp(0, H - 1)
        hi_h = (d["cx"] + ch - 1 - ch // 2).clamp(0, H - 1)
        lo_w = (d["cy"] - cw // 2).clamp(0, W - 1)
        hi_w = (d["cy"] + cw - 1 - cw // 2).clamp(0, W - 1)
        hh = torch.arange(H).view(1, H, 1)
        ww = torch.arange(W).view(1, 1, W)
        inside = (hh >= lo_h) & (hh <= hi_h) & (ww >= lo_w) & (ww <= hi_w)
        x = x * (~inside).unsqueeze(1).to(x.dtype)
    return x


# ----------------------------------------------------------------------------------------------
# R1 (train_cutpp.py:165-203), optimiser (Adam + clip_grad_norm_), EMA (io_ckpt.py:23-29)
# ----------------------------------------------------------------------------------------------
def r1_penalty(dp: Params, real: Tensor, n_layers: int = 3, num_scales: int = 1) -> Tensor:
    real = real.detach().requires_grad_(True)
    s = sum(o.sum() for o in discriminator_forward(dp, real, n_layers, num_scales))
    (g,) = torch.autograd.grad(s, real, create_graph=True)
    return g.pow(2).reshape(g.size(0), -1).sum(1).mean()


class AdamState:
    """torch.optim.Adam single-tensor maths (torch/optim/adam.py:528-546), per-parameter step counts.

    Parameters whose grad is None are skipped entirely (no moment / step update), as torch does.
    """

    def __init__(self, params: Params, lr: float = 2e-4, betas=(0.5, 0.999), eps: float = 1e-8):
        self.lr, self.b1, self.b2, self.eps = lr, betas[0], betas[1], eps
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.t = {k: 0 for k in params}

    @torch.no_grad()
    def step(self, params: Params, grads: Dict[str, Optional[Tensor]], max_norm: Optional[float] = None) -> float:
        live = [k for k in params if grads.get(k) is not None]
        total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(grads[k]) for k in live])).item() if live else 0.0
        coef = 1.0
        if max_norm is not None:  # clip_grad_norm_ (torch/nn/utils/clip_grad.py:165-169), amp_utils.py:33-38
            coef = min(1.0, max_norm / (total + 1e-6))
        for k in live:
            g = grads[k] * coef
            self.t[k] += 1
            t = self.t[k]
            self.m[k].lerp_(g, 1 - self.b1)
            self.v[k].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            bc1, bc2 = 1 - self.b1**t, 1 - self.b2**t
            denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
            params[k].addcdiv_(self.m[k], denom, value=-(self.lr / bc1))
        return total


def ema_update(shadow: Params, params: Params, decay: float):
    for k in params:
        shadow[k] = (1.0 - decay) * params[k].detach() + decay * shadow[k]


# ----------------------------------------------------------------------------------------------
# Step randomness and the step itself (train_cutpp.py:206-331)
# ----------------------------------------------------------------------------------------------
NCE_LAYERS = (0, 4, 8, 12, 16)


def feature_hw(H: int, W: int, layer_ids=NCE_LAYERS, n_blocks: int = 9, n_down: int = 2) -> List[int]:
    """H*W of each feature map get_feature_layers actually returns for a HxW input."""
    hw = [H * W] + [(H >> (i + 1)) * (W >> (i + 1)) for i in range(n_down)]
    hw += [hw[-1]] * n_blocks
    hw += [(H >> (n_down - 1 - i)) * (W >> (n_down - 1 - i)) for i in range(n_down)]
    return [v for i, v in enumerate(hw) if i in layer_ids]


def sample_step_randomness(B: int, H: int, W: int, policy=("color", "translation", "cutout"), num_patches: int = 256,
                           layer_ids=NCE_LAYERS, use_aug: bool = True, generator: Optional[torch.Generator] = None):
    """All device-RNG draws of one train_step in the reference's consumption order:
    D-step aug(photos), aug(fake); G-step aug(fake); then one randint per returned NCE feature map."""
    r = {}
    if use_aug:
        r["aug_real"] = sample_diffaugment(B, H, W, policy, generator)
        r["aug_fake_d"] = sample_diffaugment(B, H, W, policy, generator)
        r["aug_fake_g"] = sample_diffaugment(B, H, W, policy, generator)
    r["nce_ids"] = [torch.randint(0, hw, (min(num_patches, hw),), generator=generator) for hw in feature_hw(H, W, layer_ids)]
    return r


def identity_weight_at(step: int, cfg: dict) -> float:
    lw, ws = cfg["loss_weights"], cfg.get("warmup_steps", 20000)  # :224-228
    if step < ws:
        return lw["identity_warm"] + (lw["identity_final"] - lw["identity_warm"]) * (step / ws)
    return lw["identity_final"]


def _grads(loss: Tensor, params: Params) -> Dict[str, Optional[Tensor]]:
    keys = list(params)
    gs = torch.autograd.grad(loss, [params[k] for k in keys], allow_unused=True)
    return dict(zip(keys, gs))


def train_step(step: int, photos: Tensor, monets: Tensor, gp: Params, dp: Params, opt_g: AdamState, opt_d: AdamState,
               ema: Optional[Params], cfg: dict, rnd: dict, ema_decay: float = 0.999) -> Dict[str, float]:
    """One CUT step, amp disabled (pure fp32), same op order as train_cutpp.py:206-331."""
    lw = cfg["loss_weights"]
    idw = identity_weight_at(step, cfg)
    for prm in (gp, dp):
        for v in prm.values():
            v.requires_grad_(True)
    # ---- D step (:231-254)
    with torch.no_grad():
        fake = generator_forward(gp, photos)
    real_in = diffaugment(photos, rnd["aug_real"]) if "aug_real" in rnd else photos
    fake_in = diffaugment(fake, rnd["aug_fake_d"]) if "aug_fake_d" in rnd else fake
    d_loss = d_hinge(discriminator_forward(dp, real_in), discriminator_forward(dp, fake_in))
    opt_d.step(dp, _grads(d_loss, dp), cfg.get("grad_clip_d", 10.0))
    # ---- lazy R1 (:257-263)
    r1 = torch.tensor(0.0)
    if cfg["r1"]["gamma"] > 0 and step % cfg["r1"]["every"] == 0:
        r1 = r1_penalty(dp, photos)
        opt_d.step(dp, _grads(r1 * cfg["r1"]["gamma"] * cfg["r1"]["every"], dp), cfg.get("grad_clip_d", 10.0))
    # ---- G step (:266-308)
    fake = generator_forward(gp, photos)
    fake_in = diffaugment(fake, rnd["aug_fake_g"]) if "aug_fake_g" in rnd else fake
    g_adv = g_hinge(discriminator_forward(dp, fake_in))
    nce = torch.tensor(0.0)
    if lw["patchnce"] > 0:
        layers = cfg["patchnce"]["nce_layers"]
        with torch.no_grad():
            src = generator_features(gp, photos, layers)
        tgt = generator_features(gp, fake, layers)
        nce = patchnce(src, tgt, rnd["nce_ids"], cfg["patchnce"]["temperature"])
    idt = torch.tensor(0.0)
    if idw > 0:
        idt = (generator_forward(gp, monets) - monets).abs().mean()  # identity_l1.py:18-20
    g_loss = lw["adv"] * g_adv + lw["patchnce"] * nce + idw * idt
    opt_g.step(gp, _grads(g_loss, gp), cfg.get("grad_clip_g", 10.0))
    if ema is not None:
        ema_update(ema, gp, ema_decay)  # :311-312
    return {"d_loss": float(d_loss.detach()), "g_loss": float(g_loss.detach()), "g_adv": float(g_adv.detach()), "nce": float(nce.detach()),
            "identity": float(idt.detach()), "r1": float(r1.detach()), "identity_weight": idw}


def default_config() -> dict:
    """The keys train_step reads, with the values of GAN_Variant1/configs/train_gan_cutpp.yaml."""
    return {
        "loss_weights": {"adv": 1.0, "patchnce": 1.0, "identity_warm": 0.1, "identity_final": 0.0},
        "warmup_steps": 20000, "grad_clip_g": 10.0, "grad_clip_d": 10.0,
        "patchnce": {"nce_layers": [0, 4, 8, 12, 16], "temperature": 0.07, "num_patches": 256},
        "r1": {"gamma": 10.0, "every": 16}, "ema": {"decay": 0.999},
        "diffaugment": {"enable": True, "policy": ["color", "translation", "cutout"]},
    }
