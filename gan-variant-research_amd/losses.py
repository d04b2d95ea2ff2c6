"""The reference's loss / augmentation callables on the HIP kernels, as ordinary differentiable PyTorch functions.

Same names, signatures and RNG consumption as GAN_Variant1/losses/{adv_hinge,patchnce_cut,identity_l1}.py,
GAN_Variant1/training/diffaugment.py and Basic_GAN/src/losses.py, so a script written against the reference imports them from
here unchanged.  Each call is one autograd node whose forward launches the fused HIP kernel (value and input gradient come out
of the same pass) on NCHW fp32 tensors; the fused trainers (cut.py / basic.py) use the same kernels without the layout round
trip.  There is no CPU fallback: tensors must live on the GPU.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from . import autograd as AG
from ._lib import F32, HALO_NONE, HALO_ZERO
from .cut import DiffAugment as _DiffAugmentSampler
from .runtime import Program, cpad

_PLANS: Dict[tuple, object] = {}


def _plan(key, make):
    key = key + (torch.cuda.current_stream().cuda_stream if torch.cuda.is_available() else 0,)      # see ops_library._stream_key
    p = _PLANS.get(key)
    if p is None:
        p = _PLANS[key] = make()
    return p


class _Plan:
    """Persistent staging tensors + prebuilt launches for one (op, shape, device): ops hold raw pointers, so nothing per call."""

    def __init__(self, device):
        self.ctx = AG._new_ctx(device, F32)
        self.ops = self.ctx.ops
        self.device = device


# ------------------------------------------------------------------------------------------------ patch losses (a9)
class _PatchLossPlan(_Plan):
    def __init__(self, shape, device, mode, target):
        super().__init__(device)
        B, C, H, W = shape
        assert C == 1, "PatchGAN logits are (B,1,h,w)"
        self.x = torch.zeros(shape, dtype=torch.float32, device=device)
        self.g = torch.zeros(shape, dtype=torch.float32, device=device)
        self.loss = self.ctx.f32(1)
        v, gv = self.ctx.view(B, H, W, cpad(1), 0), self.ctx.view(B, H, W, cpad(1), 0)
        self.prog = Program("patch_loss")
        self.prog.add(self.ops.nchw_to_view(self.x, 1, v, HALO_NONE))
        self.prog.add(self.ops.patch_loss(v, mode, float(target), 1.0, self.loss, gv))
        self.prog.add(self.ops.view_to_nchw(gv, 1, self.g))


class _PatchLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mode, target):
        p = _plan(("patch_loss", tuple(x.shape), x.device, mode, float(target)), lambda: _PatchLossPlan(tuple(x.shape), x.device, mode, target))
        p.x.copy_(x)
        p.prog.run()
        ctx.save_for_backward(p.g.clone())
        return p.loss.clone().reshape(())

    @staticmethod
    def backward(ctx, g):
        return ctx.saved_tensors[0] * g, None, None


def _patch_loss(x: torch.Tensor, mode: int, target: float = 0.0) -> torch.Tensor:
    return _PatchLossFn.apply(x.float().contiguous(), mode, target)


def _as_list(p):
    return list(p) if isinstance(p, (list, tuple)) else [p]


def discriminator_hinge_loss(real_preds, fake_preds):
    """adv_hinge.py:6-36: mean over scales of 0.5 * (mean(relu(1 - real)) + mean(relu(1 + fake)))."""
    real_preds, fake_preds = _as_list(real_preds), _as_list(fake_preds)
    loss = 0.0
    for r, f in zip(real_preds, fake_preds):
        loss = loss + (_patch_loss(r, 0) + _patch_loss(f, 1)) * 0.5
    return loss / len(real_preds)


def generator_hinge_loss(fake_preds):
    """adv_hinge.py:39-62: mean over scales of -mean(fake)."""
    fake_preds = _as_list(fake_preds)
    loss = 0.0
    for f in fake_preds:
        loss = loss + _patch_loss(f, 2)
    return loss / len(fake_preds)


class GANLoss:
    """Basic_GAN/src/losses.py:5-22: 'lsgan' -> MSE against {1,0}; 'bce' -> BCE-with-logits."""

    def __init__(self, mode: str = "lsgan"):
        assert mode in ("lsgan", "bce")
        self.mode = mode

    def __call__(self, pred: torch.Tensor, is_real: bool) -> torch.Tensor:
        return _patch_loss(pred, 3 if self.mode == "lsgan" else 4, 1.0 if is_real else 0.0)


# ------------------------------------------------------------------------------------------------ L1 (a9)
class _L1Plan(_Plan):
    def __init__(self, shape, device):
        super().__init__(device)
        B, C, H, W = shape
        self.x = torch.zeros(shape, dtype=torch.float32, device=device)
        self.t = torch.zeros(shape, dtype=torch.float32, device=device)
        self.g = torch.zeros(shape, dtype=torch.float32, device=device)
        self.loss = self.ctx.f32(1)
        v, gv = self.ctx.view(B, H, W, cpad(C), 0), self.ctx.view(B, H, W, cpad(C), 0)
        self.prog = Program("l1")
        self.prog.add(self.ops.nchw_to_view(self.x, C, v, HALO_NONE))
        self.prog.add(self.ops.l1_loss(v, C, self.t, 1.0, None, self.loss, gv, self.ctx.scratch("l1_ws", 1024)))
        self.prog.add(self.ops.view_to_nchw(gv, C, self.g))


class _L1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, target):
        p = _plan(("l1", tuple(x.shape), x.device), lambda: _L1Plan(tuple(x.shape), x.device))
        p.x.copy_(x); p.t.copy_(target)
        p.prog.run()
        ctx.save_for_backward(p.g.clone())
        return p.loss.clone().reshape(())

    @staticmethod
    def backward(ctx, g):
        d = ctx.saved_tensors[0] * g
        return d, (-d if ctx.needs_input_grad[1] else None)


def l1_loss(x: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """mean |x - target| (Basic_GAN/src/losses.py:24-30 cycle_loss / identity_loss bodies)."""
    return _L1Fn.apply(x.float().contiguous(), target.float().contiguous())


def identity_loss(generator, monet_images):
    """identity_l1.py:6-22: mean |G(monet) - monet| in fp32."""
    return l1_loss(generator(monet_images.float()), monet_images)


def cycle_loss(rec, real, lambda_cycle: float = 10.0):
    """Basic_GAN/src/losses.py:24-26."""
    return lambda_cycle * l1_loss(rec, real)


# ------------------------------------------------------------------------------------------------ PatchNCE (a8)
class _NcePlan(_Plan):
    def __init__(self, shape, device, P, temperature):
        super().__init__(device)
        B, C, H, W = shape
        self.src = torch.zeros(shape, dtype=torch.float32, device=device)
        self.tgt = torch.zeros(shape, dtype=torch.float32, device=device)
        self.g = torch.zeros(shape, dtype=torch.float32, device=device)
        self.ids = torch.zeros(P, dtype=torch.int32, device=device)
        self.loss = self.ctx.f32(1)
        ops, ctx = self.ops, self.ctx
        vs, vt, gv = ctx.view(B, H, W, cpad(C), 0), ctx.view(B, H, W, cpad(C), 0), ctx.view(B, H, W, cpad(C), 0)
        ws = ctx.f32(ops.patchnce_ws_floats(B, P, C))
        self.fwd = Program("patchnce.fwd")
        self.fwd.add(ops.nchw_to_view(self.src, C, vs, HALO_NONE))
        self.fwd.add(ops.nchw_to_view(self.tgt, C, vt, HALO_NONE))
        self.fwd.add(ops.fill(self.loss, 0.0))
        self.fwd.add(ops.patchnce_fwd(vs, vt, self.ids, P, C, temperature, 1.0, self.loss, ws))
        self.fwd.add(ops.zero_(gv.t))                       # the backward kernel ADDS into the feature gradient
        self.fwd.add(ops.patchnce_bwd(vt, self.ids, P, C, temperature, 1.0, gv, ws))
        self.fwd.add(ops.view_to_nchw(gv, C, self.g))


class PatchNCELoss(nn.Module):
    """patchnce_cut.py:7-110.  One patch-id draw per layer (`torch.randint(0, H*W, (num_patches,), device=...)`, :63) shared by
    the batch and by source / target; the source features carry no gradient (compute_patchnce_loss detaches them)."""

    def __init__(self, temperature: float = 0.07, num_patches: int = 256, nce_layers: Sequence[int] = (0, 4, 8, 12, 16)):
        super().__init__()
        self.temperature, self.num_patches, self.nce_layers = temperature, num_patches, list(nce_layers)

    def _compute_nce_loss(self, src_feat, tgt_feat, patch_ids: Optional[torch.Tensor] = None):
        B, C, H, W = tgt_feat.shape
        if patch_ids is None:
            patch_ids = torch.randint(0, H * W, (self.num_patches,), device=tgt_feat.device)
        from . import ops_library  # noqa: F401  (registers torch.ops.mi355x_gan.*)
        loss, _ = torch.ops.mi355x_gan.patchnce_fwd(src_feat.detach(), tgt_feat, patch_ids.to(torch.int32), self.temperature)
        return loss

    def forward(self, src_feats, tgt_feats, patch_ids: Optional[List[torch.Tensor]] = None):
        total = 0.0
        for i, (s, t) in enumerate(zip(src_feats, tgt_feats)):
            total = total + self._compute_nce_loss(s, t, None if patch_ids is None else patch_ids[i])
        return total / len(src_feats)


def compute_patchnce_loss(generator, src_images, tgt_images, nce_layers, temperature=0.07, num_patches=256):
    """patchnce_cut.py:113-149."""
    fn = PatchNCELoss(temperature, num_patches, nce_layers)
    with torch.no_grad():
        src_feats = generator.get_feature_layers(src_images, nce_layers)
    src_feats = [f.detach() for f in src_feats]
    tgt_feats = generator.get_feature_layers(tgt_images, nce_layers)
    return fn(src_feats, tgt_feats)


# ------------------------------------------------------------------------------------------------ DiffAugment (a11)
class _AugPlan(_Plan):
    def __init__(self, shape, device):
        super().__init__(device)
        B, C, H, W = shape
        self.x = torch.zeros(shape, dtype=torch.float32, device=device)
        self.y = torch.zeros(shape, dtype=torch.float32, device=device)
        self.gy = torch.zeros(shape, dtype=torch.float32, device=device)
        self.gx = torch.zeros(shape, dtype=torch.float32, device=device)
        self.prm = torch.zeros(B, 12, dtype=torch.float32, device=device)
        ops, ctx = self.ops, self.ctx
        vx, vy, vgy, vgx = (ctx.view(B, H, W, cpad(C), 0) for _ in range(4))
        ws = ctx.scratch("aug_ws", B + 16)
        self.fwd = Program("diffaug.fwd")
        self.fwd.add(ops.nchw_to_view(self.x, C, vx, HALO_NONE))
        self.fwd.add(ops.diffaug_fwd(vx, C, self.prm, vy, ws))
        self.fwd.add(ops.view_to_nchw(vy, C, self.y))
        self.bwd = Program("diffaug.bwd")
        self.bwd.add(ops.nchw_to_view(self.gy, C, vgy, HALO_NONE))
        self.bwd.add(ops.diffaug_bwd(vgy, C, self.prm, vgx, ws))
        self.bwd.add(ops.view_to_nchw(vgx, C, self.gx))


class DiffAugment(_DiffAugmentSampler):
    """training/diffaugment.py:76-106 as a differentiable callable: brightness, saturation, contrast, translation and cutout in
    one gather pass each way.  `generator` (optional) makes the per-sample draws reproducible; `last_draws` keeps them."""

    def __call__(self, x: torch.Tensor, generator: Optional[torch.Generator] = None, draws: Optional[dict] = None) -> torch.Tensor:
        B, C, H, W = x.shape
        self.last_draws = draws if draws is not None else self.sample(B, H, W, generator)
        prm = self.to_params(self.last_draws, B, H, W).to(x.device)
        from . import ops_library  # noqa: F401
        return torch.ops.mi355x_gan.diffaugment_fwd(x, prm)
