"""The reference's CUT `train_step` (GAN_Variant1/training/train_cutpp.py:206-331) written against the drop-in module API of this
package: differentiable HIP `generator` / `discriminator` modules (autograd.py), the reference-named losses (losses.py) and
training utilities (training.py).  Same signature, same order of operations, same returned keys, same NaN behaviour.

This is the compatibility path -- every forward the reference runs is run (five generator passes per step); the fused
`cut.CutTrainer` computes the same step with one shared `G(photos)` forward and is what `bench.py` measures.
`rnd` (optional, not in the reference's signature) injects the device-RNG draws of the step for parity tests:
{"aug_real", "aug_fake_d", "aug_fake_g": DiffAugment draws, "nce_ids": one id tensor per PatchNCE layer}.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from .cut import identity_weight_at
from .autograd import r1_regularization
from .losses import PatchNCELoss, discriminator_hinge_loss, generator_hinge_loss, identity_loss


def train_step(step, photos, monets, generator, discriminator, opt_G, opt_D, ema_G, amp_ctx, diffaugment, config, device, rnd: Optional[dict] = None):
    lw = config["loss_weights"]
    identity_weight = identity_weight_at(step, config)                                    # :224-228
    rnd = rnd or {}

    def aug(x, key):
        return x if diffaugment is None else diffaugment(x, draws=rnd.get(key))

    # ---- discriminator step (:231-254)
    opt_D.zero_grad()
    with amp_ctx.autocast():
        fake = generator(photos)
        real_pred = discriminator(aug(photos, "aug_real"))
        fake_pred = discriminator(aug(fake.detach(), "aug_fake_d"))
        d_loss = discriminator_hinge_loss(real_pred, fake_pred)
    amp_ctx.scale_backward(d_loss)
    amp_ctx.step_optimizer(opt_D, max_grad_norm=config.get("grad_clip_d", 10.0))
    # ---- lazy R1 (:257-263)
    r1_loss = torch.tensor(0.0, device=device)
    if config["r1"]["gamma"] > 0 and step % config["r1"]["every"] == 0:
        opt_D.zero_grad()
        r1_loss = r1_regularization(discriminator, photos, amp_ctx)
        amp_ctx.scale_backward(r1_loss * config["r1"]["gamma"] * config["r1"]["every"])
        amp_ctx.step_optimizer(opt_D, max_grad_norm=config.get("grad_clip_d", 10.0))
    # ---- generator step (:266-308)
    opt_G.zero_grad()
    with amp_ctx.autocast():
        fake = generator(photos)
        g_adv_loss = generator_hinge_loss(discriminator(aug(fake, "aug_fake_g")))
        nce_loss = torch.tensor(0.0, device=device)
        if lw["patchnce"] > 0:
            pn = config["patchnce"]
            fn = PatchNCELoss(pn["temperature"], pn["num_patches"], pn["nce_layers"])
            with torch.no_grad():
                src_feats = generator.get_feature_layers(photos, pn["nce_layers"])
            tgt_feats = generator.get_feature_layers(fake, pn["nce_layers"])
            nce_loss = fn(src_feats, tgt_feats, rnd.get("nce_ids"))                           # compute_patchnce_loss, :113-149
        idt_loss = torch.tensor(0.0, device=device)
        if identity_weight > 0:
            idt_loss = identity_loss(generator, monets)
        g_loss = lw["adv"] * g_adv_loss + lw["patchnce"] * nce_loss + identity_weight * idt_loss
    amp_ctx.scale_backward(g_loss)
    amp_ctx.step_optimizer(opt_G, max_grad_norm=config.get("grad_clip_g", 10.0))
    if ema_G is not None:
        ema_G.update()
    losses = {"d_loss": d_loss.item(), "g_loss": g_loss.item(), "g_adv": g_adv_loss.item(), "nce": nce_loss.item(), "identity": idt_loss.item(),
              "r1": r1_loss.item(), "identity_weight": identity_weight}
    if any(not math.isfinite(v) for k, v in losses.items() if k != "identity_weight"):
        raise ValueError(f"NaN loss detected at step {step}. Training stopped to prevent corruption.")   # :325-329
    return losses
