"""CPU oracle (TEST INFRASTRUCTURE, never on the product path) for the Basic_GAN CycleGAN inner loop.

Functional PyTorch-CPU fp32 restatement of Basic_GAN/src/{models,losses,train}.py (paths relative to
/root/reference).  Pinned by tests/golden/basic_*.npz (oracle/make_golden.py).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

from .cut_ref import AdamState, Params, _conv_params, _grads, _inorm, _rpad

Tensor = torch.Tensor


def init_generator(ngf: int = 64, n_blocks: int = 9, in_c: int = 3, out_c: int = 3) -> Params:
    """Keys/order of ResnetGenerator.__init__ (Basic_GAN/src/models.py:24-62): all convs bias-free but the last."""
    p: Params = OrderedDict()
    _conv_params(p, "net.1", in_c, ngf, 7, bias=False)
    idx, mult = 4, 1
    for _ in range(2):
        _conv_params(p, f"net.{idx}", ngf * mult, ngf * mult * 2, 3, bias=False)
        idx += 3
        mult *= 2
    for _ in range(n_blocks):
        _conv_params(p, f"net.{idx}.block.1", ngf * mult, ngf * mult, 3, bias=False)
        _conv_params(p, f"net.{idx}.block.5", ngf * mult, ngf * mult, 3, bias=False)
        idx += 1
    for _ in range(2):
        _conv_params(p, f"net.{idx}", ngf * mult, ngf * mult // 2, 3, transposed=True, bias=False)
        idx += 3
        mult //= 2
    _conv_params(p, f"net.{idx + 1}", ngf, out_c, 7, bias=True)
    return p


def init_discriminator(ndf: int = 64, n_layers: int = 3, in_c: int = 3) -> Params:
    """Keys/order of NLayerDiscriminator.__init__ (Basic_GAN/src/models.py:75-105)."""
    p: Params = OrderedDict()
    _conv_params(p, "net.0", in_c, ndf, 4, bias=True)
    idx, mult = 2, 1
    for n in range(1, n_layers + 1):
        prev, mult = mult, min(2**n, 8)
        _conv_params(p, f"net.{idx}", ndf * prev, ndf * mult, 4, bias=False)
        idx += 3
    _conv_params(p, f"net.{idx}", ndf * mult, 1, 4, bias=True)
    return p


def generator_forward(p: Params, x: Tensor, n_blocks: int = 9) -> Tensor:
    """ResnetGenerator.forward (models.py:23-65)."""
    h = F.relu(_inorm(F.conv2d(_rpad(x, 3), p["net.1.weight"])))
    idx = 4
    for _ in range(2):
        h = F.relu(_inorm(F.conv2d(h, p[f"net.{idx}.weight"], stride=2, padding=1)))
        idx += 3
    for _ in range(n_blocks):  # ResnetBlock :7-21
        t = F.relu(_inorm(F.conv2d(_rpad(h, 1), p[f"net.{idx}.block.1.weight"])))
        t = _inorm(F.conv2d(_rpad(t, 1), p[f"net.{idx}.block.5.weight"]))
        h = h + t
        idx += 1
    for _ in range(2):
        h = F.relu(_inorm(F.conv_transpose2d(h, p[f"net.{idx}.weight"], stride=2, padding=1, output_padding=1)))
        idx += 3
    return torch.tanh(F.conv2d(_rpad(h, 3), p[f"net.{idx + 1}.weight"], p[f"net.{idx + 1}.bias"]))


def discriminator_forward(p: Params, x: Tensor, n_layers: int = 3) -> Tensor:
    """NLayerDiscriminator.forward (models.py:71-107): conv-lrelu, 3x(conv-IN-lrelu), conv."""
    h = F.leaky_relu(F.conv2d(x, p["net.0.weight"], p["net.0.bias"], stride=2, padding=1), 0.2)
    idx = 2
    for n in range(1, n_layers + 1):
        stride = 2 if n < n_layers else 1
        h = F.leaky_relu(_inorm(F.conv2d(h, p[f"net.{idx}.weight"], stride=stride, padding=1)), 0.2)
        idx += 3
    return F.conv2d(h, p[f"net.{idx}.weight"], p[f"net.{idx}.bias"], stride=1, padding=1)


def gan_loss(pred: Tensor, is_real: bool, mode: str = "lsgan") -> Tensor:
    """GANLoss (Basic_GAN/src/losses.py:5-22)."""
    tgt = torch.ones_like(pred) if is_real else torch.zeros_like(pred)
    return F.mse_loss(pred, tgt) if mode == "lsgan" else F.binary_cross_entropy_with_logits(pred, tgt)


def train_iteration(real_a: Tensor, real_b: Tensor, g_ab: Params, g_ba: Params, d_a: Params, d_b: Params,
                    opt_g: AdamState, opt_da: AdamState, opt_db: AdamState, lam_cyc: float = 10.0, lam_id: float = 0.5,
                    mode: str = "lsgan") -> Dict[str, float]:
    """Inner loop body of train() (Basic_GAN/src/train.py:66-122), amp disabled.

    opt_g owns the parameters of both generators under the keys 'ab.<k>' / 'ba.<k>' (train.py:45-48).
    """
    for prm in (g_ab, g_ba, d_a, d_b):
        for v in prm.values():
            v.requires_grad_(True)
    fake_b = generator_forward(g_ab, real_a)
    rec_a = generator_forward(g_ba, fake_b)
    fake_a = generator_forward(g_ba, real_b)
    rec_b = generator_forward(g_ab, fake_a)
    idt_b = generator_forward(g_ab, real_b)
    idt_a = generator_forward(g_ba, real_a)
    loss_g = (gan_loss(discriminator_forward(d_b, fake_b), True, mode) + gan_loss(discriminator_forward(d_a, fake_a), True, mode)
              + lam_cyc * (rec_a - real_a).abs().mean() + lam_cyc * (rec_b - real_b).abs().mean()
              + lam_id * (idt_a - real_a).abs().mean() + lam_id * (idt_b - real_b).abs().mean())
    both = {**{"ab." + k: v for k, v in g_ab.items()}, **{"ba." + k: v for k, v in g_ba.items()}}
    opt_g.step(both, _grads(loss_g, both))
    loss_da = 0.5 * (gan_loss(discriminator_forward(d_a, real_a), True, mode)
                     + gan_loss(discriminator_forward(d_a, fake_a.detach()), False, mode))
    opt_da.step(d_a, _grads(loss_da, d_a))
    loss_db = 0.5 * (gan_loss(discriminator_forward(d_b, real_b), True, mode)
                     + gan_loss(discriminator_forward(d_b, fake_b.detach()), False, mode))
    opt_db.step(d_b, _grads(loss_db, d_b))
    return {"loss_G": float(loss_g), "loss_D_A": float(loss_da), "loss_D_B": float(loss_db)}
