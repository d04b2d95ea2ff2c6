"""torch.ops.mi355x_gan.* (gan_variant_research_amd/ops_library.py): the op-level custom ops with autograd formulas and the drop-in
Conv2d / ConvTranspose2d / InstanceNorm2d / ReflectionPad2d modules.  A generator and a discriminator assembled LAYER BY LAYER the
way the reference's model files do (generator_resnet_attn.py:104-163, discriminator_patchgan.py:26-54) must reproduce the outputs
the reference itself produced (tests/golden/cut_models.npz) and the oracle's gradients.  Here on CPU through the emulator (host
logic: geometry, halos, phases, autograd wiring); tests/test_gpu_parity.py::test_ops_library_hip runs the same body on the kernels."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from gan_variant_research_amd import autograd as AG
from gan_variant_research_amd import cut as C
from oracle import cut_ref
from tests import cases
from tests.emulator import EmuOps

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cut_models.npz")


def build_layerwise_generator(L, ngf=64, n_blocks=9):
    """ResNetGenerator as generator_resnet_attn.py:104-163 builds it (same Sequential indices -> same state_dict keys), from the
    op-level modules of ops_library."""
    class ResidualBlock(nn.Module):            # generator_resnet_attn.py:7-71 (reflect padding, instance norm, ReLU)
        def __init__(self, dim):
            super().__init__()
            self.conv_block = nn.Sequential(L.ReflectionPad2d(1), L.Conv2d(dim, dim, 3, padding=0), L.InstanceNorm2d(dim), nn.ReLU(True),
                                            L.ReflectionPad2d(1), L.Conv2d(dim, dim, 3, padding=0), L.InstanceNorm2d(dim))

        def forward(self, x):
            return x + self.conv_block(x)

    class Gen(nn.Module):
        def __init__(self):
            super().__init__()
            self.initial = nn.Sequential(L.ReflectionPad2d(3), L.Conv2d(3, ngf, 7, padding=0), L.InstanceNorm2d(ngf), nn.ReLU(True))
            self.downsample = nn.Sequential(L.Conv2d(ngf, 2 * ngf, 3, stride=2, padding=1), L.InstanceNorm2d(2 * ngf), nn.ReLU(True),
                                            L.Conv2d(2 * ngf, 4 * ngf, 3, stride=2, padding=1), L.InstanceNorm2d(4 * ngf), nn.ReLU(True))
            self.res_blocks = nn.ModuleList([ResidualBlock(4 * ngf) for _ in range(n_blocks)])
            self.upsample = nn.Sequential(L.ConvTranspose2d(4 * ngf, 2 * ngf, 3, stride=2, padding=1, output_padding=1), L.InstanceNorm2d(2 * ngf), nn.ReLU(True),
                                          L.ConvTranspose2d(2 * ngf, ngf, 3, stride=2, padding=1, output_padding=1), L.InstanceNorm2d(ngf), nn.ReLU(True))
            self.output = nn.Sequential(L.ReflectionPad2d(3), L.Conv2d(ngf, 3, 7, padding=0), nn.Tanh())

        def forward(self, x):
            x = self.downsample(self.initial(x))
            for blk in self.res_blocks:
                x = blk(x)
            return self.output(self.upsample(x))
    return Gen()


def build_layerwise_discriminator(L, ndf=64):
    """PatchGANDiscriminator (discriminator_patchgan.py:26-54: 4x4 convs, zero pad 1, LeakyReLU(0.2) after the first four), one scale."""
    class D(nn.Module):
        def __init__(self):
            super().__init__()
            self.model = nn.Sequential(L.Conv2d(3, ndf, 4, 2, 1), nn.LeakyReLU(0.2, True), L.Conv2d(ndf, 2 * ndf, 4, 2, 1), nn.LeakyReLU(0.2, True),
                                       L.Conv2d(2 * ndf, 4 * ndf, 4, 2, 1), nn.LeakyReLU(0.2, True), L.Conv2d(4 * ndf, 8 * ndf, 4, 1, 1), nn.LeakyReLU(0.2, True),
                                       L.Conv2d(8 * ndf, 1, 4, 1, 1))

        def forward(self, x):
            return self.model(x)
    return D()


def layerwise_cases(device, tol):
    from gan_variant_research_amd import ops_library as L
    g = np.load(GOLDEN)
    C.set_seed(42)
    gen, disc = C.build_models(cases.small_config(), "cpu")       # same init as the reference (golden init KATs pin it)
    G = build_layerwise_generator(L)
    assert list(G.state_dict()) == list(gen.state_dict())          # the reference's key contract
    G.load_state_dict(gen.state_dict())
    G.to(device)
    D = build_layerwise_discriminator(L)
    D.load_state_dict({k.replace("discriminators.0.", ""): v for k, v in disc.state_dict().items()})
    D.to(device)
    x = torch.from_numpy(g["x64"]).to(device)
    y = G(x)
    assert y.requires_grad
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["G64"], rtol=tol, atol=tol)
    np.testing.assert_allclose(D(x).detach().cpu().numpy(), g["D64"], rtol=tol, atol=tol)
    # gradients of a scalar loss through every op's autograd formula vs the oracle under torch autograd (small image: the emulator is slow)
    xs = x[:1, :, :32, :32].clone().requires_grad_(True)
    gp = {k: v.detach().clone().requires_grad_(True) for k, v in gen.state_dict().items()}
    tgt = torch.from_numpy(g["x64"])[:1, :, :32, :32].flip(3)
    loss_ref = (cut_ref.generator_forward(gp, xs.detach().cpu().requires_grad_(False)) - tgt).abs().mean()
    loss_ref.backward()
    loss = (G(xs) - tgt.to(device)).abs().mean()
    loss.backward()
    np.testing.assert_allclose(float(loss), float(loss_ref), rtol=tol)
    assert xs.grad is not None and torch.isfinite(xs.grad).all()
    for k, p in G.named_parameters():
        ref = gp[k].grad
        scale = float(ref.abs().max()) + 1e-12
        if k.endswith(".bias") and ("res_blocks" in k or "initial" in k or "downsample" in k or "upsample" in k):
            continue        # a bias in front of a non-affine InstanceNorm: gradient identically zero, both sides hold rounding noise
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref.numpy(), rtol=0, atol=20 * tol * scale, err_msg=k)


def fused_adam_case(device):
    """mi355x_gan::fused_clip_adam_ema_ vs torch.optim.Adam + clip_grad_norm_ + the reference's EMA.update, three steps."""
    from gan_variant_research_amd import ops_library as L  # noqa: F401  (registers the ops)
    torch.manual_seed(3)
    shapes = [(8, 4, 3, 3), (8,), (5, 7)]
    ps = [torch.randn(s, device=device) for s in shapes]
    ref = [p.detach().cpu().clone().requires_grad_(True) for p in ps]
    opt = torch.optim.Adam(ref, lr=2e-4, betas=(0.5, 0.999), eps=1e-8)
    m, v = [torch.zeros_like(p) for p in ps], [torch.zeros_like(p) for p in ps]
    ema, ema_ref = [p.clone() for p in ps], [p.detach().cpu().clone() for p in ps]
    steps = torch.zeros(len(ps), dtype=torch.int32, device=device)
    gs = [torch.zeros_like(p) for p in ps]
    for it in range(3):
        grads = [torch.randn(s) * (5.0 if it == 0 else 0.1) for s in shapes]
        for r, gr in zip(ref, grads):
            r.grad = gr.clone()
        want_norm = torch.nn.utils.clip_grad_norm_(ref, 10.0)
        opt.step()
        for e, r in zip(ema_ref, ref):
            e.mul_(0.999).add_(r.detach(), alpha=0.001)
        for dst, gr in zip(gs, grads):
            dst.copy_(gr)
        fresh = [gr.clone().to(device) for gr in grads]       # new gradient tensors every step, as zero_grad(set_to_none=True) leaves them
        norm, found_inf = torch.ops.mi355x_gan.fused_clip_adam_ema_(ps, fresh, m, v, ema, steps, 2e-4, 0.5, 0.999, 1e-8, 10.0, 1.0, 0.999)
        np.testing.assert_allclose(float(norm), float(want_norm), rtol=1e-5)
        assert float(found_inf) == 0.0
        for p, r, e, er in zip(ps, ref, ema, ema_ref):
            np.testing.assert_allclose(p.cpu().numpy(), r.detach().numpy(), rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(e.cpu().numpy(), er.numpy(), rtol=1e-5, atol=1e-7)
    assert steps.tolist() == [3, 3, 3]
    from gan_variant_research_amd import training as T
    assert len(T._FUSED_PLANS) == 1, "one prebuilt launch per optimiser state, whatever the gradients' addresses"
    # GradScaler semantics through the op: an overflow skips the step and reports found_inf
    bad = [g.clone() for g in fresh]
    bad[0].view(-1)[0] = float("inf")
    before = [p.clone() for p in ps]
    inv = torch.full((1,), 0.5, device=device)
    norm, found_inf = torch.ops.mi355x_gan.fused_clip_adam_ema_(ps, bad, m, v, ema, steps, 2e-4, 0.5, 0.999, 1e-8, 10.0, 1.0, 0.999, inv, True)
    assert float(found_inf) == 1.0 and steps.tolist() == [3, 3, 3] and all(torch.equal(a, b) for a, b in zip(ps, before))


def new_op_cases(device, tol):
    """torch.ops.mi355x_gan.patchnce_fwd / _bwd and diffaugment_fwd / _bwd, called DIRECTLY, against the reference's own values
    (tests/golden/cut_losses.npz: PatchNCELoss._compute_nce_loss and DiffAugment of the imported reference, values and gradients), and
    allreduce_bucket_ in a single process (no group: the bucket is returned untouched)."""
    import os
    from gan_variant_research_amd import cut as C
    from gan_variant_research_amd import ops_library as L  # noqa: F401
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "cut_losses.npz"))
    for tag in ("a", "b", "c"):
        src = torch.tensor(g[f"nce.{tag}.src"]).to(device)
        tgt = torch.tensor(g[f"nce.{tag}.tgt"]).to(device).requires_grad_(True)
        ids = torch.tensor(g[f"nce.{tag}.ids"]).to(device)
        loss, saved = torch.ops.mi355x_gan.patchnce_fwd(src, tgt, ids, 0.07)
        np.testing.assert_allclose(float(loss.detach()), float(g[f"nce.{tag}.loss"]), rtol=tol)
        (gt,) = torch.autograd.grad(loss * 1.0, tgt)
        want = g[f"nce.{tag}.gtgt"]
        np.testing.assert_allclose(gt.cpu().numpy(), want, rtol=1e-3, atol=1e-3 * float(np.abs(want).max()))     # fp32 softmax of 256 logits, summation order
        np.testing.assert_allclose(torch.ops.mi355x_gan.patchnce_bwd(saved.detach(), torch.tensor(2.0, device=device)).cpu().numpy(), 2.0 * gt.cpu().numpy(), rtol=1e-6, atol=1e-9)
    # DiffAugment: the draws of the reference's run (global generator seeded with aug.seed), then the op on the parameter table
    x = torch.tensor(g["aug.x"]).to(device).requires_grad_(True)
    aug = C.DiffAugment(["color", "translation", "cutout"])
    torch.manual_seed(int(g["aug.seed"]))
    draws = aug.sample(3, 32, 32, None)
    prm = aug.to_params(draws, 3, 32, 32).to(device)
    y = torch.ops.mi355x_gan.diffaugment_fwd(x, prm)
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["aug.y"], rtol=0, atol=max(tol, 2e-6))
    (gx,) = torch.autograd.grad((y * torch.tensor(g["aug.w"]).to(device)).sum(), x)
    np.testing.assert_allclose(gx.cpu().numpy(), g["aug.gx"], rtol=0, atol=max(tol, 2e-6) * 10)
    np.testing.assert_allclose(torch.ops.mi355x_gan.diffaugment_bwd(torch.tensor(g["aug.w"]).to(device), prm).cpu().numpy(), g["aug.gx"], rtol=0, atol=max(tol, 2e-6) * 10)
    flat = torch.arange(8, dtype=torch.float32, device=device)
    out = torch.ops.mi355x_gan.allreduce_bucket_(flat, "")
    assert out.data_ptr() == flat.data_ptr() and flat.tolist() == list(range(8))


def test_new_ops_on_emulator(monkeypatch):
    from gan_variant_research_amd import losses as LS
    monkeypatch.setattr(AG, "_OPS_FACTORY", lambda device: EmuOps())
    monkeypatch.setattr(LS, "_PLANS", {})
    new_op_cases(torch.device("cpu"), 2e-5)


def pad_op_cases(device):
    """replication_pad2d / reflection_pad2d and their gradients against torch.nn.functional.pad: ragged sizes, pads 1..3, maps too
    small for the streaming fold (H < 2*pad+2: the general gan_pad_fold takes over)."""
    import torch.nn.functional as F
    from gan_variant_research_amd import ops_library as L  # noqa: F401
    torch.manual_seed(5)
    for (B, Cc, H, W), pad in (((2, 8, 9, 7), 1), ((1, 24, 6, 5), 3), ((2, 3, 16, 16), 2), ((1, 8, 4, 4), 1)):
        for name, mode in (("replication_pad2d", "replicate"), ("reflection_pad2d", "reflect")):
            x = torch.randn(B, Cc, H, W, device=device, requires_grad=True)
            y = getattr(torch.ops.mi355x_gan, name)(x, pad)
            xr = x.detach().cpu().requires_grad_(True)
            yr = F.pad(xr, (pad,) * 4, mode=mode)
            assert torch.equal(y.detach().cpu(), yr.detach()), (name, pad)
            w = torch.randn(yr.shape)
            g, = torch.autograd.grad((y * w.to(device)).sum(), x)
            gr, = torch.autograd.grad((yr * w).sum(), xr)
            np.testing.assert_allclose(g.cpu().numpy(), gr.numpy(), rtol=1e-6, atol=1e-6, err_msg=f"{name} pad {pad}")


def test_pad_ops_on_emulator(monkeypatch):
    monkeypatch.setattr(AG, "_OPS_FACTORY", lambda device: EmuOps())
    pad_op_cases("cpu")


def test_ops_are_registered_with_schemas():
    from gan_variant_research_amd import ops_library as L  # noqa: F401
    for name in ("conv2d_fwd", "conv2d_dgrad", "conv2d_wgrad", "conv_transpose2d_fwd", "conv_transpose2d_dgrad", "conv_transpose2d_wgrad",
                 "instance_norm_fwd", "instance_norm_bwd", "reflection_pad2d", "reflection_pad2d_bwd", "replication_pad2d", "replication_pad2d_bwd",
                 "fused_clip_adam_ema_", "patchnce_fwd", "patchnce_bwd", "diffaugment_fwd", "diffaugment_bwd", "allreduce_bucket_"):
        op = getattr(torch.ops.mi355x_gan, name)
        assert "Tensor" in str(op.default._schema), name


def test_layerwise_models_on_emulator(monkeypatch):
    from gan_variant_research_amd import ops_library as L
    monkeypatch.setattr(AG, "_OPS_FACTORY", lambda device: EmuOps())
    monkeypatch.setattr(L, "_PLANS", {})
    layerwise_cases(torch.device("cpu"), 2e-4)


def test_fused_adam_op_on_emulator(monkeypatch):
    from gan_variant_research_amd import training as T
    monkeypatch.setattr(AG, "_OPS_FACTORY", lambda device: EmuOps())
    monkeypatch.setattr(T, "_FUSED_PLANS", {})
    fused_adam_case(torch.device("cpu"))
