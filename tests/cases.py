"""Shared test bodies, parametrised over the op backend: EmuOps on CPU (host-logic tests) or HipOps on a MI355X
(parity tests proper, through the C ABI)."""
import numpy as np
import torch
import torch.nn.functional as F

from gan_variant_research_amd import BF16, F32
from gan_variant_research_amd import cut as C
from gan_variant_research_amd.convplan import ConvLayer
from gan_variant_research_amd.runtime import Ctx, cpad
from oracle import cut_ref
from tests.emulator import HALO_REFLECT, HALO_ZERO


def to_view(ctx, x, halo, mode):
    B, Cc, H, W = x.shape
    v = ctx.view(B, H, W, cpad(Cc), halo)
    ctx.ops.nchw_to_view(x.contiguous().to(ctx.device), Cc, v, mode)()
    return v


def from_view(v, Cc):
    return v.nhwc().float()[..., :Cc].permute(0, 3, 1, 2).contiguous().cpu()


GEOMS = [  # cin, cout, k, s, p, transposed, H, reflect
    (3, 16, 7, 1, 3, False, 12, True),
    (16, 3, 7, 1, 3, False, 12, True),
    (16, 32, 3, 2, 1, False, 12, False),
    (32, 32, 3, 1, 1, False, 8, True),
    (32, 16, 3, 2, 1, True, 6, False),
    (3, 16, 4, 2, 1, False, 12, False),
    (16, 32, 4, 1, 1, False, 9, False),
    (32, 1, 4, 1, 1, False, 8, False),
    # full-width channels of the real networks (GPU tile paths: 128x128, 128x64, 256x16; multi-tile N, M tails)
    (256, 256, 3, 1, 1, False, 16, True),
    (64, 128, 3, 2, 1, False, 16, False),
    (256, 128, 3, 2, 1, True, 8, False),
    (128, 64, 3, 2, 1, True, 8, False),
    (64, 3, 7, 1, 3, False, 16, True),
    (256, 512, 4, 1, 1, False, 9, False),
    (512, 1, 4, 1, 1, False, 8, False),
    (128, 256, 4, 2, 1, False, 16, False),
    # the 64 <-> 3 channel 7x7 layers on multi-tile images with ragged edges (7x7 window kernel: forward of 64->3, input gradient of 3->64)
    (64, 3, 7, 1, 3, False, 37, True),
    (3, 64, 7, 1, 3, False, 21, True),
    # 128-pixel-wide maps (the residual layers of 512x512 images): 9-slice range-patch buffers, row-ring weight gradient
    (128, 128, 3, 1, 1, False, 128, True),
]


def run_conv_geometry(ctx, geom, dtype, B=2):
    cin, cout, k, s, p, tr, H, reflect = geom
    torch.manual_seed(0)
    w = (torch.randn((cin, cout, k, k) if tr else (cout, cin, k, k)) * (0.5 / (cin * k * k) ** 0.5)).to(ctx.device)
    b = (torch.randn(cout) * 0.1).to(ctx.device)
    gw, gb = torch.zeros_like(w), torch.zeros_like(b)
    layer = ConvLayer(ctx, w, b, gw, gb, k, s, p, tr)
    x = torch.randn(B, cin, H, H)
    if dtype == BF16:
        x, w_ref = x.bfloat16().float(), w.cpu().bfloat16().float()
    else:
        w_ref = w.cpu()
    xr = x.clone().requires_grad_(True)
    wr = w_ref.clone().requires_grad_(True)
    br = b.cpu().clone().requires_grad_(True)
    if tr:
        y_ref = F.conv_transpose2d(xr, wr, br, stride=2, padding=1, output_padding=1)
    else:
        xp = F.pad(xr, (p, p, p, p), mode="reflect") if reflect else F.pad(xr, (p, p, p, p))
        y_ref = F.conv2d(xp, wr, br, stride=s)
    Ho = y_ref.shape[2]
    xin = to_view(ctx, x, max(p, 1), HALO_REFLECT if reflect else HALO_ZERO)
    y = ctx.view(B, Ho, Ho, cpad(cout), 0)
    fwd_ops = layer.fwd(xin, y)
    for op in layer.repack_ops():   # operand copies are allocated when a call is planned: plan first, then pack
        op()
    for op in fwd_ops:
        op()
    tol = dict(rtol=1e-4, atol=1e-4) if dtype == F32 else dict(rtol=3e-2, atol=6e-2)
    np.testing.assert_allclose(from_view(y, cout).numpy(), y_ref.detach().numpy(), **tol)
    # backward
    gy = torch.randn_like(y_ref)
    if dtype == BF16:
        gy = gy.bfloat16().float()
    y_ref.backward(gy)
    if tr:
        dyv = to_view(ctx, gy, 1, HALO_ZERO)
        dx = ctx.view(B, H, H, cpad(cin), 0)
        ops = layer.dgrad(dyv, dx)
        fold = False
    elif s == 2:
        dyv = to_view(ctx, gy, 1, HALO_ZERO)
        dx = ctx.view(B, H, H, cpad(cin), 0)
        ops, fold = layer.dgrad(dyv, dx), False
    elif reflect:
        dyv = to_view(ctx, gy, k - 1, HALO_ZERO)
        dx = ctx.view(B, H, H, cpad(cin), p)
        ops, fold = layer.dgrad(dyv, dx, padded_domain=True), True
    else:
        dyv = to_view(ctx, gy, k - 1 - p, HALO_ZERO)
        dx = ctx.view(B, H, H, cpad(cin), 0)
        ops, fold = layer.dgrad(dyv, dx), False
    for op in layer.repack_ops():
        op()
    for op in ops:
        op()
    out = ctx.view(B, H, H, cpad(cin), 0)
    ctx.ops.fold_add(None, dx, fold, out)()
    np.testing.assert_allclose(from_view(out, cin).numpy(), xr.grad.numpy(), **tol)
    for op in layer.wgrad(xin, dyv, accumulate=False):
        op()
    wtol = dict(rtol=1e-3, atol=1e-3) if dtype == F32 else dict(rtol=3e-2, atol=3e-1)
    if ctx.device.type == 'cuda':
        torch.cuda.synchronize()
    np.testing.assert_allclose(gw.cpu().numpy(), wr.grad.numpy(), **wtol)
    np.testing.assert_allclose(gb.cpu().numpy(), br.grad.numpy(), **wtol)




def small_config():
    cfg = cut_ref.default_config()
    cfg["model"] = {"generator": {"ngf": 64, "n_blocks": 9, "n_downsampling": 2, "padding_type": "reflect", "norm": "instance", "activation": "relu"},
                    "discriminator": {"ndf": 64, "n_layers": 3, "num_scales": 1, "use_spectral_norm": False}}
    cfg["optim"] = {"G": {"lr": 2e-4, "betas": [0.5, 0.999], "weight_decay": 0.0}, "D": {"lr": 2e-4, "betas": [0.5, 0.999], "weight_decay": 0.0}}
    cfg["amp"] = False
    return cfg




def run_cut_steps(device, ops, use_aug, amp=False, S=32, B=2, nsteps=2, tol0=2e-4, tol1=2e-3, atol1=2e-4, ptol=4.5e-4, fp8=False, threads=4):
    torch.set_num_threads(threads)
    cfg = small_config()
    cfg["diffaugment"]["enable"] = use_aug
    C.set_seed(42)
    gen, disc = C.build_models(cfg, "cpu")
    tr = C.CutTrainer(gen, disc, cfg, B, S, device=device, amp=amp, ops=ops, fp8=fp8)
    cut_ref.set_seed(42)
    gp, dp = cut_ref.init_generator(), cut_ref.init_discriminator()
    og, od = cut_ref.AdamState(gp), cut_ref.AdamState(dp)
    ema = {k: v.detach().clone() for k, v in gp.items()}
    g = torch.Generator().manual_seed(1234)
    photos = torch.rand(B, 3, S, S, generator=g) * 2 - 1
    monets = torch.rand(B, 3, S, S, generator=g) * 2 - 1
    for step in range(nsteps):
        torch.manual_seed(9000 + step)
        rnd = cut_ref.sample_step_randomness(B, S, S, use_aug=use_aug)
        torch.manual_seed(9000 + step)
        rnd2 = tr.sample_randomness()
        for a, b in zip(rnd["nce_ids"], rnd2["nce_ids"]):
            assert torch.equal(a, b)
        ref = cut_ref.train_step(step, photos, monets, gp, dp, og, od, ema, cfg, rnd)
        g_before = {k: v.detach().cpu().clone() for k, v in tr.opt_G.params.items()}
        got = tr.train_step(step, photos.to(device), monets.to(device), rnd2)
        for k in ref:
            # step 1: Adam's sign-like first update makes zero-gradient parameters differ by +-lr; g_adv is a
                # mean of O(1) logits that happens to be ~1e-3, so it gets an absolute tolerance
                # g_adv is a mean of O(1) logits that is itself ~1e-2: absolute tolerance on the logit scale.  (At step 0 the last D bias has an
                # exactly-zero gradient, so Adam's sign-like first update moves it by +-lr on rounding noise alone: SURVEY.md §7.2.)
                atol = max(tol0 * 0.1 if step == 0 else atol1, 1e-3 if k == "g_adv" else 0.0)
                np.testing.assert_allclose(got[k], ref[k], rtol=tol0 if step == 0 else tol1, atol=atol, err_msg=f"step{step} {k}")
        if step == 0:
            for k in gp:
                np.testing.assert_allclose(tr.opt_G.params[k].cpu().numpy(), gp[k].detach().numpy(), rtol=0, atol=ptol, err_msg=k)
            for k in dp:
                np.testing.assert_allclose(tr.opt_D.params[k].cpu().numpy(), dp[k].detach().numpy(), rtol=0, atol=2 * ptol, err_msg=k)  # two D updates (D-step + R1)
    # G(photos) of the last step was computed with the weights before that step's update
    img = tr.generated().cpu()
    ref_img = cut_ref.generator_forward(g_before, photos).detach()
    return tr, img, ref_img


def basic_config():
    """Basic_GAN/configs/baseline.yaml (the keys the loop reads)."""
    return {"training": {"amp": False, "seed": 0}, "optim": {"lr_g": 2e-4, "lr_d": 2e-4, "betas": [0.5, 0.999]},
            "loss": {"gan": "lsgan", "lambda_cycle": 10.0, "lambda_identity": 0.5},
            "model": {"ngf": 64, "ndf": 64, "n_blocks": 9, "spectral_norm_d": False}}


def run_basic_iterations(device, ops, amp=False, S=32, B=2, niter=2, tol0=2e-4, tol1=2e-3):
    """CycleGAN inner loop (Basic_GAN/src/train.py:66-122) on the engine vs the oracle."""
    from gan_variant_research_amd import basic as BG
    from oracle import basic_ref
    cfg = basic_config()
    cfg["training"]["amp"] = amp
    torch.manual_seed(0)
    mods = BG.build_models(cfg, "cpu")
    torch.manual_seed(0)
    gab, gba = basic_ref.init_generator(), basic_ref.init_generator()
    da, db = basic_ref.init_discriminator(), basic_ref.init_discriminator()
    for m, ref in zip(mods, (gab, gba, da, db)):
        sd = m.state_dict()
        assert list(sd) == list(ref)
        for k in ref:
            assert torch.equal(sd[k], ref[k]), k
    tr = BG.CycleGANTrainer(*[m.to(device) for m in mods], cfg, B, S, device=device, amp=amp, ops=ops)
    both = {**{"ab." + k: v for k, v in gab.items()}, **{"ba." + k: v for k, v in gba.items()}}
    og, oa, ob = cut_ref.AdamState(both), cut_ref.AdamState(da), cut_ref.AdamState(db)
    g = torch.Generator().manual_seed(77)
    a = torch.rand(B, 3, S, S, generator=g) * 2 - 1
    b = torch.rand(B, 3, S, S, generator=g) * 2 - 1
    for it in range(niter):
        ref = basic_ref.train_iteration(a, b, gab, gba, da, db, og, oa, ob)
        got = tr.train_iteration(a.to(device), b.to(device))
        for k in ref:
            np.testing.assert_allclose(got[k], ref[k], rtol=tol0 if it == 0 else tol1, atol=1e-5, err_msg=f"it{it} {k}")
    return tr


def run_conv_fp8(ctx, B=2, C_=256, H=16, seed=0, with_stats=False, info=None):
    """The fp8 (e4m3) operand path of the bottleneck 3x3 256->256 convolution (BASELINE.json configs[4]): forward and input gradient
    through gan_quantize_fp8 + gan_pack_weight_batch(FP8) + gan_conv_igemm(dtype FP8) against
      (a) F.conv2d on the SAME e4m3-rounded operands (what the kernel must compute exactly, up to fp32 summation order and the bf16
          rounding of the result: rtol 1e-2 of the output's rms), and
      (b) F.conv2d on the unrounded operands with the fp8 tolerance: e4m3 keeps 3 mantissa bits (relative rounding error <= 2^-4 per
          operand), which over K = 2304 random-sign products is ~4 % of the output's rms -> bound 0.12 * rms elementwise."""
    from gan_variant_research_amd import FP8
    from tests.emulator import _e4m3
    dev = ctx.device
    g = torch.Generator().manual_seed(seed)
    w = (torch.rand(C_, C_, 3, 3, generator=g) * 2 - 1) / (C_ * 9) ** 0.5
    b = torch.randn(C_, generator=g) * 0.1
    x = torch.randn(B, C_, H, H, generator=g).bfloat16().float()
    layer = ConvLayer(ctx, w.to(dev), b.to(dev), torch.zeros_like(w).to(dev), torch.zeros_like(b).to(dev), 3, 1, 1)
    xv = to_view(ctx, x, 1, HALO_REFLECT)
    x8 = ctx.view(B, H, H, C_, 1, dtype=FP8)
    y = ctx.view(B, H, H, C_, 0)
    stats_ws = ctx.f32(B * 96 * C_ * 2) if with_stats else None
    ops_f = [ctx.ops.quantize_fp8(xv, x8)] + layer.fwd8(x8, y, stats_ws=stats_ws)
    calls = {"fwd": getattr(ops_f[-1], "conv", None)}
    # input gradient: dY with a per-image scale, zero halo 2, padded-domain result folded by the consumer
    dy = torch.randn(B, C_, H, H, generator=g) * torch.tensor([1e-3, 3e-5])[:B].view(B, 1, 1, 1)
    dy = dy.bfloat16().float()
    dyv = to_view(ctx, dy, 2, HALO_ZERO)
    dy8 = ctx.view(B, H, H, C_, 2, dtype=FP8)
    amax = dy.abs().amax((1, 2, 3)).to(dev)
    scale = torch.zeros(B, device=dev)
    dxp = ctx.view(B, H, H, C_, 1)
    dxf = ctx.view(B, H, H, C_, 0)
    ops_d = layer.dgrad8(dy8, dxp, scale, padded_domain=True)
    calls["dgrad"] = getattr(ops_d[-1], "conv", None)
    ops_b = [ctx.ops.quantize_fp8(dyv, dy8, amax, scale)] + ops_d + [ctx.ops.fold_add(None, dxp, True, dxf)]
    for o in [ctx.ops.pack_weight_batch([op.pack_args for op in layer.repack_ops()])] + ops_f + ops_b:
        o()
    if ctx.device.type == "cuda":
        torch.cuda.synchronize()
    if info is not None:
        info.update(calls)
    sw = float(w.abs().max()) / 448.0
    w8 = _e4m3(w / sw) * sw
    xp = F.pad(x, (1, 1, 1, 1), mode="reflect")
    got = from_view(y, C_)
    exact = F.conv2d(_e4m3(xp), w8, b)
    full = F.conv2d(xp, w, b)
    rms = float(full.pow(2).mean().sqrt())
    assert float((got - exact).abs().max()) < 1e-2 * rms * 4, (float((got - exact).abs().max()), rms)
    assert float((got - full).abs().max()) < 0.12 * rms * 2.5 and float((got - full).pow(2).mean().sqrt()) < 0.05 * rms
    np.testing.assert_allclose(scale.cpu().numpy(), (amax.cpu() / 448.0).numpy(), rtol=1e-6)
    if with_stats:      # the fused InstanceNorm partials of the e4m3 forward: statistics of the (unrounded) result
        assert layer.stats_parts > 0
        st = ctx.f32(B * C_ * 2)
        ctx.ops.in_stats_from_parts(stats_ws, layer.stats_parts, B, C_, H * H, 1e-5, st)()
        if ctx.device.type == "cuda":
            torch.cuda.synchronize()
        sg = st.cpu().view(B, C_, 2)
        np.testing.assert_allclose(sg[..., 0].numpy(), exact.mean((2, 3)).numpy(), rtol=2e-3, atol=2e-3)
        np.testing.assert_allclose(sg[..., 1].numpy(), (1.0 / torch.sqrt(exact.var((2, 3), unbiased=False) + 1e-5)).numpy(), rtol=2e-3)
    # reference input gradient through autograd on the reflect-padded convolution
    xr = x.clone().requires_grad_(True)
    F.conv2d(F.pad(xr, (1, 1, 1, 1), mode="reflect"), w, None).backward(dy)
    gotd = from_view(dxf, C_)
    for bi in range(B):
        r = float(xr.grad[bi].pow(2).mean().sqrt())
        assert float((gotd[bi] - xr.grad[bi]).pow(2).mean().sqrt()) < 0.06 * r, (bi, float((gotd[bi] - xr.grad[bi]).pow(2).mean().sqrt()), r)
        assert float((gotd[bi] - xr.grad[bi]).abs().max()) < 0.3 * r
    return got, gotd


def run_basic_checkpoint_case(device, make_ops, tmp_path, S=32, B=2):
    """f-1 for the CycleGAN trainer (Basic_GAN/src/train.py:27-31,54-58,124-137): two iterations, epoch end (scheduler step + checkpoint in
    the reference's dict layout), a third iteration -- against a fresh trainer that loads the checkpoint and runs the third iteration:
    identical losses and parameters.  The checkpoint must load into the reference-named modules and into torch.optim.Adam."""
    from gan_variant_research_amd import basic as BG
    cfg = basic_config()
    cfg["training"].update({"epochs": 4, "save_every": 1})
    cfg["optim"]["lr_decay_after"] = 1
    g = torch.Generator().manual_seed(77)
    a = (torch.rand(B, 3, S, S, generator=g) * 2 - 1).to(device)
    b = (torch.rand(B, 3, S, S, generator=g) * 2 - 1).to(device)

    def fresh():
        torch.manual_seed(0)
        mods = BG.build_models(cfg, "cpu")
        return BG.CycleGANTrainer(*[m.to(device) for m in mods], cfg, B, S, device=device, amp=False, ops=make_ops()), mods
    tr, mods = fresh()
    for _ in range(2):
        tr.train_iteration(a, b)
    assert tr.scheduler_step() == 1.0                         # epoch 1 < lr_decay_after + ... : lambda_rule(1, 1, 4) = 1
    lam = tr.scheduler_step()                                 # lambda_rule(2, 1, 4) = 2/3
    assert abs(lam - 2.0 / 3.0) < 1e-12 and abs(tr.opt_G.lr - 2e-4 * lam) < 1e-12
    path = str(tmp_path / "ckpt_e2.pt")
    ck = tr.save_checkpoint(path, epoch=2)
    # the reference's layout, key for key (train.py:127-137)
    assert list(ck) == ["epoch", "G_A2B", "G_B2A", "D_A", "D_B", "optim_G", "optim_D_A", "optim_D_B"] and ck["epoch"] == 2
    loaded = torch.load(path, map_location="cpu", weights_only=True)
    G1, G2, D1, D2 = BG.build_models(cfg, "cpu")
    for key, mod in (("G_A2B", G1), ("G_B2A", G2), ("D_A", D1), ("D_B", D2)):
        assert list(loaded[key]) == list(mod.state_dict())
        mod.load_state_dict(loaded[key])
    # ... and torch.optim.Adam takes the optimiser states (what a user of the reference would do with the file)
    ref_opt = torch.optim.Adam(list(G1.parameters()) + list(G2.parameters()), lr=2e-4, betas=(0.5, 0.999))
    ref_opt.load_state_dict(loaded["optim_G"])
    g0 = ref_opt.param_groups[0]
    assert abs(g0["lr"] - 2e-4 * lam) < 1e-12 and g0["initial_lr"] == 2e-4 and len(ref_opt.state) == len(list(G1.parameters())) * 2
    p0 = list(G1.parameters())[0]
    assert float(ref_opt.state[p0]["step"]) == 2.0
    np.testing.assert_array_equal(ref_opt.state[p0]["exp_avg"].numpy(), tr.opt_G.flat_m[:p0.numel()].cpu().view(p0.shape).numpy())
    # continuation: original vs resumed
    l_orig = tr.train_iteration(a, b)
    tr2, _ = fresh()
    assert tr2.load_checkpoint(path) == 2 and tr2.sched_epoch == 2 and abs(tr2.opt_G.lr - 2e-4 * lam) < 1e-12
    l_res = tr2.train_iteration(a, b)
    assert l_orig == l_res, (l_orig, l_res)
    for o1, o2 in ((tr.opt_G, tr2.opt_G), (tr.opt_DA, tr2.opt_DA), (tr.opt_DB, tr2.opt_DB)):
        assert torch.equal(o1.flat_p, o2.flat_p) and torch.equal(o1.flat_m, o2.flat_m) and torch.equal(o1.steps, o2.steps)
    # the decayed rate really drives the update: Adam's third step moves a weight by ~lr * lam (sign-like early steps)
    tr3, _ = fresh()
    tr3.load_checkpoint(path)
    tr3.opt_G.set_lr(2e-4)
    before = tr3.opt_G.flat_p.clone()
    tr3.train_iteration(a, b)
    full = (tr3.opt_G.flat_p - before).abs().max()
    dec = (tr2.opt_G.flat_p - before).abs().max()
    assert abs(float(dec / full) - lam) < 0.02, float(dec / full)
    return tr
