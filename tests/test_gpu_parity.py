"""Parity tests proper (MI355X, through the C ABI): every HIP kernel against its CPU statement / the oracle."""
import os

import numpy as np
import pytest
import torch

from gan_variant_research_amd import BF16, F32
from gan_variant_research_amd.runtime import Ctx, HipOps, View
from tests import cases
from tests.emulator import EmuOps

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def hip_ctx(dtype):
    return Ctx(HipOps(torch.device(DEV)), DEV, dtype)


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("geom", cases.GEOMS)
def test_conv_geometry_hip(geom, dtype):
    cases.run_conv_geometry(hip_ctx(dtype), geom, dtype, B=3)


@pytest.mark.parametrize("geom", [g for g in cases.GEOMS if max(g[0], g[1]) >= 64])
def test_conv_geometry_hip_tile288(geom, monkeypatch):
    """The 288-row tile of the range-patch kernel is only chosen on large launches (tile-count quantisation); force it."""
    monkeypatch.setenv("GAN_PATCH_BM", "288")
    cases.run_conv_geometry(hip_ctx(BF16), geom, BF16, B=3)


WIDE_MAP_256 = (256, 256, 3, 1, 1, False, 128, True)     # a residual layer of a 512x512 image: 9-slice buffers (256-row tile only)
# sub-pixel phases on a 256-pixel-wide map (the second up-sampling layer of a 512x512 image): the 2- and 4-tap static schedules on the 9-slice
# buffers (a 256-pixel tile plus one row and one pixel spans 515 pixels; at 128 pixels width it fits the 7-slice buffers)
WIDE_PHASES = [(128, 64, 3, 2, 1, True, 256, False),      # paired x-phases: 2 taps in a row (7 slices) and 4 taps (9 slices)
               (256, 128, 3, 2, 1, True, 256, False)]     # unpaired phases: the 2-tap phase with a vertical tap spans 514 pixels (9 slices)


def _static_taps_of(ctx, geom, B):
    """Plans forward and input gradient of one geometry (no launch) and returns the static tap schedules the planner chose."""
    from gan_variant_research_amd.convplan import ConvLayer
    cin, cout, k, s, p, tr, H, reflect = geom
    w = torch.zeros((cin, cout, k, k) if tr else (cout, cin, k, k), device=DEV)
    layer = ConvLayer(ctx, w, torch.zeros(cout, device=DEV), torch.zeros_like(w), torch.zeros(cout, device=DEV), k, s, p, tr)
    Ho = 2 * H if tr else (H + 2 * p - k) // s + 1
    x = ctx.view(B, H, H, cases.cpad(cin), max(p, 1))
    y = ctx.view(B, Ho, Ho, cases.cpad(cout), 0)
    ops = list(layer.fwd(x, y))
    if tr or s == 2:
        ops += layer.dgrad(ctx.view(B, Ho, Ho, cases.cpad(cout), 1), ctx.view(B, H, H, cases.cpad(cin), 0))
    out = set()
    for op in ops:
        c = getattr(op, "conv", None)
        if c is not None and c.w_frag:
            v = ctx.ops.conv_patch_variant(c)
            out.add((v.get("static_taps", 0), v.get("slices")))
    return out


def test_static_tap_schedules_are_reached_and_correct():
    """The branch-free tap schedules for 2, 4 and 16 taps (conv_patch_kernel<256, 2, NT>): the geometries of cases.GEOMS that the parity
    tests above run (transposed 256->128 / 128->64 layers, the 4x4 256->512 layer) really plan them, and the 9-slice instantiations
    (256-pixel-wide maps: 512x512 images) are run against F.conv_transpose2d here."""
    ctx = hip_ctx(BF16)
    seen = set()
    for geom in [g for g in cases.GEOMS if g[0] >= 128 and (g[5] or g[2] == 4)]:
        seen |= _static_taps_of(ctx, geom, 3)
    assert {(2, 7), (4, 7), (16, 7)} <= seen, seen
    wide = set()
    for geom in WIDE_PHASES:
        wide |= _static_taps_of(ctx, geom, 1)
    assert {(2, 9), (4, 9)} <= wide, wide
    for geom in WIDE_PHASES:
        cases.run_conv_geometry(ctx, geom, BF16, B=1)



@pytest.mark.parametrize("bm", ["256", "288"])
@pytest.mark.parametrize("geom", [g for g in cases.GEOMS if g[1] % 256 == 0 and g[3] == 1 and not g[5]] + [WIDE_MAP_256])
def test_conv_geometry_hip_tile_cols256(geom, bm, monkeypatch):
    """The 256-channel tiles of the range-patch kernel (whole Cout of the residual layers per tile; chosen on launches of >= 192 such
    tiles) at both tile heights: forced here on the small geometries (3x3 256->256 forward / reflect-padded input gradient, 4x4 256->512)
    and on one 128-pixel-wide map."""
    if geom == WIDE_MAP_256 and bm == "288":
        pytest.skip("maps wider than 64 pixels: 256-row tile only")
    monkeypatch.setenv("GAN_PATCH_BN", "256")
    monkeypatch.setenv("GAN_PATCH_BM", bm)
    ctx = hip_ctx(BF16)
    cases.run_conv_geometry(ctx, geom, BF16, B=3 if geom != WIDE_MAP_256 else 1)
    cin, cout, k = geom[0], geom[1], geom[2]
    from gan_variant_research_amd.runtime import ConvCall  # noqa: F401  (the planner really chose the wide tile)
    from gan_variant_research_amd.convplan import ConvLayer
    w = torch.zeros(cout, cin, k, k, device=DEV)
    layer = ConvLayer(ctx, w, torch.zeros(cout, device=DEV), torch.zeros_like(w), torch.zeros(cout, device=DEV), k, 1, geom[4], False)
    H = geom[6]
    Ho = H + 2 * geom[4] - k + 1
    x, y = ctx.view(3, H, H, cin, max(geom[4], 1)), ctx.view(3, Ho, Ho, cout, 0)
    layer.fwd(x, y)
    assert layer.last_call.tile_cols == 256 and layer.last_call.tile_rows == int(bm)


def test_wgrad_patch_splits_spanning_images():
    """Many small maps (Basic_GAN's residual layers: 16x16 at batch 256): the range-patch weight gradient lets one split accumulate over
    several whole images (gan_wgrad_patch_splits < 0).  Batch 64 is the smallest that triggers it for 256 -> 256 channels (2 images per split)."""
    from gan_variant_research_amd.convplan import ConvLayer
    geom = (256, 256, 3, 1, 1, False, 16, True)
    ctx = hip_ctx(BF16)
    cases.run_conv_geometry(ctx, geom, BF16, B=64)
    w = torch.zeros(256, 256, 3, 3, device=DEV)
    layer = ConvLayer(ctx, w, torch.zeros(256, device=DEV), torch.zeros_like(w), torch.zeros(256, device=DEV), 3, 1, 1, False)
    x, dy = ctx.view(64, 16, 16, 256, 1), ctx.view(64, 16, 16, 256, 2)
    call = layer.wgrad(x, dy, False, bias_too=False)[0].wgrad
    assert call.variant == 1 and call.nsplit == 32, (call.variant, call.nsplit)


@pytest.mark.parametrize("geom", [g for g in cases.GEOMS if g[2] == 7 and max(g[0], g[1]) == 64])
def test_conv_7x7_window_kernel_is_taken(geom):
    """The 64 <-> 3 channel 7x7 layers run on the two window kernels in bf16 (64->3: output conv forward, first conv's input gradient;
    3->64: first conv forward, output conv's input gradient) and stay correct (run_conv_geometry compares with torch);
    GAN_NO_WIN7=1 would send them back to the generic kernel."""
    ctx = hip_ctx(BF16)
    seen, wg = [], []
    orig, orig_w = ctx.ops.conv_igemm, ctx.ops.conv_wgrad
    ctx.ops.conv_igemm = lambda c: (seen.append(c.win7), orig(c))[1]
    ctx.ops.conv_wgrad = lambda c: (wg.append(c.variant), orig_w(c))[1]
    cases.run_conv_geometry(ctx, geom, BF16, B=3)
    assert sum(w is not None for w in seen) == 2, f"forward and input gradient should both run on the 7x7 window kernels: {seen}"
    assert wg == [2], f"the weight gradient runs on the window kernel: {wg}"


# ---------------------------------------------------------------------------------------------- op twins
class Twin:
    """The same buffers on CPU (emulator) and GPU (HIP); ops are built on both and every buffer is compared afterwards."""

    def __init__(self, dtype, seed=0):
        self.dtype = dtype
        self.c, self.g = Ctx(EmuOps(), "cpu", dtype), hip_ctx(dtype)
        self.gen = torch.Generator().manual_seed(seed)
        self.views, self.tensors = [], []

    def view(self, B, H, W, C, halo, rand=True, scale=1.0, zero_halo=False):
        vc = self.c.view(B, H, W, C, halo)
        if rand:
            data = torch.randn(vc.t.shape, generator=self.gen) * scale
            vc.t.copy_(data.to(vc.t.dtype))
            if zero_halo and halo:
                interior = vc.nhwc().clone()
                vc.t.zero_()
                vc.nhwc().copy_(interior)
        vg = self.g.view(B, H, W, C, halo)
        vg.t.copy_(vc.t)
        self.views.append((vc, vg))
        return vc, vg

    def f32(self, data):
        tc = data.clone().float()
        tg = tc.to(DEV)
        self.tensors.append((tc, tg))
        return tc, tg

    def i32(self, data):
        tc = torch.as_tensor(data, dtype=torch.int32)
        return tc, tc.to(DEV)

    def run(self, cpu_ops, gpu_ops):
        for o in cpu_ops if isinstance(cpu_ops, (list, tuple)) else [cpu_ops]:
            o()
        for o in gpu_ops if isinstance(gpu_ops, (list, tuple)) else [gpu_ops]:
            o()
        torch.cuda.synchronize()

    def check(self, rtol, atol):
        for i, (vc, vg) in enumerate(self.views):
            np.testing.assert_allclose(vg.t.float().cpu().numpy(), vc.t.float().numpy(), rtol=rtol, atol=atol, err_msg=f"view {i}")
        for i, (tc, tg) in enumerate(self.tensors):
            np.testing.assert_allclose(tg.cpu().numpy(), tc.numpy(), rtol=rtol, atol=atol, err_msg=f"tensor {i}")


TOL = {F32: (2e-5, 2e-5), BF16: (2e-2, 2e-2)}


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("shape", [(2, 16, 16, 64, 1), (3, 12, 20, 256, 1), (2, 32, 32, 8, 3), (2, 9, 9, 512, 1)])
def test_instance_norm_twins(shape, dtype):
    B, H, W, C, halo = shape
    for act, res, mode in ((1, False, 2), (0, True, 2), (2, False, 1)):
        tw = Twin(dtype, seed=act)
        xc, xg = tw.view(B, H, W, C, 0)
        rc, rg = tw.view(B, H, W, C, halo) if res else (None, None)
        yc, yg = tw.view(B, H, W, C, halo, rand=False)
        sc, sg = tw.f32(torch.zeros(B * C * 2))
        wc, wg = tw.f32(torch.zeros(B * 96 * C * 2 + B * C * 2))
        tw.run([tw.c.ops.in_stats(xc, 1e-5, sc, wc), tw.c.ops.in_apply(xc, sc, act, rc, yc, mode)],
               [tw.g.ops.in_stats(xg, 1e-5, sg, wg), tw.g.ops.in_apply(xg, sg, act, rg, yg, mode)])
        tw.tensors.pop()  # workspace contents are a kernel detail
        tw.check(*TOL[dtype])
        # the same forward with the statistics summed inside the apply pass from <= 16 partials per image (no finalize launch)
        y2c, y2g = tw.view(B, H, W, C, halo, rand=False)
        s2c, s2g = tw.f32(torch.zeros(B * C * 2))
        npc, npg = tw.c.ops.in_partial_count(xc), tw.g.ops.in_partial_count(xg)
        assert 1 <= npg <= 16
        pc, pg = torch.zeros(B * npc * C * 2), torch.zeros(B * npg * C * 2, device=DEV)
        tw.run([tw.c.ops.in_partial(xc, pc), tw.c.ops.in_apply_parts(xc, pc, npc, 1e-5, s2c, act, rc, y2c, mode)],
               [tw.g.ops.in_partial(xg, pg), tw.g.ops.in_apply_parts(xg, pg, npg, 1e-5, s2g, act, rg, y2g, mode)])
        tw.check(*TOL[dtype])
        np.testing.assert_allclose(s2g.cpu().numpy(), sg.cpu().numpy(), rtol=1e-5, atol=1e-6)      # same (mean, rstd) as gan_in_stats
        # backward: folded padded-domain gradient + second addend
        gc, gg = tw.view(B, H, W, C, halo, scale=0.5)
        g2c, g2g = tw.view(B, H, W, C, 0, scale=0.5)
        dc, dg = tw.view(B, H, W, C, 2, rand=False)
        fold = halo if H >= 2 * halo + 2 else 0
        tw.run(tw.c.ops.in_bwd(xc, sc, act, gc, bool(fold), g2c, dc, wc), tw.g.ops.in_bwd(xg, sg, act, gg, bool(fold), g2g, dg, wg))
        tw.check(TOL[dtype][0] * 5, TOL[dtype][1] * 5)
        # the same backward with the fused bias gradient, twice (the second call accumulates)
        nb = max(1, C - 3)
        bc, bg = tw.f32(torch.zeros(nb))
        w2c, w2g = tw.f32(torch.zeros(B * 96 * C * 2 + B * C * 2 + (B * 1024 + 32) * C))
        tw.tensors.pop()
        for acc in (False, True):
            tw.run(tw.c.ops.in_bwd_bias(xc, sc, act, gc, bool(fold), g2c, dc, w2c, bc, nb, acc),
                   tw.g.ops.in_bwd_bias(xg, sg, act, gg, bool(fold), g2g, dg, w2g, bg, nb, acc))
        # column sums of dx are zero in exact arithmetic (non-affine norm): what is compared is rounding noise of B*H*W addends
        np.testing.assert_allclose(bg.cpu().numpy(), bc.numpy(), rtol=TOL[dtype][0] * 5, atol=TOL[dtype][1] * 5 * (B * H * W) ** 0.5)
        tw.tensors.pop()   # bias compared above with a magnitude-aware tolerance
        tw.check(TOL[dtype][0] * 5, TOL[dtype][1] * 5)
        # the deferred variant: partials to a caller-owned buffer, one batched launch sums them (two layers share the launch):
        # bit-identical on the GPU to nothing but itself, so it is compared with the immediate variant's result on the same dx
        npc, npg = tw.c.ops.in_bwd_bias_parts(xc), tw.g.ops.in_bwd_bias_parts(xg)
        pc, pg = torch.zeros(npc * C), torch.zeros(npg * C, device=DEV)
        p2c, p2g = torch.zeros(npc * C), torch.zeros(npg * C, device=DEV)
        b1c, b1g = torch.zeros(nb), torch.zeros(nb, device=DEV)
        b2c, b2g = torch.ones(nb), torch.ones(nb, device=DEV)
        tw.run([tw.c.ops.in_bwd_bias_deferred(xc, sc, act, gc, bool(fold), g2c, dc, w2c, pc), tw.c.ops.in_bwd_bias_deferred(xc, sc, act, gc, bool(fold), g2c, dc, w2c, p2c),
                tw.c.ops.bias_finalize_batch([(pc, npc, C, b1c, nb, False), (p2c, npc, C, b2c, nb, True)])],
               [tw.g.ops.in_bwd_bias_deferred(xg, sg, act, gg, bool(fold), g2g, dg, w2g, pg), tw.g.ops.in_bwd_bias_deferred(xg, sg, act, gg, bool(fold), g2g, dg, w2g, p2g),
                tw.g.ops.bias_finalize_batch([(pg, npg, C, b1g, nb, False), (p2g, npg, C, b2g, nb, True)])])
        atol_b = TOL[dtype][1] * 5 * (B * H * W) ** 0.5
        np.testing.assert_allclose(b1g.cpu().numpy(), b1c.numpy(), rtol=TOL[dtype][0] * 5, atol=atol_b)
        np.testing.assert_allclose(b2g.cpu().numpy() - 1.0, b1g.cpu().numpy(), rtol=1e-5, atol=1e-5 * (1 + float(b1g.abs().max())))   # accumulate = same sum + 1
        tw.check(TOL[dtype][0] * 5, TOL[dtype][1] * 5)
        oc, og = tw.view(B, H, W, C, 0, rand=False)
        tw.run(tw.c.ops.fold_add(g2c, gc, bool(fold), oc), tw.g.ops.fold_add(g2g, gg, bool(fold), og))
        tw.run(tw.c.ops.act_bwd(yc, 3, gc, bool(fold), g2c, dc), tw.g.ops.act_bwd(yg, 3, gg, bool(fold), g2g, dg))
        tw.check(TOL[dtype][0] * 5, TOL[dtype][1] * 5)


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_layout_and_losses_twins(dtype):
    B, S = 3, 24
    tw = Twin(dtype)
    img_c, img_g = tw.f32(torch.rand(B, 3, S, S, generator=tw.gen) * 2 - 1)
    for halo, mode in ((3, 2), (1, 1)):
        vc, vg = tw.view(B, S, S, 8, halo, rand=False)
        tw.run(tw.c.ops.nchw_to_view(img_c, 3, vc, mode), tw.g.ops.nchw_to_view(img_g, 3, vg, mode))
        wc, wg = tw.view(B, S, S, 8, halo, rand=False)
        tw.run(tw.c.ops.view_copy(vc, wc, mode), tw.g.ops.view_copy(vg, wg, mode))
    oc, og = tw.f32(torch.zeros(B, 3, S, S))
    tw.run(tw.c.ops.view_to_nchw(vc, 3, oc), tw.g.ops.view_to_nchw(vg, 3, og))
    # DiffAugment
    from gan_variant_research_amd.cut import DiffAugment
    aug = DiffAugment(["color", "translation", "cutout"])
    prm = DiffAugment.to_params(aug.sample(B, S, S, tw.gen), B, S, S).reshape(-1)
    pc, pg = tw.f32(prm)
    xc, xg = tw.view(B, S, S, 8, 0)
    xc.nhwc()[..., 3:] = 0; xg.t.copy_(xc.t)
    yc, yg = tw.view(B, S, S, 8, 1, rand=False)
    wsc, wsg = torch.zeros(64), torch.zeros(64, device=DEV)
    tw.run(tw.c.ops.diffaug_fwd(xc, 3, pc, yc, wsc), tw.g.ops.diffaug_fwd(xg, 3, pg, yg, wsg))
    gyc, gyg = tw.view(B, S, S, 8, 0)
    gxc, gxg = tw.view(B, S, S, 8, 0, rand=False)
    tw.run(tw.c.ops.diffaug_bwd(gyc, 3, pc, gxc, wsc), tw.g.ops.diffaug_bwd(gyg, 3, pg, gxg, wsg))
    # patch losses on a logits view, all modes
    lc, lg = tw.view(B, 6, 6, 8, 0)
    for mode, tgt in ((0, 0.0), (1, 0.0), (2, 0.0), (3, 1.0), (4, 1.0), (4, 0.0)):
        sc, sg = tw.f32(torch.zeros(1))
        gc, gg = tw.view(B, 6, 6, 8, 2, rand=False)
        tw.run(tw.c.ops.patch_loss(lc, mode, tgt, 0.5, sc, gc), tw.g.ops.patch_loss(lg, mode, tgt, 0.5, sg, gg))
    # L1 and R1
    sc, sg = tw.f32(torch.zeros(1)); dsc, dsg = tw.f32(torch.tensor([0.1]))
    gc, gg = tw.view(B, S, S, 8, 0, rand=False)
    w1c, w1g = torch.zeros(1024), torch.zeros(1024, device=DEV)
    tw.run(tw.c.ops.l1_loss(xc, 3, img_c, 1.0, dsc, sc, gc, w1c), tw.g.ops.l1_loss(xg, 3, img_g, 1.0, dsg, sg, gg, w1g))
    sc, sg = tw.f32(torch.zeros(1))
    uc, ug = tw.view(B, S, S, 8, 1, rand=False)
    tw.run(tw.c.ops.r1_reduce(xc, 3, 160.0, sc, uc, w1c), tw.g.ops.r1_reduce(xg, 3, 160.0, sg, ug, w1g))
    tw.check(*([1e-4, 1e-5] if dtype == F32 else [2e-2, 2e-2]))


@pytest.mark.parametrize("shape", [(512, 256, 4), (64, 3, 4), (1, 512, 4), (37, 5, 3)])
def test_spectral_norm_twins(shape):
    """gan_spectral_norm_fwd (training: one power iteration in place; eval: sigma from the stored u, v) and _bwd against the
    torch formulas of torch.nn.utils.spectral_norm (tests/emulator.py), at the largest discriminator weight (512 x 4096)."""
    cout, cin, k = shape
    tw = Twin(F32)
    Wc, Wg = tw.f32(torch.randn(cout, cin, k, k, generator=tw.gen) * 0.05)
    h, w = cout, cin * k * k
    uc, ug = tw.f32(torch.nn.functional.normalize(torch.randn(h, generator=tw.gen), dim=0))
    vc, vg = tw.f32(torch.nn.functional.normalize(torch.randn(w, generator=tw.gen), dim=0))
    sc, sg = tw.f32(torch.zeros(1))
    nc, ng = tw.f32(torch.zeros(cout, cin, k, k))
    n = tw.c.ops.spectral_norm_ws_floats(h, w)
    wsc, wsg = torch.zeros(n), torch.zeros(tw.g.ops.spectral_norm_ws_floats(h, w), device=DEV)
    for train in (True, True, False):
        tw.run(tw.c.ops.spectral_norm_fwd(Wc, uc, vc, train, 1e-12, sc, nc, wsc), tw.g.ops.spectral_norm_fwd(Wg, ug, vg, train, 1e-12, sg, ng, wsg))
    Gc, Gg = tw.f32(torch.randn(cout, cin, k, k, generator=tw.gen))
    dc, dg = tw.f32(torch.zeros(cout, cin, k, k))
    tw.run(tw.c.ops.spectral_norm_bwd(Gc, nc, uc, vc, sc, dc, wsc), tw.g.ops.spectral_norm_bwd(Gg, ng, ug, vg, sg, dg, wsg))
    tw.check(2e-4, 2e-5)


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("hw", [(24, 24), (25, 31), (7, 2), (1, 1)])
def test_avgpool_twins(hw, dtype):
    """AvgPool2d(3, 2, 1, count_include_pad=False) forward / transpose (multiscale discriminator), even, odd and degenerate sizes."""
    B, (H, W) = 2, hw
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    tw = Twin(dtype)
    xc, xg = tw.view(B, H, W, 8, 1)
    yc, yg = tw.view(B, Ho, Wo, 8, 1, rand=False)
    tw.run(tw.c.ops.avgpool_fwd(xc, yc), tw.g.ops.avgpool_fwd(xg, yg))
    gyc, gyg = tw.view(B, Ho, Wo, 8, 0)
    for acc in (False, True):
        gxc, gxg = tw.view(B, H, W, 8, 0, rand=acc)
        tw.run(tw.c.ops.avgpool_bwd(gyc, gxc, acc), tw.g.ops.avgpool_bwd(gyg, gxg, acc))
    tw.check(*([1e-5, 1e-6] if dtype == F32 else [1e-2, 1e-2]))


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("shape", [(2, 16, 16, 64, 1, 256), (3, 8, 8, 256, 1, 64), (2, 12, 12, 128, 0, 144),   # MFMA tiling
                                   (2, 12, 12, 32, 0, 100), (2, 8, 8, 512, 1, 48)])                            # scalar-FMA fallback
def test_patchnce_twins(shape, dtype):
    B, H, W, C, halo, P = shape
    tw = Twin(dtype, seed=5)
    sc, sg = tw.view(B, H, W, C, halo)
    tc, tg = tw.view(B, H, W, C, halo)
    tc.t.copy_((sc.t.float() + 0.5 * tc.t.float()).to(tc.t.dtype)); tg.t.copy_(tc.t)
    ids = torch.randint(0, H * W, (P,), generator=tw.gen)
    ids[1] = ids[0]; ids[-1] = ids[0]          # duplicates must accumulate
    ic, ig = tw.i32(ids)
    lc, lg = tw.f32(torch.zeros(1))
    gc, gg = tw.view(B, H, W, C, halo, scale=0.01)
    n = tw.c.ops.patchnce_ws_floats(B, P, C)
    wc, wg = torch.zeros(n), torch.zeros(n, device=DEV)
    tw.run([tw.c.ops.patchnce_fwd(sc, tc, ic, P, C, 0.07, 0.25, lc, wc), tw.c.ops.patchnce_bwd(tc, ic, P, C, 0.07, 0.25, gc, wc)],
           [tw.g.ops.patchnce_fwd(sg, tg, ig, P, C, 0.07, 0.25, lg, wg), tw.g.ops.patchnce_bwd(tg, ig, P, C, 0.07, 0.25, gg, wg)])
    tw.check(*([2e-4, 2e-6] if dtype == F32 else [2e-2, 2e-3]))


def test_adam_twins(golden):
    """Fused clip+Adam+EMA against the golden vectors of the reference's AMPContext.step_optimizer + EMA.update."""
    from gan_variant_research_amd.cut import FusedAdam
    g = golden("cut_optim.npz")
    ctx = hip_ctx(F32)
    names = ["a", "b", "z"]
    init = {k: torch.from_numpy(g[f"p0.{k}"]).to(DEV) for k in names}
    opt = FusedAdam(ctx, names, [init[k].shape for k in names], init, ema_decay=0.999)
    step = opt.step_op(10.0)
    for s in range(3):
        for k in names:
            opt.grads[k].copy_(torch.from_numpy(g[f"g{s}.{k}"]))
        step()
        torch.cuda.synchronize()
        for k in names:
            np.testing.assert_allclose(opt.params[k].cpu().numpy(), g[f"p{s+1}.{k}"], rtol=2e-6, atol=2e-7)
            np.testing.assert_allclose(opt.shadow[k].cpu().numpy(), g[f"ema{s+1}.{k}"], rtol=2e-6, atol=2e-7)
    assert opt.steps.tolist() == [3, 3, 3]
    skip = opt.step_op(10.0, skip=["z"])
    skip(); torch.cuda.synchronize()
    assert opt.steps.tolist() == [4, 4, 3]


# ---------------------------------------------------------------------------------------------- whole step
@pytest.mark.parametrize("use_aug", [True, False])
def test_cut_train_step_fp32_vs_oracle(use_aug):
    """fp32 (parity) mode: losses within 1e-3 relative of the PyTorch-CPU oracle at steps 0 and 1 (north_star tolerance)."""
    tr, img, ref_img = cases.run_cut_steps(DEV, HipOps(torch.device(DEV)), use_aug, amp=False, S=64, B=2, tol0=1e-3, tol1=2e-3, atol1=5e-4)
    np.testing.assert_allclose(img.numpy(), ref_img.numpy(), rtol=1e-3, atol=1e-3)


def test_cut_train_step_fp32_vs_oracle_at_benchmark_resolution():
    """The same comparison at 256x256 (the benchmark's image size; batch 2, one step, DiffAugment on): the kernels the bench runs -- 64x64
    residual maps, 66x66 input-gradient domains, the 7x7 window kernels on 256x256 maps -- against the PyTorch-CPU oracle within 1e-3."""
    torch.set_num_threads(16)
    tr, img, ref_img = cases.run_cut_steps(DEV, HipOps(torch.device(DEV)), True, amp=False, S=256, B=2, nsteps=1, tol0=1e-3, ptol=4.5e-4, threads=16)
    np.testing.assert_allclose(img.numpy(), ref_img.numpy(), rtol=1e-3, atol=1e-3)


def test_cut_train_step_bf16_vs_oracle_at_benchmark_resolution(monkeypatch):
    """The benched mode (bf16 operands, fp32 accumulation) at 256x256 with the planner's large-batch choices forced (256-channel
    range-patch tiles), against the oracle with the bf16 tolerance (8-bit mantissa: 4e-2 on the losses, 5e-2 absolute on the image)."""
    monkeypatch.setenv("GAN_PATCH_BN", "256")
    tr, img, ref_img = cases.run_cut_steps(DEV, HipOps(torch.device(DEV)), True, amp=True, S=256, B=2, nsteps=1, tol0=4e-2, ptol=4.5e-4, threads=16)
    assert float((img - ref_img).abs().max()) < 5e-2


@pytest.mark.parametrize("B,S", [(1, 64), (3, 64), (5, 64), (2, 80), (3, 96)])
def test_cut_train_step_ragged_shapes_vs_oracle(B, S):
    """Batch sizes that are not powers of two (and a single image) and image sizes whose maps are no multiple of any tile (80 -> 20x20 and
    40x40 maps, 96 -> 24x24): GEMM row counts with tails in every tile size (128, 256, 288), persistent blocks with uneven tile counts,
    weight-gradient stages that straddle image rows and images (the general addressing path), split-K over 2B = 2 ... 10 images -- fp32
    mode against the oracle within 1e-3."""
    tr, img, ref_img = cases.run_cut_steps(DEV, HipOps(torch.device(DEV)), True, amp=False, S=S, B=B, nsteps=1, tol0=1e-3)
    np.testing.assert_allclose(img.numpy(), ref_img.numpy(), rtol=1e-3, atol=1e-3)


def test_cut_train_step_bf16_vs_oracle():
    """bf16 throughput mode (fp32 accumulation): same step, tolerance widened to bf16's 8-bit mantissa."""
    tr, img, ref_img = cases.run_cut_steps(DEV, HipOps(torch.device(DEV)), True, amp=True, S=64, B=2, nsteps=1, tol0=4e-2, ptol=4.5e-4)
    np.testing.assert_allclose(img.numpy(), ref_img.numpy(), rtol=5e-2, atol=5e-2)


def test_generator_module_forward_and_features(golden):
    """ResNetGenerator.forward / get_feature_layers on HIP vs the golden outputs of the reference modules."""
    from gan_variant_research_amd import cut as C
    g = golden("cut_models.npz")
    C.set_seed(42)
    gen, disc = C.build_models(cases.small_config(), DEV)
    x = torch.from_numpy(g["x64"]).to(DEV)
    assert gen(x).requires_grad                     # the modules are differentiable nn.Modules now (autograd.py)
    with torch.no_grad():
        np.testing.assert_allclose(gen(x).cpu().numpy(), g["G64"], rtol=1e-3, atol=1e-3)
        feats = gen.get_feature_layers(x, [0, 4, 8, 12, 16])
    assert len(feats) == 4
    for i, f in enumerate(feats):
        assert list(f.shape) == list(g[f"feat{i}.shape"])
        np.testing.assert_allclose(f[:, :8, :4, :4].cpu().numpy(), g[f"feat{i}.slice"], rtol=1e-3, atol=1e-3)


def test_basic_gan_iterations_fp32_vs_oracle():
    """Basic_GAN CycleGAN inner loop (Basic_GAN/src/train.py:66-122), fp32 parity mode, 64x64, two iterations."""
    cases.run_basic_iterations(DEV, HipOps(torch.device(DEV)), amp=False, S=64, B=2, tol0=1e-3, tol1=2e-3)


def test_basic_gan_iterations_bf16_vs_oracle():
    cases.run_basic_iterations(DEV, HipOps(torch.device(DEV)), amp=True, S=64, B=2, niter=1, tol0=4e-2)


# ---------------------------------------------------------------------------------------------- drop-in nn.Module / loss / optimiser API
def test_autograd_bridge_hip():
    """G(x), get_feature_layers, D(x), r1_regularization as differentiable nn.Module calls on the HIP kernels vs the oracle under autograd."""
    from tests.test_autograd_bridge import bridge_cases
    bridge_cases(DEV, 5e-4)


@pytest.mark.parametrize("tag", ["ms3", "sn2", "bsn"])
def test_optional_discriminators_hip(monkeypatch, tag):
    """MultiscaleDiscriminator with num_scales=3 (ms3) and with spectral norm over two scales (sn2) -- outputs, gradients, the power
    iteration's buffers, R1 through the pooled scales, eval mode -- against vectors produced by the reference itself."""
    from gan_variant_research_amd import losses as L
    from tests.test_autograd_bridge import optional_cases
    monkeypatch.setattr(L, "_PLANS", {})
    optional_cases(torch.device(DEV), 5e-4, tags=(tag,))


@pytest.mark.parametrize("tag", ["gz", "grl"])
def test_generator_variants_hip(tag):
    """padding_type='zero' / activation='leaky_relu' generators on the HIP kernels vs vectors produced by the reference."""
    from tests.test_autograd_bridge import generator_variant_cases
    generator_variant_cases(torch.device(DEV), 5e-4, tags=(tag,))


@pytest.mark.parametrize("tag", ["grep", "gbn", "gnn", "gd1", "gd3"])
def test_generator_switches_hip(tag):
    """The remaining constructor switches (replicate padding, norm 'batch' / 'none', no block activation, 1 / 3 down-samplings): the
    layer-wise generator on the HIP kernels vs vectors produced by the reference (tests/golden/cut_variants.npz)."""
    from tests.test_autograd_bridge import generator_switch_cases
    generator_switch_cases(torch.device(DEV), 5e-4, tags=(tag,))


def test_pad_ops_hip():
    """mi355x_gan::replication_pad2d / reflection_pad2d and their gradients (gan_nchw_to_view halo modes, gan_pad_fold, gan_fold_add)
    against torch.nn.functional.pad, ragged sizes and pads 1..3."""
    from tests.test_ops_library import pad_op_cases
    pad_op_cases(DEV)


@pytest.mark.parametrize("H", [32, 37])
def test_first_conv_window_kernel_fused_statistics(H):
    """The 3 -> 64 channel 7x7 window kernel also writes per-tile (sum, sum of squares): with gan_in_stats_from_parts they are the
    InstanceNorm statistics of its (unrounded) result, ragged tiles included."""
    from gan_variant_research_amd.convplan import ConvLayer
    B, Cc = 3, 64
    ctx = hip_ctx(BF16)
    g = torch.Generator().manual_seed(4)
    x = (torch.rand(B, 3, H, H, generator=g) * 2 - 1).bfloat16().float()
    w = torch.randn(Cc, 3, 7, 7, generator=g) * 0.08
    b = torch.randn(Cc, generator=g) * 0.5
    layer = ConvLayer(ctx, w.to(DEV), b.to(DEV), torch.zeros_like(w).to(DEV), torch.zeros_like(b).to(DEV), 7, 1, 3)
    xin = cases.to_view(ctx, x, 3, 2)
    y = ctx.view(B, H, H, Cc, 0)
    ws = ctx.f32(B * 1024 * Cc * 2)
    ops = layer.fwd(xin, y, stats_ws=ws)
    assert layer.stats_parts == (-(-H // 16)) ** 2, layer.stats_parts
    st = ctx.f32(B * Cc * 2)
    for o in layer.repack_ops() + ops + [ctx.ops.in_stats_from_parts(ws, layer.stats_parts, B, Cc, H * H, 1e-5, st)]:
        o()
    torch.cuda.synchronize()
    ref = torch.nn.functional.conv2d(torch.nn.functional.pad(x, (3, 3, 3, 3), mode="reflect"), w.bfloat16().float(), b)
    sg = st.cpu().view(B, Cc, 2)
    np.testing.assert_allclose(sg[..., 0].numpy(), ref.mean((2, 3)).numpy(), rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(sg[..., 1].numpy(), (1.0 / torch.sqrt(ref.var((2, 3), unbiased=False) + 1e-5)).numpy(), rtol=2e-3)


def test_rccl_path_with_one_rank_changes_nothing():
    """bench.py with a one-rank RCCL group (GAN_FORCE_DIST=1: stream binding before the group exists, discriminator all-reduce on its
    stream, the generator's two gradient buckets with the tail overlapping the backward) must produce the same losses, bit for bit, as
    the plain run and as the single-all-reduce variant: a collective issued before its gradients are final would show up here only with
    more ranks, but a mis-sliced bucket, a missing wait or a wrong stream shows up as a different loss or a hang."""
    import json
    import subprocess
    import sys
    import socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def free_port():
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            return str(sk.getsockname()[1])
    outs = []
    for extra in ({}, {"GAN_FORCE_DIST": "1"}, {"GAN_FORCE_DIST": "1", "GAN_NO_BUCKET_AR": "1"}):
        env = dict(os.environ, **extra)
        if extra:
            env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=free_port(), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        r = subprocess.run([sys.executable, "bench.py", "--steps", "4", "--warmup", "2", "--batch", "4", "--size", "128", "--no-cpu-baseline"], cwd=root, env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, lines                      # ONE JSON line on stdout, RCCL's banner included nowhere
        outs.append(json.loads(lines[0])["last_losses"])
    assert outs[0] == outs[1] == outs[2], outs


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (one RCCL rank per GPU)")
def test_two_gpu_ranks_equal_one_rank(tmp_path):
    """Two fresh ranks, one per GPU, over RCCL with the overlapped two-bucket generator all-reduce on: the summed gradient blocks and
    the mean losses of CUT step 0 must equal the one-rank run on the global batch (the assertion of tests/test_dp_gloo.py, on HIP)."""
    import socket
    import subprocess
    import sys
    from gan_variant_research_amd import cut as C
    from tests import test_dp_gloo as T
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    out = str(tmp_path / "rank")
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "dp_hip_worker.py"), out], cwd=root,
                              env=dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                                       HSA_ENABLE_IPC_MODE_LEGACY="0")) for r in range(2)]
    for pr in procs:
        assert pr.wait(timeout=900) == 0
    res = [torch.load(f"{out}.{r}", weights_only=True) for r in range(2)]
    cfg = cases.small_config()
    C.set_seed(42)
    gen, disc = C.build_models(cfg, "cpu")
    single = C.CutTrainer(gen, disc, cfg, T.BG, T.S, device=DEV, amp=False, ops=HipOps(torch.device(DEV)))
    photos, monets = T._inputs()
    ref_losses = single.train_step(0, photos.to(DEV), monets.to(DEV), T._global_randomness(T._make(T.BG)))
    torch.cuda.synchronize()
    assert torch.equal(res[0]["flat_g"], res[1]["flat_g"]) and torch.equal(res[0]["flat_gd"], res[1]["flat_gd"])   # both ranks hold the same sum
    assert T._grad_err(res[0]["flat_g"], single.opt_G.flat_g.cpu()) < 1e-3
    assert T._grad_err(res[0]["flat_gd"], single.opt_D.flat_g.cpu()) < 1e-3
    for k in ("d_loss", "g_adv", "nce", "identity", "r1"):
        np.testing.assert_allclose(0.5 * (res[0]["losses"][k] + res[1]["losses"][k]), ref_losses[k], rtol=1e-3, atol=1e-4, err_msg=k)


@pytest.mark.parametrize("H", [16, 24])
def test_fp8_conv_path_hip(H):
    """BASELINE.json configs[4]: the e4m3 operand path of the bottleneck 3x3 256->256 convolution (forward and input gradient) on
    v_mfma_scale_f32_16x16x128_f8f6f4 -- against F.conv2d on the same e4m3-rounded operands (exact up to summation order / bf16 output
    rounding), against the unrounded convolution within the stated fp8 tolerance (tests/cases.py:run_conv_fp8), and against the CPU
    statement of the same launches."""
    got, gotd = cases.run_conv_fp8(hip_ctx(BF16), H=H)
    emu, emud = cases.run_conv_fp8(Ctx(EmuOps(), "cpu", BF16), H=H)
    rms, rmsd = float(emu.pow(2).mean().sqrt()), emud.pow(2).mean((1, 2, 3)).sqrt()
    assert float((got - emu).abs().max()) < 2e-2 * rms
    for bi in range(got.shape[0]):
        assert float((gotd[bi] - emud[bi]).abs().max()) < 2e-2 * float(rmsd[bi])


def test_fp8_conv_path_wide_maps_hip():
    """The kernel `bench.py --fp8 --size 512` runs: at 512x512 the residual maps are 128 pixels wide, a 256-pixel tile spans 522 padded
    pixels and the launch goes to the 9-slice e4m3 instantiation (conv_patch_fp8_wide_kernel).  Forward (with the fused InstanceNorm
    partials) and padded-domain input gradient with a per-image scale, same assertions as the narrow case; the planner's choice is
    asserted through gan_conv_patch_variant."""
    ctx, info = hip_ctx(BF16), {}
    got, gotd = cases.run_conv_fp8(ctx, B=1, H=128, with_stats=True, info=info)
    for which in ("fwd", "dgrad"):
        v = ctx.ops.conv_patch_variant(info[which])
        assert v == {"rows": 256, "cols": 128, "slices": 9, "fp8": True, "static9": False, "static_taps": 0}, (which, v)
    emu, emud = cases.run_conv_fp8(Ctx(EmuOps(), "cpu", BF16), B=1, H=128)
    rms, rmsd = float(emu.pow(2).mean().sqrt()), emud.pow(2).mean((1, 2, 3)).sqrt()
    # 4.2 M outputs: some sit on a bf16 rounding boundary, where the two fp32 summation orders land one output ulp (<= 2^-7 |v|) apart
    assert bool(((got - emu).abs() <= 2e-2 * rms + 2.0 ** -7 * emu.abs()).all()), float((got - emu).abs().max())
    assert bool(((gotd[0] - emud[0]).abs() <= 2e-2 * float(rmsd[0]) + 2.0 ** -7 * emud[0].abs()).all()), float((gotd[0] - emud[0]).abs().max())


def test_cut_train_step_fp8_at_512_vs_oracle():
    """BASELINE.json configs[4] at its own resolution: one fp8 CUT step at 512x512 (batch 1), where every residual convolution and input
    gradient runs on the 9-slice e4m3 kernel (asserted), against the fp32 oracle with the tolerances of the 64x64 fp8 step test."""
    ops = HipOps(torch.device(DEV))
    seen = []
    orig = ops.conv_igemm
    ops.conv_igemm = lambda c: (seen.append(ops.conv_patch_variant(c)) if c.x.dtype == 2 else None, orig(c))[1]
    tr, img, ref_img = cases.run_cut_steps(DEV, ops, True, amp=True, S=512, B=1, nsteps=1, tol0=8e-2, ptol=4.5e-4, fp8=True, threads=16)
    assert tr.fp8 and len(seen) >= 36 and all(v.get("fp8") and v.get("slices") == 9 for v in seen), seen[:3]
    assert float((img - ref_img).abs().max()) < 0.3 and float((img - ref_img).pow(2).mean().sqrt()) < 0.06


def test_cut_train_step_fp8_vs_oracle():
    """BASELINE.json configs[4] at test size: the CUT step with e4m3 operand copies for the residual convolutions (forward and input
    gradient) at 64x64 against the fp32 oracle.  Stated fp8 tolerance: step-0 losses within 8 %, generated image within 0.3 max / 0.06 rms."""
    tr, img, ref_img = cases.run_cut_steps(DEV, HipOps(torch.device(DEV)), True, amp=True, S=64, B=2, nsteps=1, tol0=8e-2, ptol=4.5e-4, fp8=True)
    assert tr.fp8
    assert float((img - ref_img).abs().max()) < 0.3 and float((img - ref_img).pow(2).mean().sqrt()) < 0.06


def test_ops_library_hip(monkeypatch):
    """torch.ops.mi355x_gan.* on the kernels: the generator and discriminator assembled layer by layer from the op-level drop-in modules
    reproduce the reference's own outputs (golden) and the oracle's gradients; the fused clip+Adam+EMA op matches torch.optim.Adam."""
    from gan_variant_research_amd import ops_library as L, training as T
    from tests.test_ops_library import fused_adam_case, layerwise_cases, new_op_cases
    from gan_variant_research_amd import losses as LS
    monkeypatch.setattr(L, "_PLANS", {})
    monkeypatch.setattr(LS, "_PLANS", {})
    monkeypatch.setattr(T, "_FUSED_PLANS", {})
    layerwise_cases(torch.device(DEV), 1e-3)
    fused_adam_case(torch.device(DEV))
    new_op_cases(torch.device(DEV), 1e-4)       # patchnce_fwd/_bwd, diffaugment_fwd/_bwd vs the reference's golden vectors, allreduce_bucket_


def test_inference_chain_vs_oracle(tmp_path):
    """SURVEY §8f-2 on the GPU (generate_folder.py:125-205, 183-185): a reference-layout checkpoint whose `ema_G.shadow` differs from
    `generator` -> inference.load_generator (EMA preferred) -> stylize on HIP must equal the oracle's generator_forward of the EMA weights
    -> clamp -> *0.5+0.5 -> *255 -> round within +-1 LSB (fp32 operands); bf16 operands within +-6 LSB = the 5e-2 image tolerance of the
    bf16 step test on the 127.5-per-unit scale."""
    from gan_variant_research_amd import cut as C, inference as I
    from oracle import cut_ref
    C.set_seed(5)
    G = C.ResNetGenerator(3, 3, 64, 9)
    shadow = {k: (v.detach() * 0.9).clone() for k, v in G.state_dict().items()}
    ck = tmp_path / "ckpt_final.pt"
    torch.save({"step": 3, "generator": G.state_dict(), "discriminator": {}, "opt_G": {}, "opt_D": {}, "metrics": {}, "config": {},
                "ema_G": {"decay": 0.999, "shadow": shadow}, "scaler": {}}, ck)
    x = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(8)) * 2 - 1
    want = cut_ref.generator_forward(shadow, x).detach()
    want = (want.clamp(-1, 1) * 0.5 + 0.5).mul(255).round()
    for bf16, lsb in ((False, 1), (True, 6)):
        G2 = I.load_generator(str(ck), device=DEV, bf16=bf16)
        for k, v in G2.state_dict().items():
            assert torch.equal(v.cpu(), shadow[k]), k
        u8 = I.stylize(G2, x.to(DEV))
        assert u8.dtype == torch.uint8 and u8.is_cuda and tuple(u8.shape) == (2, 3, 64, 64)
        assert int((u8.cpu().float() - want).abs().max()) <= lsb, (bf16, int((u8.cpu().float() - want).abs().max()))


def test_inference_graph_replay_equals_eager():
    """Forward-only passes of a module with use_graph=True are captured once per input shape and replayed as one hipGraph launch:
    same bits as the eager launches, also after the weights changed (the graph reads the refreshed operand copies)."""
    from gan_variant_research_amd import cut as C, inference as I
    C.set_seed(1)
    G = C.ResNetGenerator(3, 3, 16, 3).to(DEV).eval()
    G.compute_dtype = BF16
    x = (torch.rand(2, 3, 64, 64) * 2 - 1).to(DEV)
    eager = I.stylize(G, x)
    G.use_graph = True
    first, replay = I.stylize(G, x), I.stylize(G, x)
    assert torch.equal(eager, first) and torch.equal(eager, replay)
    with torch.no_grad():
        for p_ in G.parameters():
            p_.mul_(1.01)
    after = I.stylize(G, x)
    G.use_graph = False
    assert torch.equal(after, I.stylize(G, x)) and not torch.equal(after, eager)


def test_loss_callables_hip(monkeypatch):
    from gan_variant_research_amd import losses as L
    from tests.test_autograd_bridge import loss_cases
    monkeypatch.setattr(L, "_PLANS", {})
    loss_cases(DEV, 1e-4)


def test_training_utilities_hip(tmp_path):
    from tests.test_autograd_bridge import training_cases
    training_cases(DEV, 1e-5, tmp_path)


def test_module_step_default_discriminator_family_hip(monkeypatch):
    """module_step.train_step with a two-scale spectral-norm discriminator vs the reference's own train_step (golden), steps 0-1."""
    from gan_variant_research_amd import losses as L
    from tests.test_autograd_bridge import optional_step_cases
    monkeypatch.setattr(L, "_PLANS", {})
    optional_step_cases(torch.device(DEV), 1e-3, 2e-3)


def test_module_step_hip(monkeypatch):
    """The reference's train_step, written against the drop-in module API, on the HIP kernels vs the oracle (steps 0-1, DiffAugment on)."""
    from gan_variant_research_amd import losses as L
    from tests.test_autograd_bridge import module_step_cases
    monkeypatch.setattr(L, "_PLANS", {})
    module_step_cases(DEV, 1e-3, 2e-3)


@pytest.mark.parametrize("bn", ["128", "256"])
@pytest.mark.parametrize("bm", ["256", "288"])
@pytest.mark.parametrize("H", [16, 20])
def test_conv_fused_instance_norm_statistics(bm, H, bn, monkeypatch):
    """Per-tile (sum, sum of squares) written by the range-patch epilogue + gan_in_stats_from_parts == InstanceNorm statistics of the
    convolution result; both tile heights and widths, a map with a partial last tile (H=20: 400 pixels)."""
    from gan_variant_research_amd.convplan import ConvLayer
    monkeypatch.setenv("GAN_PATCH_BM", bm)
    monkeypatch.setenv("GAN_PATCH_BN", bn)
    B, Cc = 3, 256
    tw = Twin(BF16, seed=11)
    xc, xg = tw.view(B, H, H, Cc, 1)
    yc, yg = tw.view(B, H, H, Cc, 0, rand=False)
    w = torch.randn(Cc, Cc, 3, 3, generator=tw.gen) * 0.03
    b = torch.randn(Cc, generator=tw.gen) * 0.5
    stats, parts = [], []
    for ctx, x, y in ((tw.c, xc, yc), (tw.g, xg, yg)):
        dev = ctx.device
        layer = ConvLayer(ctx, w.to(dev), b.to(dev), torch.zeros_like(w).to(dev), torch.zeros_like(b).to(dev), 3, 1, 1)
        ws = ctx.f32(B * 96 * Cc * 2)
        ops = layer.fwd(x, y, stats_ws=ws)
        assert layer.stats_parts > 0
        parts.append(layer.stats_parts)
        st = ctx.f32(B * Cc * 2)
        for o in layer.repack_ops() + ops + [ctx.ops.in_stats_from_parts(ws, layer.stats_parts, B, Cc, H * H, 1e-5, st)]:
            o()
        stats.append(st)
    torch.cuda.synchronize()
    assert parts[1] == -(-H * H // int(bm))
    tw.check(2e-2, 2e-2)                       # the convolution result itself
    sc, sg = stats[0].view(B, Cc, 2), stats[1].cpu().view(B, Cc, 2)
    np.testing.assert_allclose(sg[..., 0].numpy(), sc[..., 0].numpy(), rtol=2e-3, atol=2e-3)   # mean
    np.testing.assert_allclose(sg[..., 1].numpy(), sc[..., 1].numpy(), rtol=2e-3)              # rstd
    # and they are the statistics of the (unrounded) convolution result
    ref = torch.nn.functional.conv2d(xc.padded().float().permute(0, 3, 1, 2), w.to(torch.bfloat16).float(), b)   # the halo is part of the view
    np.testing.assert_allclose(sg[..., 0].numpy(), ref.mean((2, 3)).numpy(), rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(sg[..., 1].numpy(), (1.0 / torch.sqrt(ref.var((2, 3), unbiased=False) + 1e-5)).numpy(), rtol=2e-3)


def test_multi_stream_step_is_deterministic(monkeypatch):
    """Three HIP streams per step (main, weight gradients, discriminator) and no atomics anywhere: two runs from the same seeds must
    give bit-identical losses and weights -- a race between streams shows up here as run-to-run noise."""
    from gan_variant_research_amd import cut as C
    import bench
    cfg = bench.default_config()
    cfg["model"]["generator"]["ngf"], cfg["model"]["discriminator"]["ndf"] = 32, 32
    cfg["r1"]["every"] = 2                               # R1 inside the window
    B, S = 4, 128

    def run():
        C.set_seed(42)
        gen, disc = C.build_models(cfg, DEV)
        tr = C.CutTrainer(gen, disc, cfg, B, S, device=DEV, amp=True)
        g = torch.Generator().manual_seed(3)
        photos = (torch.rand(B, 3, S, S, generator=g) * 2 - 1).to(DEV)
        monets = (torch.rand(B, 3, S, S, generator=g) * 2 - 1).to(DEV)
        rg = torch.Generator().manual_seed(9)
        out = []
        for step in range(4):
            out.append(tr.train_step(step, photos, monets, tr.sample_randomness(rg), sync="lag" if step % 2 else True))
        out.append(tr.flush_losses())
        torch.cuda.synchronize()
        return [o for o in out if o is not None], tr.opt_G.flat_p.clone(), tr.opt_D.flat_p.clone()
    la, ga, da = run()
    lb, gb, db = run()
    assert la == lb, (la, lb)
    assert torch.equal(ga, gb) and torch.equal(da, db)
    monkeypatch.setenv("GAN_SINGLE_STREAM", "1")          # the same launches on ONE stream: concurrency must not change a bit
    lc, gc, dc = run()
    assert la == lc, (la, lc)
    assert torch.equal(ga, gc) and torch.equal(da, dc)


@pytest.mark.parametrize("bn", ["128", "256"])
@pytest.mark.parametrize("bm", ["256", "288"])
@pytest.mark.parametrize("H", [16, 20])
def test_conv_backward_chain_epilogue(bm, bn, H, monkeypatch):
    """The backward-chain epilogue of the range-patch kernel (gan_conv_desc.stats_mode 1) against its CPU statement: the padded-domain
    input gradient, and the per-image sums the InstanceNorm backward behind the ReLU needs (sum g [y > 0], sum g y) -- summed over the
    kernel's tiles they must equal the emulator's single partial; both tile heights and widths, a map with a partial last tile
    (22x22 = 484 pixels).  Then gan_in_bwd_parts on those partials against the two-pass gan_in_bwd on the same gradient."""
    from gan_variant_research_amd.convplan import ConvLayer
    from gan_variant_research_amd._lib import ACT_RELU
    monkeypatch.setenv("GAN_PATCH_BM", bm)
    monkeypatch.setenv("GAN_PATCH_BN", bn)
    B, Cc = 3, 256
    tw = Twin(BF16, seed=5)
    dyc, dyg = tw.view(B, H, H, Cc, 2, zero_halo=True)
    xc, xg = tw.view(B, H, H, Cc, 0)                                       # raw norm input
    stc, stg = tw.f32(torch.stack([xc.nhwc().float().mean((1, 2)), 1.0 / torch.sqrt(xc.nhwc().float().var((1, 2), unbiased=False) + 1e-5)], -1).reshape(-1))
    yc, yg = tw.view(B, H, H, Cc, 1, rand=False)                           # relu(xhat) with its reflect halo, as the forward saved it
    outc, outg = tw.view(B, H, H, Cc, 1, rand=False)
    d1c, d1g = tw.view(B, H, H, Cc, 2, rand=False)
    d2c, d2g = tw.view(B, H, H, Cc, 2, rand=False)
    w = torch.randn(Cc, Cc, 3, 3, generator=tw.gen) * 0.03
    sums, parts = [], []
    for ctx, dy, x, st, y, out, d1, d2 in ((tw.c, dyc, xc, stc, yc, outc, d1c, d2c), (tw.g, dyg, xg, stg, yg, outg, d1g, d2g)):
        dev = ctx.device
        from gan_variant_research_amd._lib import HALO_REFLECT
        ctx.ops.in_apply(x, st, ACT_RELU, None, y, HALO_REFLECT)()
        layer = ConvLayer(ctx, w.to(dev), None, torch.zeros_like(w).to(dev), None, 3, 1, 1)
        ws = ctx.f32(B * 96 * Cc * 2)
        ops = layer.dgrad(dy, out, padded_domain=True, chain={"operand": y, "ws": ws})
        for o in layer.repack_ops() + ops:
            o()
        n = layer.chain_parts
        parts.append(n)
        sums.append(ws[:B * n * Cc * 2].view(B, n, Cc, 2).double().sum(1).cpu())
        if ctx is tw.g:
            v = ctx.ops.conv_patch_variant(ops[-1].conv)
            assert v["rows"] == int(bm) and v["cols"] == int(bn), v
        # the norm backward from the partials, and the two-pass one on the same gradient
        ctx.ops.in_bwd_parts(x, st, ACT_RELU, out, True, d1, ws, n, 1)()
        ctx.ops.in_bwd(x, st, ACT_RELU, out, True, None, d2, ctx.f32(B * 96 * Cc * 2 + B * Cc * 2 + (B * 1024 + 32) * Cc))()
    torch.cuda.synchronize()
    assert parts[1] == -(-(H + 2) * (H + 2) // int(bm))
    tw.check(2e-2, 2e-2)
    scale = sums[0].abs().max()
    np.testing.assert_allclose(sums[1].numpy(), sums[0].numpy(), rtol=2e-2, atol=float(scale) * 2e-3)
    np.testing.assert_allclose(d1g.t.float().cpu().numpy(), d2g.t.float().cpu().numpy(), rtol=2e-2, atol=2e-2)


def test_basic_gan_checkpoint_continuation_hip(tmp_path):
    """SURVEY §8f-1 for the CycleGAN trainer on the GPU (Basic_GAN/src/train.py:27-31,54-58,124-137): the reference's checkpoint dict, its
    LambdaLR schedule as one device float per fused optimiser, and a bit-identical continuation after load_checkpoint."""
    cases.run_basic_checkpoint_case(DEV, lambda: HipOps(torch.device(DEV)), tmp_path)


def test_adam_gradscaler_knobs_twins():
    """gan_adam_step's lr_dev / inv_scale_dev / skip_nonfinite and gan_scaler_update (torch.amp.GradScaler's unscale_, step, update:
    amp_utils.py:29-41) on the GPU against the CPU statement: clean step, overflow step (update and step counters skipped, scale backed
    off), growth after the interval, learning rate read from the device."""
    res = []
    for ops, dev in ((EmuOps(), "cpu"), (HipOps(torch.device(DEV)), DEV)):
        gen = torch.Generator().manual_seed(3)
        p = torch.randn(40000, generator=gen).to(dev)
        g = (torch.randn(40000, generator=gen) * 300).to(dev)
        m, v, ema = torch.zeros_like(p), torch.zeros_like(p), p.clone()
        step = torch.zeros(1, dtype=torch.int32, device=dev)
        table = ops.make_adam_table([{"p": p, "g": g, "m": m, "v": v, "ema": ema, "step": step}])
        ct = torch.tensor([0, 0, 0], dtype=torch.int32, device=dev)
        co = torch.tensor([0, 16384, 32768], dtype=torch.int64, device=dev)
        norm, ws = torch.zeros(4, device=dev), torch.zeros(32, device=dev)
        scale, inv = torch.tensor([256.0], device=dev), torch.tensor([1.0 / 256.0], device=dev)
        tracker, lr_dev = torch.zeros(1, dtype=torch.int32, device=dev), torch.tensor([3e-3], device=dev)
        step_op = ops.adam_step(table, 1, ct, co, 3, 1.0, 0.5, 0.999, 1e-8, 5.0, 1.0, 0.9, norm, ws, lr_dev=lr_dev, inv_scale=inv, skip_nonfinite=True)
        upd = ops.scaler_update(scale, inv, tracker, norm[2:3], 2.0, 0.5, 2)
        log = []
        for it in range(4):
            if it == 1:
                g[123] = float("nan")
            if it == 2:
                g[123] = 1.0
                lr_dev.fill_(1e-3)
            step_op(); upd()
            log.append(torch.cat([norm[:3].cpu(), scale.cpu(), step.float().cpu(), tracker.float().cpu()]))
        res.append((p.cpu(), m.cpu(), v.cpu(), ema.cpu(), torch.stack(log)))
    for a, b in zip(res[0][:4], res[1][:4]):
        np.testing.assert_allclose(b.numpy(), a.numpy(), rtol=2e-5, atol=2e-6)
    la, lb = res[0][4], res[1][4]
    assert torch.isnan(lb[1, 0]) and lb[1, 2] == 1 and lb[1, 3] == 128.0 and lb[1, 4] == 1           # overflow: skipped, scale halved
    np.testing.assert_allclose(np.nan_to_num(lb.numpy(), nan=-1.0), np.nan_to_num(la.numpy(), nan=-1.0), rtol=2e-5)
    assert lb[3, 4] == 3 and lb[3, 3] == 256.0                                                       # two clean steps: grown back


def test_caller_streams_are_ordered_with_the_bound_stream():
    """A trainer launches on the stream it was bound to at construction; its inputs may be produced on ANOTHER stream (a prefetch stream,
    `with torch.cuda.stream(s)`) and its results read there.  train_step must order the two (wait_stream both ways), and the op-level
    library must build its plan for the stream that is current at the call: results equal those of the plain default-stream run."""
    from gan_variant_research_amd import cut as C, ops_library as L  # noqa: F401
    cfg = cases.small_config()
    cfg["model"]["generator"]["ngf"], cfg["model"]["discriminator"]["ndf"] = 16, 16
    B, S = 2, 64

    def make():
        C.set_seed(42)
        gen, disc = C.build_models(cfg, "cpu")
        return C.CutTrainer(gen, disc, cfg, B, S, device=DEV, amp=False)
    g = torch.Generator().manual_seed(3)
    ph, mo = (torch.rand(B, 3, S, S, generator=g) * 2 - 1), (torch.rand(B, 3, S, S, generator=g) * 2 - 1)
    tr = make()
    ref = tr.train_step(0, ph.to(DEV), mo.to(DEV), tr.sample_randomness(torch.Generator().manual_seed(9)))
    tr2 = make()
    side = torch.cuda.Stream()
    big = torch.randn(4096, 4096, device=DEV)
    with torch.cuda.stream(side):
        for _ in range(20):                      # keeps `side` busy: the inputs below are ready long after the host reaches train_step
            big = big @ big * 1e-4
        p2 = ph.to(DEV, non_blocking=True) + 0 * big[0, 0]
        m2 = mo.to(DEV, non_blocking=True) + 0 * big[0, 1]
        got = tr2.train_step(0, p2, m2, tr2.sample_randomness(torch.Generator().manual_seed(9)))
        fake_side = tr2.generated().clone()      # read on the caller's stream right after the step
    torch.cuda.synchronize()
    assert got == ref and torch.equal(fake_side, tr.generated())
    # op-level library: the same op from two streams (two plans), both equal to F.conv2d
    x, w = torch.randn(2, 16, 12, 12, device=DEV), torch.randn(8, 16, 3, 3, device=DEV) * 0.1
    want = torch.nn.functional.conv2d(x.cpu(), w.cpu(), padding=1)
    y0 = torch.ops.mi355x_gan.conv2d_fwd(x, w, None, 1, 1, L.PAD_ZERO, 0)
    with torch.cuda.stream(side):
        xs = x + 0 * (big @ big)[0, 0]
        y1 = torch.ops.mi355x_gan.conv2d_fwd(xs, w, None, 1, 1, L.PAD_ZERO, 0)
    torch.cuda.synchronize()
    np.testing.assert_allclose(y0.cpu().numpy(), want.numpy(), rtol=1e-4, atol=1e-4)
    assert torch.equal(y0, y1)


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_in_bwd_parts_raw_sums_twins(dtype):
    """gan_in_bwd_parts with parts_mode 2 (sum g, sum g x against the RAW norm input, in any number of partials): equal to the two-pass
    gan_in_bwd on the same gradient, on the GPU and in the CPU statement."""
    from gan_variant_research_amd._lib import ACT_NONE
    B, H, W, Cc, NP = 3, 12, 20, 256, 4
    tw = Twin(dtype, seed=13)
    xc, xg = tw.view(B, H, W, Cc, 0)
    gc, gg = tw.view(B, H, W, Cc, 1)
    d1c, d1g = tw.view(B, H, W, Cc, 2, rand=False)
    d2c, d2g = tw.view(B, H, W, Cc, 2, rand=False)
    xf = xc.nhwc().float()
    stc, stg = tw.f32(torch.stack([xf.mean((1, 2)), 1.0 / torch.sqrt(xf.var((1, 2), unbiased=False) + 1e-5)], -1).reshape(-1))
    from tests.emulator import _fold
    gf = _fold(gc, True)                                            # what the kernel folds from the padded gradient
    rows = torch.chunk(torch.arange(H), NP)                         # NP partials per image: row bands
    assert len(rows) == NP
    parts = torch.stack([torch.stack([gf[:, r].sum((1, 2)), (gf[:, r] * xf[:, r]).sum((1, 2))], -1) for r in rows], 1).reshape(-1)   # [B][NP][C][2]
    pc, pg = tw.f32(parts)
    for ctx, x, g, d1, d2, st, p in ((tw.c, xc, gc, d1c, d2c, stc, pc), (tw.g, xg, gg, d1g, d2g, stg, pg)):
        ctx.ops.in_bwd_parts(x, st, ACT_NONE, g, True, d1, p, NP, 2)()
        ctx.ops.in_bwd(x, st, ACT_NONE, g, True, None, d2, ctx.f32(B * 96 * Cc * 2 + B * Cc * 2 + (B * 1024 + 32) * Cc))()
    torch.cuda.synchronize()
    rtol, atol = TOL[dtype]
    tw.check(rtol, atol)
    np.testing.assert_allclose(d1g.t.float().cpu().numpy(), d2g.t.float().cpu().numpy(), rtol=rtol, atol=atol)
