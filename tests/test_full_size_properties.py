"""Size-independent properties at BASELINE.json's full size (CUT 256x256, batch 16 per GPU, bf16), where the oracle is too slow to
run in a test: determinism of the three-stream step, batch independence of the generator (neither network has cross-sample
statistics, patchnce_cut.py:69-101 / InstanceNorm), and adjointness of the convolution kernels -- <conv(x), y> = <x, dgrad(y)> and
<conv_w(x), y> = <w, wgrad(x, y)> -- for the layer geometries that dominate the step, on the kernels the full-size launches select
(range-patch 288-row tile, 7x7 window kernels)."""
import numpy as np
import pytest
import torch

from gan_variant_research_amd import BF16
from gan_variant_research_amd import cut as C
from gan_variant_research_amd.convplan import ConvLayer
from gan_variant_research_amd.runtime import Ctx, HipOps, cpad
from gan_variant_research_amd._lib import HALO_REFLECT, HALO_ZERO
from tests import cases

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
B, S = 16, 256


def _trainer():
    import bench
    cfg = bench.default_config()
    C.set_seed(42)
    gen, disc = C.build_models(cfg, DEV)
    return C.CutTrainer(gen, disc, cfg, B, S, device=DEV, amp=True), gen


def _inputs():
    g = torch.Generator().manual_seed(3)
    return (torch.rand(B, 3, S, S, generator=g) * 2 - 1).to(DEV), (torch.rand(B, 3, S, S, generator=g) * 2 - 1).to(DEV)


def test_full_size_step_is_deterministic_and_finite():
    photos, monets = _inputs()

    def run():
        tr, _ = _trainer()
        rg = torch.Generator().manual_seed(9)
        out = [tr.train_step(step, photos, monets, tr.sample_randomness(rg)) for step in range(2)]     # step 0 includes R1
        torch.cuda.synchronize()
        return out, tr.generated().clone(), tr.opt_G.flat_p.clone()
    la, ia, pa = run()
    lb, ib, pb = run()
    assert la == lb and torch.equal(ia, ib) and torch.equal(pa, pb)
    assert all(np.isfinite(v) for d in la for v in d.values())
    assert 0.5 < la[0]["d_loss"] < 1.5 and 4.0 < la[0]["nce"] < 7.0 and la[0]["r1"] > 0.0      # hinge at init ~1, NCE ~ log(256) = 5.5
    assert float(ia.abs().max()) <= 1.0


def test_generator_is_batch_independent_at_full_size():
    """G(x)[i] does not depend on the other images of the batch: the 16-image pass equals two 8-image passes bit for bit."""
    _, gen = _trainer()
    gen.compute_dtype = BF16
    x, _ = _inputs()
    with torch.no_grad():
        full = gen(x)
        halves = torch.cat([gen(x[:8]), gen(x[8:])])
    assert torch.equal(full, halves), float((full - halves).abs().max())


@pytest.mark.parametrize("geom", [(256, 256, 3, 1, 1, 64, True), (64, 3, 7, 1, 3, 256, True), (3, 64, 7, 1, 3, 256, True), (64, 128, 3, 2, 1, 256, False),
                                  (256, 128, 3, 2, 1, 64, False, True)])
def test_convolution_adjoints_at_full_size(geom):
    cin, cout, k, s, p, H, reflect = geom[:7]
    tr = len(geom) > 7
    ctx = Ctx(HipOps(torch.device(DEV)), DEV, BF16)
    g = torch.Generator().manual_seed(1)
    w = (torch.randn((cin, cout, k, k) if tr else (cout, cin, k, k), generator=g) * (0.5 / (cin * k * k) ** 0.5)).to(DEV)
    b = torch.zeros(cout, device=DEV)
    gw, gb = torch.zeros_like(w), torch.zeros_like(b)
    layer = ConvLayer(ctx, w, b, gw, gb, k, s, p, tr)
    x = torch.randn(B, cin, H, H, generator=g).bfloat16().float()
    Ho = H * 2 if tr else (H + 2 * p - k) // s + 1
    y = torch.randn(B, cout, Ho, Ho, generator=g).bfloat16().float()
    xin = cases.to_view(ctx, x, max(p, 1), HALO_REFLECT if reflect else HALO_ZERO)
    out = ctx.view(B, Ho, Ho, cpad(cout), 0)
    fwd = layer.fwd(xin, out)
    if tr or s == 2:
        dyv = cases.to_view(ctx, y, 1, HALO_ZERO)
        dx = ctx.view(B, H, H, cpad(cin), 0)
        bwd, fold = layer.dgrad(dyv, dx), False
    else:
        dyv = cases.to_view(ctx, y, k - 1, HALO_ZERO)
        dx = ctx.view(B, H, H, cpad(cin), p)
        bwd, fold = layer.dgrad(dyv, dx, padded_domain=True), True
    wg = layer.wgrad(xin, dyv if not tr else cases.to_view(ctx, y, 1, HALO_ZERO), accumulate=False, bias_too=False) if not tr else layer.wgrad(xin, dyv, accumulate=False, bias_too=False)
    for op in layer.repack_ops():
        op()
    for op in fwd + bwd + wg:
        op()
    torch.cuda.synchronize()
    conv_x = out.nhwc().float()[..., :cout].permute(0, 3, 1, 2)
    if fold:      # gradient wrt the reflect-padded input: fold it back onto the image (reflection_pad2d_backward)
        gp = dx.padded().float()[..., :cin].permute(0, 3, 1, 2)
        probe = torch.zeros(B, cin, H, H, device=DEV, requires_grad=True)
        (torch.nn.functional.pad(probe, (p, p, p, p), mode="reflect") * gp).sum().backward()
        dgrad_y = probe.grad
    else:
        dgrad_y = dx.nhwc().float()[..., :cin].permute(0, 3, 1, 2)
    xd, yd = x.to(DEV).double(), y.to(DEV).double()
    lhs = float((conv_x.double() * yd).sum())
    rhs_x = float((xd * dgrad_y.double()).sum())
    rhs_w = float((w.bfloat16().double() * gw.double()).sum())
    scale = float(conv_x.double().norm() * yd.norm())
    assert abs(lhs - rhs_x) < 2e-3 * scale, (lhs, rhs_x, scale)       # bf16 outputs: each side carries ~2^-9 relative rounding per element
    assert abs(lhs - rhs_w) < 2e-3 * scale, (lhs, rhs_w, scale)
