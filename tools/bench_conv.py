"""Per-layer microbenchmark of the MFMA kernels at the bench shapes (B=16, 256x256 CUT): forward, dgrad, wgrad TFLOP/s."""
import sys
import time
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_variant_research_amd import BF16, F32
from gan_variant_research_amd.convplan import ConvLayer
from gan_variant_research_amd.runtime import Ctx, HipOps, cpad

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dtype = BF16
ctx = Ctx(HipOps(dev), dev, dtype)


def timeit(ops, iters=20, warm_s=0.5):
    # MFMA kernels on real operands run at the board power limit: time them at the clock the chip settles at under their own load
    t0 = time.time()
    while time.time() - t0 < warm_s:
        for _ in range(20):
            for o in ops: o()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        for o in ops: o()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


# name, cin, cout, k, s, p, transposed, H(in), reflect
LAYERS = [("g.init 7x7 3->64", 3, 64, 7, 1, 3, False, 256, True), ("g.down1 3x3s2 64->128", 64, 128, 3, 2, 1, False, 256, False),
          ("g.down2 3x3s2 128->256", 128, 256, 3, 2, 1, False, 128, False), ("g.res 3x3 256->256", 256, 256, 3, 1, 1, False, 64, True),
          ("g.up1 convT 256->128", 256, 128, 3, 2, 1, True, 64, False), ("g.up2 convT 128->64", 128, 64, 3, 2, 1, True, 128, False),
          ("g.out 7x7 64->3", 64, 3, 7, 1, 3, False, 256, True), ("d.1 4x4s2 3->64", 3, 64, 4, 2, 1, False, 256, False),
          ("d.2 4x4s2 64->128", 64, 128, 4, 2, 1, False, 128, False), ("d.3 4x4s2 128->256", 128, 256, 4, 2, 1, False, 64, False),
          ("d.4 4x4s1 256->512", 256, 512, 4, 1, 1, False, 32, False), ("d.5 4x4s1 512->1", 512, 1, 4, 1, 1, False, 31, False)]
only = sys.argv[2] if len(sys.argv) > 2 else None
print(f"{'layer':28s} {'GFLOP':>8s} {'fwd us':>8s} {'TF/s':>7s} {'dgrad us':>9s} {'TF/s':>7s} {'wgrad us':>9s} {'TF/s':>7s}")
for name, cin, cout, k, s, p, tr, H, reflect in LAYERS:
    if only and only not in name:
        continue
    w = torch.randn((cin, cout, k, k) if tr else (cout, cin, k, k), device=dev) * 0.05
    b = torch.zeros(cout, device=dev)
    layer = ConvLayer(ctx, w, b, torch.zeros_like(w), torch.zeros_like(b), k, s, p, tr)
    Ho = 2 * H if tr else (H + 2 * p - k) // s + 1
    x = ctx.view(B, H, H, cpad(cin), max(p, 1)); x.t.normal_()
    y = ctx.view(B, Ho, Ho, cpad(cout), 0)
    if tr:
        dy = ctx.view(B, Ho, Ho, cpad(cout), 1); dx = ctx.view(B, H, H, cpad(cin), 0); dg = layer.dgrad(dy, dx)
    elif s == 2:
        dy = ctx.view(B, Ho, Ho, cpad(cout), 1); dx = ctx.view(B, H, H, cpad(cin), 0); dg = layer.dgrad(dy, dx)
    elif reflect:
        dy = ctx.view(B, Ho, Ho, cpad(cout), k - 1); dx = ctx.view(B, H, H, cpad(cin), p); dg = layer.dgrad(dy, dx, padded_domain=True)
    else:
        dy = ctx.view(B, Ho, Ho, cpad(cout), k - 1 - p); dx = ctx.view(B, H, H, cpad(cin), 0); dg = layer.dgrad(dy, dx)
    dy.t.normal_()
    fw = layer.fwd(x, y)
    for o in layer.repack_ops(): o()      # after planning: the operand copies exist only then (packed earlier, the launches would multiply by zeros)
    flop = 2.0 * B * (H * H if tr else Ho * Ho) * cin * cout * k * k / (4 if tr else 1) * (1 if not tr else 1)
    if tr:
        flop = 2.0 * B * H * H * cin * cout * 9
    tf = timeit(fw); td = timeit(dg); tw = timeit(layer.wgrad(x, dy, False, bias_too=False))
    print(f"{name:28s} {flop/1e9:8.2f} {tf*1e3:8.1f} {flop/tf/1e9:7.1f} {td*1e3:9.1f} {flop/td/1e9:7.1f} {tw*1e3:9.1f} {flop/tw/1e9:7.1f}")
