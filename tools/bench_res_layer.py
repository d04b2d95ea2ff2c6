"""Forward / input gradient / weight gradient of one 3x3 256->256 residual-layer convolution at any (B, H): python tools/bench_res_layer.py B H"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_variant_research_amd import BF16
from gan_variant_research_amd.convplan import ConvLayer
from gan_variant_research_amd.runtime import Ctx, HipOps
dev = torch.device("cuda:0")
B, H = int(sys.argv[1]), int(sys.argv[2])
ctx = Ctx(HipOps(dev), dev, BF16)
w = torch.randn(256, 256, 3, 3, device=dev) * 0.05
b = torch.zeros(256, device=dev)
layer = ConvLayer(ctx, w, b, torch.zeros_like(w), torch.zeros_like(b), 3, 1, 1)
x = ctx.view(B, H, H, 256, 1); x.t.normal_()
y = ctx.view(B, H, H, 256, 0)
dy = ctx.view(B, H, H, 256, 2); dy.t.normal_()
dx = ctx.view(B, H, H, 256, 1)
def timeit(ops, iters=20, warm_s=0.5):
    import time
    t0 = time.time()
    while time.time() - t0 < warm_s:       # the clock the chip settles at under this load (real operands: the board power limit)
        for _ in range(20):
            for o in ops: o()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        for o in ops: o()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
flop = 2.0 * B * H * H * 256 * 256 * 9
wg = layer.wgrad(x, dy, False, bias_too=False)
fw, dg = layer.fwd(x, y), layer.dgrad(dy, dx, padded_domain=True)
for o in layer.repack_ops(): o()      # after planning: the operand copies exist only then
for name, ops in (("fwd", fw), ("dgrad (padded domain)", dg), ("wgrad", wg[:1]), ("wgrad reduce", wg[1:])):
    us = timeit(ops)
    c = getattr(ops[0], "conv", None) or getattr(ops[0], "wgrad", None)
    kind = ("patch" if getattr(c, "w_frag", False) or getattr(c, "variant", 0) else "generic") if c is not None else ""
    print(f"B={B} {H}x{H} {name:22s} {us:8.1f} us  {flop/us/1e6 if 'reduce' not in name else 0:7.1f} TF/s  {kind} {getattr(c, 'nsplit', '')}")
