"""bf16 vs fp8 (e4m3) on the bottleneck 3x3 256->256 convolution: forward and input gradient, python tools/bench_fp8.py [B] [H]."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_variant_research_amd import BF16, FP8
from gan_variant_research_amd.convplan import ConvLayer
from gan_variant_research_amd.runtime import Ctx, HipOps

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ctx = Ctx(HipOps(dev), dev, BF16)


def timeit(ops, iters=20):
    for _ in range(3):
        for o in ops: o()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        for o in ops: o()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


w = torch.randn(256, 256, 3, 3, device=dev) * 0.02
b = torch.zeros(256, device=dev)
layer = ConvLayer(ctx, w, b, torch.zeros_like(w), torch.zeros_like(b), 3, 1, 1)
x = ctx.view(B, H, H, 256, 1); x.t.normal_()
x8 = ctx.view(B, H, H, 256, 1, dtype=FP8)
y = ctx.view(B, H, H, 256, 0)
dy = ctx.view(B, H, H, 256, 2); dy.nhwc().normal_()
dy8 = ctx.view(B, H, H, 256, 2, dtype=FP8)
dxp = ctx.view(B, H, H, 256, 1)
amax, scale = torch.ones(B, device=dev) * 4.0, torch.zeros(B, device=dev)
f16, f8 = layer.fwd(x, y), layer.fwd8(x8, y)
d16, d8 = layer.dgrad(dy, dxp, padded_domain=True), layer.dgrad8(dy8, dxp, scale, padded_domain=True)
q_act, q_grad = [ctx.ops.quantize_fp8(x, x8)], [ctx.ops.quantize_fp8(dy, dy8, amax, scale)]
ctx.ops.pack_weight_batch([op.pack_args for op in layer.repack_ops()])()
q_act[0](); q_grad[0]()
flop = 2.0 * B * H * H * 256 * 256 * 9
print(f"3x3 256->256, B={B}, {H}x{H}: {flop/1e9:.1f} GFLOP")
for name, ops in (("forward bf16", f16), ("forward fp8", f8), ("dgrad bf16 (padded domain)", d16), ("dgrad fp8 (padded domain)", d8),
                  ("quantize activations", q_act), ("quantize gradient (per-image scale)", q_grad)):
    us = timeit(ops)
    print(f"{name:38s} {us:8.1f} us" + (f"  {flop/us/1e6:8.1f} TFLOP/s" if "quantize" not in name else f"  {B*(H+2)*(H+2)*256*3/us/1e6:8.2f} TB/s"))
