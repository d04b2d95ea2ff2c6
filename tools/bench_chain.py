"""The backward chain of one residual block on cold operands: plain padded-domain input gradient against the chain epilogues
(ConvLayer.dgrad(chain=...)), and the HBM-bound passes they replace / keep (gan_in_bwd two-pass, gan_in_bwd_parts, gan_fold_add).
Every launch cycles through K operand sets (> 256 MiB together) so that the epilogue operands come from HBM as they do in the step.
usage: bench_chain.py [B] [K]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_variant_research_amd import BF16
from gan_variant_research_amd._lib import ACT_NONE, ACT_RELU
from gan_variant_research_amd.convplan import ConvLayer
from gan_variant_research_amd.runtime import Ctx, HipOps

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
H, C = 64, 256
ctx = Ctx(HipOps(dev), dev, BF16)
ops = ctx.ops
w = torch.randn(C, C, 3, 3, device=dev) * 0.05
layer = ConvLayer(ctx, w, None, torch.zeros_like(w), None, 3, 1, 1)


def rnd(halo):
    v = ctx.view(B, H, H, C, halo)
    v.t.normal_()
    return v


dy = [rnd(2) for _ in range(K)]
ypad = [rnd(1) for _ in range(K)]      # saved relu output with halo (mode 1 operand)
xraw = [rnd(0) for _ in range(K)]      # raw norm input (mode 2 operand, in_bwd x)
out = [ctx.view(B, H, H, C, 1) for _ in range(K)]
gplain = [rnd(0) for _ in range(K)]
dxo = [ctx.view(B, H, H, C, 2) for _ in range(K)]
stats = ctx.f32(B * C * 2, 1.0)
ws = ctx.f32(B * 96 * C * 2 + B * C * 2 + (B * 1024 + 32) * C)
pa = ctx.f32(B * 96 * C * 2)
cases = {
    "dgrad plain (padded domain)": [layer.dgrad(dy[i], out[i], padded_domain=True) for i in range(K)],
    "dgrad + relu sums (stats_mode 1)": [layer.dgrad(dy[i], out[i], padded_domain=True, chain={"operand": ypad[i], "ws": pa}) for i in range(K)],
    "in_bwd relu fold (two passes)": [[ops.in_bwd(xraw[i], stats, ACT_RELU, out[i], True, None, dxo[i], ws)] for i in range(K)],
    "in_bwd plain (two passes)": [[ops.in_bwd(xraw[i], stats, ACT_NONE, gplain[i], False, None, dxo[i], ws)] for i in range(K)],
    "in_bwd_parts relu fold": [[ops.in_bwd_parts(xraw[i], stats, ACT_RELU, out[i], True, dxo[i], pa, 16, 1)] for i in range(K)],
    "fold_add": [[ops.fold_add(gplain[i], out[i], True, gplain[(i + 1) % K])] for i in range(K)],
}
for o in layer.repack_ops():
    o()
torch.cuda.synchronize()
print(f"B={B}, {K} operand sets of {B * H * H * C * 2 / 1e6:.0f} MB tensors")
for name, sets in cases.items():
    for s in sets:
        for o in s:
            o()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    R = 4
    e0.record()
    for _ in range(R):
        for s in sets:
            for o in s:
                o()
    e1.record()
    torch.cuda.synchronize()
    print(f"  {name:36s} {e0.elapsed_time(e1) / (R * K) * 1e3:8.1f} us")
