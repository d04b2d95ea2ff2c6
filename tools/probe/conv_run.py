"""Launches the 3x3 256->256 residual-layer convolution N times (forward or reflect-padded input gradient): the target of
`rocprofv3 --pmc ... -- python3 tools/probe/conv_run.py fwd|dgrad [B] [N]` counter passes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
N = int(sys.argv[3]) if len(sys.argv) > 3 else 20
from gan_variant_research_amd import BF16
from gan_variant_research_amd.convplan import ConvLayer
from gan_variant_research_amd.runtime import Ctx, HipOps
ctx = Ctx(HipOps(dev), dev, BF16)
w = torch.randn(256, 256, 3, 3, device=dev) * 0.05
b = torch.zeros(256, device=dev)
layer = ConvLayer(ctx, w, b, torch.zeros_like(w), torch.zeros_like(b), 3, 1, 1)
for o in layer.repack_ops(): o()
if mode == "fwd":
    x = ctx.view(B, 64, 64, 256, 1); x.t.normal_()
    ops = layer.fwd(x, ctx.view(B, 64, 64, 256, 0))
else:
    dy = ctx.view(B, 64, 64, 256, 2); dy.t.normal_()
    ops = layer.dgrad(dy, ctx.view(B, 64, 64, 256, 1), padded_domain=True)
for _ in range(N):
    for o in ops: o()
torch.cuda.synchronize()
