"""Times the InstanceNorm family on one shape: python tools/bench_norm.py [B H W C]  (bf16)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_variant_research_amd import BF16
from gan_variant_research_amd.runtime import Ctx, HipOps
from gan_variant_research_amd import _lib
a = [int(v) for v in sys.argv[1:5]] if len(sys.argv) >= 5 else [16, 64, 64, 256]
B, H, W, Cc = a
dev = torch.device("cuda:0")
ops = HipOps(dev)
ctx = Ctx(ops, dev, BF16)
x = ctx.view(B, H, W, Cc, 0); x.t.normal_()
y = ctx.view(B, H, W, Cc, 1)
gy = ctx.view(B, H, W, Cc, 1); gy.t.normal_()
dx = ctx.view(B, H, W, Cc, 0)
stats = torch.zeros(B * Cc * 2, device=dev)
ws = torch.zeros(B * 96 * Cc * 2 + B * Cc * 2 + (B * 1024 + 32) * Cc, device=dev)
bg = torch.zeros(Cc, device=dev)
nbytes = B * H * W * Cc * 2
cases = {
    "in_stats  (read x)": ([ops.in_stats(x, 1e-5, stats, ws)], 1),
    "in_apply  (read x, write y+halo)": ([ops.in_apply(x, stats, _lib.ACT_RELU, None, y, _lib.HALO_REFLECT)], 2),
    "in_apply+res (read x,res, write y)": ([ops.in_apply(x, stats, _lib.ACT_NONE, dx, y, _lib.HALO_REFLECT)], 3),
    "in_bwd fold (2x read x,gy, write dx)": ([ops.in_bwd(x, stats, _lib.ACT_RELU, gy, True, None, dx, ws)], 5),
    "in_bwd nofold": ([ops.in_bwd(x, stats, _lib.ACT_RELU, gy, False, None, dx, ws)], 5),
    "in_bwd nofold noact": ([ops.in_bwd(x, stats, _lib.ACT_NONE, gy, False, None, dx, ws)], 5),
    "in_bwd_bias fold": ([ops.in_bwd_bias(x, stats, _lib.ACT_RELU, gy, True, None, dx, ws, bg, Cc, False)], 5),
    "fold_add (read a,g, write out)": ([ops.fold_add(x, gy, True, dx)], 3),
}
for name, (ol, mult) in cases.items():
    for _ in range(3):
        for o in ol: o()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        for o in ol: o()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name:40s} {us:8.1f} us  {mult * nbytes / us / 1e6:6.2f} TB/s (algorithmic {mult}x{nbytes/1e6:.1f} MB)")
