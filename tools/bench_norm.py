"""Times the InstanceNorm family on one shape: python tools/bench_norm.py [B H W C]  (bf16).
Every case is replayed back to back between two HIP events (operands stay cache-resident: optimistic for HBM-bound passes; judge the
in-step figures by the rocprofv3 profile of bench.py)."""
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
dx2 = ctx.view(B, H, W, Cc, 2)
stats = torch.zeros(B * Cc * 2, device=dev)
ws = torch.zeros(B * 96 * Cc * 2 + B * Cc * 2 + (B * 1024 + 32) * Cc, device=dev)
bg = torch.zeros(Cc, device=dev)
nbytes = B * H * W * Cc * 2
npart = ops.in_partial_count(x)
bp = torch.zeros(ops.in_bwd_bias_parts(x) * Cc, device=dev)
cases = {
    "in_stats  (read x)": ([ops.in_stats(x, 1e-5, stats, ws)], 1),
    "in_partial (read x)": ([ops.in_partial(x, ws)], 1),
    "in_apply  (read x, write y+halo)": ([ops.in_apply(x, stats, _lib.ACT_RELU, None, y, _lib.HALO_REFLECT)], 2),
    f"in_apply_parts[{npart}]": ([ops.in_apply_parts(x, ws, npart, 1e-5, stats, _lib.ACT_RELU, None, y, _lib.HALO_REFLECT)], 2),
    "in_apply+res (read x,res, write y)": ([ops.in_apply(x, stats, _lib.ACT_NONE, dx, y, _lib.HALO_REFLECT)], 3),
    "in_bwd fold (read x,gy, write dx)": ([ops.in_bwd(x, stats, _lib.ACT_RELU, gy, True, None, dx2, ws)], 3),
    "in_bwd nofold noact": ([ops.in_bwd(x, stats, _lib.ACT_NONE, gy, False, None, dx2, ws)], 3),
    "in_bwd_bias_deferred fold": ([ops.in_bwd_bias_deferred(x, stats, _lib.ACT_RELU, gy, True, None, dx2, ws, bp)], 3),
    "fold_add (read a,g, write out)": ([ops.fold_add(x, gy, True, dx)], 3),
}
print(f"shape B={B} {H}x{W}x{Cc} bf16, {nbytes/1e6:.1f} MB per tensor, GAN_NORM_UNR={os.environ.get('GAN_NORM_UNR', '4')} GAN_NORM_WORK={os.environ.get('GAN_NORM_WORK', 'default')}")
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
