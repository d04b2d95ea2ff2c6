"""Does the DATA decide the clock?  Replays the first residual convolution of the CUT step (forward, 32 images) alone, back to back,
first on the step's own activations / weights, then with each operand replaced in place by synthetic data, and prints the time per
launch, the in-kernel clock (GAN_PATCH_STAMPS, see tools/step_clock.py) and the board power.  Then the same layer shape built stand-alone
with its weight copy still ZERO (a repack program built before the forward call was planned packs nothing: how the round-2 / early
round-3 microbenchmarks ran, ConvLayer.repack_ops now refuses) and packed.  Result (profiles/r03_data_power.txt): every non-zero data
set runs at the 1400 W board limit and ~2.0 GHz; zero weights draw 1160 W and run at 2.39 GHz, 20 % faster at the same cycle count.

    python tools/probe/data_power.py
"""
import glob
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dev = torch.device("cuda:0")
buf = torch.zeros(256 * 32, dtype=torch.int64, device=dev)
os.environ["GAN_PATCH_STAMPS"] = str(buf.data_ptr())
os.environ["GAN_PATCH_STAMPS_SEL"] = "256,256,32,0,9"
os.environ.setdefault("GAN_SINGLE_STREAM", "1")
import bench  # noqa: E402
from gan_variant_research_amd import cut as C  # noqa: E402


def read(p):
    try:
        with open(p) as f:
            return float(f.read().strip())
    except Exception:
        return float("nan")


nodes = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")
before = {h: read(h + "/power1_input") for h in nodes}
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
t0 = time.time()
while time.time() - t0 < 1.0:
    (a @ a)
torch.cuda.synchronize()
HW = max(nodes, key=lambda h: read(h + "/power1_input") - before[h])

B = 16
cfg = bench.default_config()
torch.manual_seed(0)
G, D = C.build_models(cfg, dev)
tr = C.CutTrainer(G, D, cfg, B, 256, device=dev, amp=True)
g = torch.Generator().manual_seed(1)
ph = (torch.rand(B, 3, 256, 256, generator=g) * 2 - 1).to(dev)
mo = (torch.rand(B, 3, 256, 256, generator=g) * 2 - 1).to(dev)
for s in range(1, 4):
    tr.train_step(s, ph, mo)
torch.cuda.synchronize()
op = next(o for o in tr.prog_gfwd.ops if getattr(o, "conv", None) is not None and o.conv.w_frag and o.conv.Cin == 256 and o.conv.Nst == 256 and o.conv.ntaps == 9)
c = op.conv


def stats(t, name):
    f = t.float()
    print(f"   {name}: {tuple(t.shape)} {t.dtype}, zeros {float((f == 0).float().mean()):.3f}, mean {float(f.mean()):+.4f}, std {float(f.std()):.4f}, max |.| {float(f.abs().max()):.3f}")


def run(tag, seconds=2.5, op=None):
    op = op or globals()["op"]
    p = []
    stop = [False]

    def sample():
        while not stop[0]:
            p.append(read(HW + "/power1_input") * 1e-6)
            time.sleep(0.01)
    th = threading.Thread(target=sample, daemon=True); th.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time()
    while time.time() - t0 < seconds:
        e0.record()
        for _ in range(100):
            op()
        e1.record(); torch.cuda.synchronize()
    stop[0] = True; th.join()
    s = buf.view(256, 32).cpu().numpy().astype(np.int64)
    n = int((s[0, :30] != 0).sum())
    clk = (s[:, n - 1] - s[:, 0]) / np.maximum(s[:, 31] - s[:, 30], 1) * 100.0
    pw = p[len(p) // 2:]
    print(f"{tag}: {e0.elapsed_time(e1) * 10:.1f} us per launch, clock {np.median(clk):.0f} MHz, {np.median(s[:, n - 1] - s[:, 0]):.0f} cycles, power {sum(pw) / len(pw):.0f} W")
    print("      phases: " + " ".join(f"{np.median(col):.0f}" for col in np.diff(s[:, :n], axis=1).T))
    buf.zero_()


xt, wt = c.x.t, c.w
stats(xt, "activations (halo-NHWC view storage)")
stats(wt, "weights (fragment-major copy)")
if c.bias is not None:
    stats(c.bias, "bias")
run("the step's own data")
x_keep, w_keep = xt.clone(), wt.clone()
wt.copy_((torch.randn(wt.shape, device=dev) * 0.05).to(wt.dtype))
run("weights <- N(0, 0.05)")
wt.copy_(w_keep)
xt.copy_(torch.randn(xt.shape, device=dev).clamp_(min=0).to(xt.dtype))
run("activations <- relu(N(0, 1))")
xt.copy_(torch.randn(xt.shape, device=dev).to(xt.dtype))
run("activations <- N(0, 1)")
xt.copy_(x_keep)
wt.copy_((w_keep.float() * 2.5).to(wt.dtype))
run("weights x 2.5")
wt.copy_(w_keep)
xt.copy_((x_keep.float() * 0.25).to(xt.dtype))
run("activations x 0.25")
xt.copy_(x_keep)
run("the step's own data again")

# the same layer shape built stand-alone in this process (what tools/probe/stamps.py times)
from gan_variant_research_amd import BF16  # noqa: E402
from gan_variant_research_amd.convplan import ConvLayer  # noqa: E402
from gan_variant_research_amd.runtime import Ctx, HipOps  # noqa: E402
ctx = Ctx(HipOps(dev), dev, BF16)
w = torch.randn(256, 256, 3, 3, device=dev) * 0.05
b = torch.zeros(256, device=dev)
layer = ConvLayer(ctx, w, b, torch.zeros_like(w), torch.zeros_like(b), 3, 1, 1)
x = ctx.view(32, 64, 64, 256, 1); x.t.normal_().clamp_(min=0)
y = ctx.view(32, 64, 64, 256, 0)
ops = layer.fwd(x, y)
print("stand-alone layer:", len(ops), "launches;", {k: getattr(ops[0].conv, k) for k in ("act", "tile_rows", "tile_cols", "out_y0", "out_x0")}, "stats" if ops[0].conv.stats is not None else "no stats",
      "| step op:", {k: getattr(c, k) for k in ("act", "tile_rows", "tile_cols", "out_y0", "out_x0")}, "stats" if c.stats is not None else "no stats")
run("stand-alone layer, weight copy still ZERO (not packed yet)", op=ops[0])
for o in layer.repack_ops(): o()
run("stand-alone layer, weights packed", op=ops[0])

# which part of the step's launch costs the power?  the same descriptor without the fused statistics / without the bias
import dataclasses  # noqa: E402
c_nostats = dataclasses.replace(c, stats=None)
run("step op WITHOUT the fused statistics", op=tr.ops.conv_igemm(c_nostats))
c_nobias = dataclasses.replace(c, bias=None)
run("step op without the bias (statistics on)", op=tr.ops.conv_igemm(c_nobias))
run("step op again", op=tr.ops.conv_igemm(dataclasses.replace(c)))

