"""In-kernel s_memtime stamps of the range-patch convolution (GAN_PATCH_STAMPS): cycles per phase of wave 0 of every block.
usage: stamps.py [fwd|dgrad|chain] [B] [random|zeros|relu] [alt]   (GAN_PATCH_BN=128 in the environment: the 128-channel tiles)
"alt" puts an HBM-bound pass (a 71 MB device copy, like the InstanceNorm apply between two convolutions of the step) after every
launch.  The third argument picks the operand data; the run also prints the clock the chip held inside the kernel (MI355X_MICROARCH 'DVFS
give-back' item 6: d(s_memtime) / d(s_memrealtime) x 100 MHz over wave 0 of every block, after >= 2 s of back-to-back launches)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
dev = torch.device("cuda:0")
buf = torch.zeros(256 * 32, dtype=torch.int64, device=dev)
# GAN_STAMPS_W0=1: diagnostic bit 1 of the stamp pointer -- every weight fetch reads fragment block 0 (L1-resident; results wrong): what do
# the L2 -> L1 weight streams cost in time and, at the board power limit, in clock?
os.environ["GAN_PATCH_STAMPS"] = str(buf.data_ptr() | (2 if os.environ.get("GAN_STAMPS_W0") else 0))
mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
data = sys.argv[3] if len(sys.argv) > 3 else "random"


def fill(t):
    if data == "zeros": t.zero_()
    elif data == "relu": t.normal_().clamp_(min=0)      # what the residual blocks' first convolution reads
    else: t.normal_()


from gan_variant_research_amd import BF16
from gan_variant_research_amd.convplan import ConvLayer
from gan_variant_research_amd.runtime import Ctx, HipOps
ctx = Ctx(HipOps(dev), dev, BF16)
w = torch.randn(256, 256, 3, 3, device=dev) * (0.0 if data == "zeros" else 0.05)
b = torch.zeros(256, device=dev)
layer = ConvLayer(ctx, w, b, torch.zeros_like(w), torch.zeros_like(b), 3, 1, 1)
if mode == "fwd":
    x = ctx.view(B, 64, 64, 256, 1); fill(x.t)
    y = ctx.view(B, 64, 64, 256, 0)
    ops = layer.fwd(x, y)
else:
    dy = ctx.view(B, 64, 64, 256, 2); fill(dy.t)
    dx = ctx.view(B, 64, 64, 256, 1)
    chain = None
    if mode == "chain":      # backward-chain epilogue (stats_mode 1); operands evicted between launches: cold, as in the step
        opd = ctx.view(B, 64, 64, 256, 1); opd.t.normal_()
        chain = {"operand": opd, "ws": ctx.f32(B * 96 * 256 * 2)}
        layer.bias = layer.bias_k = None
    ops = layer.dgrad(dy, dx, padded_domain=True, chain=chain)
    flush = torch.empty(300 << 20, dtype=torch.uint8, device=dev)
    ops = ops + [lambda: flush.zero_()]        # evict the operands between launches
if len(sys.argv) > 4 and sys.argv[4] == "alt":
    big = torch.empty(B * 66 * 66 * 256, dtype=torch.bfloat16, device=dev).normal_()
    big2 = torch.empty_like(big)
    ops = ops + [lambda: big2.copy_(big)]
if data != "zeros":
    for o in layer.repack_ops(): o()     # after planning (the operand copies exist only then); "zeros" leaves the weight copy zero
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
import time
t0 = time.time()
while time.time() - t0 < 2.5:          # reach the clock the chip settles at under this load
    for _ in range(200):
        for o in ops: o()
    torch.cuda.synchronize()
e0.record()
for _ in range(20):
    for o in ops: o()
e1.record()
torch.cuda.synchronize()
print(f"{mode} B={B}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch (with the stamp code active)")
s = buf.view(256, 32).cpu().numpy().astype(np.int64)
n = int((s[0, :30] != 0).sum())
clk = (s[:, n - 1] - s[:, 0]) / np.maximum(s[:, 31] - s[:, 30], 1) * 100.0
print(f"data = {data}: in-kernel clock median {np.median(clk):.0f} MHz (min {clk.min():.0f}, max {clk.max():.0f})")
d = np.diff(s[:, :n], axis=1)
print(f"{n} stamps per block; wave 0 lifetime median {np.median(s[:, n - 1] - s[:, 0]):.0f} cycles")
for i, col in enumerate(d.T):
    print(f"  phase {i:2d} median {np.median(col):9.0f}  min {col.min():9.0f}  max {col.max():9.0f}")
print("block start spread:", s[:, 0].max() - s[:, 0].min(), " end spread:", s[:, n - 1].max() - s[:, n - 1].min())
