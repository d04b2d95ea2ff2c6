"""The clock the chip holds inside the range-patch kernels DURING the train step, and the step time against the time the chip
has been under load.

    python tools/step_clock.py [batch] [steps] [rows,cols,batch,chain,taps]

The optional selection (GAN_PATCH_STAMPS_SEL) restricts the stamps to one kernel of the step, e.g. 256,256,32,0,9 = the forward of the
residual blocks on 32 images: its phases (cycles) and clock inside the running step, to set against tools/probe/stamps.py (alone).

GAN_PATCH_STAMPS makes wave 0 of every conv_patch block stamp s_memtime (shader cycles) and s_memrealtime (100 MHz) at its
first and latest phase (conv_patch.hip `stamp`); the quotient is the in-kernel clock (MI355X_MICROARCH 'DVFS give-back' item 6).
Every launch overwrites its blocks' slots, so a read after a program shows that program's last range-patch launch."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")
buf = torch.zeros(256 * 32, dtype=torch.int64, device=dev)
os.environ["GAN_PATCH_STAMPS"] = str(buf.data_ptr())
if len(sys.argv) > 3:
    os.environ["GAN_PATCH_STAMPS_SEL"] = sys.argv[3]
import bench  # noqa: E402
from gan_variant_research_amd import cut as C  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
NSTEP = int(sys.argv[2]) if len(sys.argv) > 2 else 200
cfg = bench.default_config()
torch.manual_seed(0)
G, D = C.build_models(cfg, dev)
tr = C.CutTrainer(G, D, cfg, B, 256, device=dev, amp=True)
g = torch.Generator().manual_seed(1)
ph = (torch.rand(B, 3, 256, 256, generator=g) * 2 - 1).to(dev)
mo = (torch.rand(B, 3, 256, 256, generator=g) * 2 - 1).to(dev)


def clock(tag):
    torch.cuda.synchronize()
    s = buf.view(256, 32).cpu().numpy().astype(np.int64)
    ok = (s[:, 31] > s[:, 30]) & (s[:, 1] > 0)
    if not ok.any():
        print(f"{tag}: no stamps")
        return
    n = (s[:, :30] != 0).sum(axis=1)
    last = s[np.arange(256), np.maximum(n - 1, 0)]
    clk = (last - s[:, 0])[ok] / (s[:, 31] - s[:, 30])[ok] * 100.0
    print(f"{tag}: in-kernel clock median {np.median(clk):.0f} MHz (p10 {np.percentile(clk, 10):.0f}, p90 {np.percentile(clk, 90):.0f}; {int(ok.sum())} blocks)")
    if len(sys.argv) > 3:
        nn = int(np.median(n[ok]))
        d = np.diff(s[ok][:, :nn], axis=1)
        life = (last - s[:, 0])[ok]
        print(f"      wave 0 lifetime median {np.median(life):.0f} cycles = {np.median(life / clk):.1f} us; phases (median cycles): " + " ".join(f"{np.median(c):.0f}" for c in d.T))
    buf.zero_()


torch.cuda.synchronize()
time.sleep(1.0)
t_load = time.time()
step = 1
for chunk in [5, 5, 10, 20, 40, 80, NSTEP]:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(chunk):
        tr.train_step(step, ph, mo)
        step += 1
    e1.record()
    torch.cuda.synchronize()
    print(f"steps {step - chunk:4d}..{step - 1:4d}: {e0.elapsed_time(e1) / chunk:7.3f} ms/step   ({time.time() - t_load:5.1f} s under load)")
    clock("   last range-patch launches of the step")
