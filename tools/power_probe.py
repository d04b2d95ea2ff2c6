"""Board power and shader clock (hwmon / sysfs, sampled from a thread) under three loads: the CUT train step, the residual blocks' forward
convolution alone, and the InstanceNorm apply pass alone -- is the step power-limited?

    python tools/power_probe.py [batch]
"""
import glob
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gan_variant_research_amd import BF16  # noqa: E402
from gan_variant_research_amd import cut as C  # noqa: E402
from gan_variant_research_amd.convplan import ConvLayer  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16


def find_sensors():
    out = {}
    for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for name in ("power1_average", "power1_input", "power1_cap", "freq1_input", "freq2_input", "temp1_input", "temp2_input"):
            p = os.path.join(h, name)
            if os.path.exists(p):
                out.setdefault(h, {})[name] = p
    return out


def read(p):
    try:
        with open(p) as f:
            return float(f.read().strip())
    except Exception:
        return float("nan")


def own_sensor(sens):
    """The hwmon node of THIS process's GPU (the box shows all eight): the one whose power rises under a short load."""
    before = {h: read(v["power1_input"]) for h, v in sens.items() if "power1_input" in v}
    a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    t0 = time.time()
    while time.time() - t0 < 1.0:
        (a @ a)
    torch.cuda.synchronize()
    after = {h: read(sens[h]["power1_input"]) for h in before}
    h = max(before, key=lambda k: after[k] - before[k])
    return {h: sens[h]}


SENS = own_sensor(find_sensors())
for h, v in SENS.items():
    print(f"sensor {h}: power cap {read(v['power1_cap']) / 1e6:.0f} W" if "power1_cap" in v else f"sensor {h}")


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.stop = False
        self.rows = []

    def run(self):
        while not self.stop:
            row = {}
            for h, v in SENS.items():
                for k in ("power1_average", "power1_input", "freq1_input", "freq2_input", "temp1_input"):
                    if k in v:
                        row[(h[-6:], k)] = read(v[k])
            self.rows.append(row)
            time.sleep(0.02)


def measure(tag, fn, seconds=4.0):
    fn(); torch.cuda.synchronize()
    s = Sampler(); s.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < seconds:
        fn(); n += 1
        torch.cuda.synchronize()
    dt = time.time() - t0
    s.stop = True; s.join()
    rows = s.rows[len(s.rows) // 3:]           # the settled part
    keys = sorted({k for r in rows for k in r})
    txt = []
    for k in keys:
        vals = [r[k] for r in rows if k in r and r[k] == r[k]]
        if not vals:
            continue
        scale, unit = (1e-6, "W") if k[1].startswith("power") else ((1e-6, "MHz") if k[1].startswith("freq") else (1e-3, "C"))
        txt.append(f"{k[1]} {sum(vals) / len(vals) * scale:.0f} {unit} (max {max(vals) * scale:.0f})")
    print(f"{tag}: {dt / n * 1e3:.3f} ms per call; " + "; ".join(txt))


cfg = bench.default_config()
torch.manual_seed(0)
G, D = C.build_models(cfg, dev)
tr = C.CutTrainer(G, D, cfg, B, 256, device=dev, amp=True)
g = torch.Generator().manual_seed(1)
ph = (torch.rand(B, 3, 256, 256, generator=g) * 2 - 1).to(dev)
mo = (torch.rand(B, 3, 256, 256, generator=g) * 2 - 1).to(dev)
step = [1]


def one_step():
    tr.train_step(step[0], ph, mo); step[0] += 1


time.sleep(2.0)
measure("idle", lambda: time.sleep(0.05), 2.0)
measure("train step", one_step)

for pname in ("prog_gfwd", "prog_g_features", "prog_g_features_bwd", "prog_g_compute"):      # the phases of the step, each replayed alone
    prog = getattr(tr, pname, None)
    if prog is not None:
        measure(f"{pname} alone", prog.run, 3.0)
from gan_variant_research_amd.runtime import Ctx, HipOps  # noqa: E402
ctx = Ctx(HipOps(dev), dev, BF16)
w = torch.randn(256, 256, 3, 3, device=dev) * 0.05
b = torch.zeros(256, device=dev)
layer = ConvLayer(ctx, w, b, torch.zeros_like(w), torch.zeros_like(b), 3, 1, 1)
x = ctx.view(2 * B, 64, 64, 256, 1); x.t.normal_().clamp_(min=0)
y = ctx.view(2 * B, 64, 64, 256, 0)
fwd = layer.fwd(x, y)
for o in layer.repack_ops(): o()      # after planning: the operand copies exist only then


def conv_loop():
    for _ in range(100):
        for o in fwd: o()


measure("conv 3x3 256->256 forward x100", conv_loop)
dy = ctx.view(2 * B, 64, 64, 256, 2); dy.t.normal_()
wg = layer.wgrad(x, dy, False, bias_too=False)


def wgrad_loop():
    for _ in range(100):
        for o in wg: o()


measure("weight gradient x100", wgrad_loop)
big = torch.empty(2 * B * 66 * 66 * 256, dtype=torch.bfloat16, device=dev).normal_()
big2 = torch.empty_like(big)


def copy_loop():
    for _ in range(100):
        big2.copy_(big)


measure("HBM copy 71 MB x100", copy_loop)
