"""Board power and firmware-reported shader clock of every distinct launch of the CUT step, each replayed alone in a loop (~1 s):
which kernels pull the clock down?  (GAN_SINGLE_STREAM=1: every launch on one stream.)

    python tools/op_power.py [batch] [seconds-per-op] [substring of the launches to keep]
"""
import collections
import glob
import os
import sys
import threading
import time

os.environ.setdefault("GAN_SINGLE_STREAM", "1")
import torch

if os.environ.get("GAN_W0"):       # diagnostic: every range-patch weight fetch reads fragment block 0 (L1-resident; results wrong)
    _stamp_buf = torch.zeros(256 * 32, dtype=torch.int64, device="cuda:0")
    os.environ["GAN_PATCH_STAMPS"] = str(_stamp_buf.data_ptr() | 2)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gan_variant_research_amd import cut as C  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
SEC = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
ONLY = sys.argv[3] if len(sys.argv) > 3 else ""


def read(p):
    try:
        with open(p) as f:
            return float(f.read().strip())
    except Exception:
        return float("nan")


def own_hwmon():
    nodes = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")
    before = {h: read(h + "/power1_input") for h in nodes}
    a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    t0 = time.time()
    while time.time() - t0 < 1.0:
        (a @ a)
    torch.cuda.synchronize()
    return max(nodes, key=lambda h: read(h + "/power1_input") - before[h])


HW = own_hwmon()
print("sensor", HW, "cap", read(HW + "/power1_cap") / 1e6, "W")


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.stop, self.p, self.f = False, [], []

    def run(self):
        while not self.stop:
            self.p.append(read(HW + "/power1_input") * 1e-6)
            self.f.append(read(HW + "/freq1_input") * 1e-6)
            time.sleep(0.01)


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


def key_of(op):
    c = getattr(op, "conv", None) or getattr(op, "wgrad", None)
    if c is None:
        return getattr(op, "__name__", "op")
    if hasattr(op, "wgrad"):
        return f"wgrad{'.patch' if c.variant else ''} B{c.B} {c.Ho}x{c.Wo} Cx{c.Cx} N{c.N} taps{c.ntaps} s{c.x_sy}"
    return f"{'conv.patch' if c.w_frag else 'conv'} B{c.B} {c.Ho}x{c.Wo} Cin{c.Cin} Nst{c.Nst} taps{c.ntaps} s{c.in_sy}{' chain' if getattr(c, 'stats_mode', 0) else ''}"


groups = collections.OrderedDict()
for pname in ["prog_gfwd", "prog_d_compute", "prog_g_features", "prog_g_adversarial", "prog_g_features_bwd", "prog_g_compute"]:
    prog = getattr(tr, pname, None)
    for op in (prog.ops if prog is not None else []):
        k = key_of(op)
        if k in ("op", "<lambda>") or k.startswith("_") or ONLY not in k:
            continue
        groups.setdefault(k, []).append(op)

rows = []
for k, ops in groups.items():
    op = ops[0]
    try:
        for _ in range(3):
            op()
        torch.cuda.synchronize()
    except Exception as e:   # callbacks that need the step's context
        print("skip", k, type(e).__name__)
        continue
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s = Sampler(); s.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < SEC:
        e0.record()
        for _ in range(50):
            op()
        e1.record(); torch.cuda.synchronize(); n += 50
    us = e0.elapsed_time(e1) / 50 * 1e3
    s.stop = True; s.join()
    p, f = s.p[len(s.p) // 2:], s.f[len(s.f) // 2:]
    rows.append((len(ops), us, sum(p) / len(p), sum(f) / len(f), k))
    time.sleep(0.3)
print(f"{'calls':>5} {'us':>8} {'W':>6} {'MHz':>6}  launch (alone, back to back)")
for n, us, p, f, k in sorted(rows, key=lambda r: -r[0] * r[1]):
    print(f"{n:5d} {us:8.1f} {p:6.0f} {f:6.0f}  {k}")
