"""Per-launch cost table of one CUT train step: replays every op of the step's programs on its own between HIP events.

    python tools/step_ops.py [batch] [size] [--fp32]

Convolution launches are priced in algorithmic FLOPs (2*B*Ho*Wo*Nst*Cin*ntaps); everything else is listed by time only.
The sum differs from the step time: ops run back to back here with their operands cache-resident (a replayed HBM-bound launch reads
optimistic: judge those by the in-step rocprofv3 profile), and the step overlaps three streams."""
import collections
import os
import sys

os.environ.setdefault("GAN_SINGLE_STREAM", "1")   # every launch on the one stream the HIP events below are recorded on

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gan_variant_research_amd import cut as C  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
amp = "--fp32" not in sys.argv
dev = torch.device("cuda:0")
cfg = bench.default_config()
torch.manual_seed(0)
G, D = C.build_models(cfg, dev)
tr = C.CutTrainer(G, D, cfg, B, S, device=dev, amp=amp)
g = torch.Generator().manual_seed(1)
ph = (torch.rand(B, 3, S, S, generator=g) * 2 - 1).to(dev)
mo = (torch.rand(B, 3, S, S, generator=g) * 2 - 1).to(dev)
for s in range(1, 4):
    tr.train_step(s, ph, mo)
torch.cuda.synchronize()

rows = collections.OrderedDict()
tot = 0.0
for pname in ["prog_gfwd", "prog_d_compute", "prog_d_update", "prog_g_features", "prog_g_adversarial", "prog_g_features_bwd", "prog_g_compute", "prog_g_identity", "prog_g_update"]:
    for op in (getattr(tr, pname).ops if getattr(tr, pname) is not None else []):
        for _ in range(2):
            op()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            op()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 5 * 1e3
        c = getattr(op, "conv", None) or getattr(op, "wgrad", None)
        if c is not None:
            kind = "wgrad" if hasattr(op, "wgrad") else ("conv.patch" if c.w_frag else "conv")
            if kind == "wgrad":
                key = f"{kind}{'.patch' if c.variant else ''} B{c.B} {c.Ho}x{c.Wo} Cx{c.Cx} N{c.N} taps{c.ntaps} s{c.x_sy} split{c.nsplit}"
                fl = 2.0 * c.B * c.Ho * c.Wo * c.N * c.Cx * c.ntaps
            else:
                key = f"{kind} B{c.B} {c.Ho}x{c.Wo} Cin{c.Cin} Nst{c.Nst} taps{c.ntaps} s{c.in_sy} act{c.act}{' mask' if c.mask is not None else ''}"
                fl = c.alg_flops()
        else:
            key, fl = getattr(op, "__name__", "op"), 0.0
        r = rows.setdefault((pname, key), [0, 0.0, 0.0])
        r[0] += 1; r[1] += us; r[2] += fl
        tot += us
print(f"sum of op times {tot/1e3:.2f} ms  (B={B}, S={S}, {'bf16' if amp else 'fp32'})")
agg = collections.OrderedDict()
for (pn, key), (n, us, fl) in rows.items():
    a = agg.setdefault(key, [0, 0.0, 0.0]); a[0] += n; a[1] += us; a[2] += fl
print(f"{'ms/step':>8} {'calls':>5} {'avg us':>8} {'TF/s':>7}  op")
for key, (n, us, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{us/1e3:8.3f} {n:5d} {us/n:8.1f} {fl/us/1e6 if fl else 0:7.1f}  {key}")
