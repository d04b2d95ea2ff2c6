"""Program-level timeline of one CUT train step WITHOUT a profiler: HIP events recorded on all three streams around every program of the
step (the host stays a step ahead, unlike under rocprofv3, whose per-dispatch cost serialises the host's enqueue order into the trace).

    python tools/prog_timeline.py [batch] [size]

Per program: when each stream passed the markers before / after its launches were queued, relative to the step's first marker."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gan_variant_research_amd import cut as C  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
cfg = bench.default_config()
torch.manual_seed(0)
G, D = C.build_models(cfg, dev)
tr = C.CutTrainer(G, D, cfg, B, S, device=dev, amp=True)
g = torch.Generator().manual_seed(1)
ph = (torch.rand(B, 3, S, S, generator=g) * 2 - 1).to(dev)
mo = (torch.rand(B, 3, S, S, generator=g) * 2 - 1).to(dev)
for s in range(1, 6):
    tr.train_step(s, ph, mo, sync="lag")
streams = {"main": tr.ops._ts(), "disc": tr.ops.fork()._ts(), "side": tr.ops.side()._ts()}
names = [n for n in ("prog_gfwd", "prog_d_compute", "prog_d_update", "prog_g_adversarial", "prog_g_features", "prog_g_features_bwd", "prog_g_compute",
                     "prog_g_identity", "prog_g_update") if getattr(tr, n, None) is not None]
marks = []   # (label, {stream: event})


def mark(label):
    evs = {}
    for k, st in streams.items():
        e = torch.cuda.Event(enable_timing=True)
        e.record(st)
        evs[k] = e
    marks.append((label, evs))


class Wrapped:
    def __init__(self, name, prog):
        self.name, self.prog, self.ops = name, prog, prog.ops

    def run(self):
        mark(self.name + " >")
        self.prog.run()
        mark(self.name + " <")

    def __len__(self):
        return len(self.prog)


NSTEP = 4
saved = {n: getattr(tr, n) for n in names}
torch.cuda.synchronize()
for n in names:
    setattr(tr, n, Wrapped(n, saved[n]))
steps = []
for s in range(6, 6 + NSTEP):
    marks = []
    mark("step >")
    tr.train_step(s, ph, mo, sync="lag")
    mark("step <")
    steps.append(marks)
torch.cuda.synchronize()
for n in names:
    setattr(tr, n, saved[n])
m = steps[-2]       # a steady-state step (not the first wrapped one, not the last)
t0 = m[0][1]["main"]
print(f"B={B} S={S}: marker times in ms since the step's first marker on the main stream (a marker is passed when everything queued before it on that stream is done)")
print(f"{'marker':28s} " + " ".join(f"{k:>9s}" for k in streams))
for label, evs in m:
    print(f"{label:28s} " + " ".join(f"{t0.elapsed_time(evs[k]):9.3f}" for k in streams))
nxt = steps[-1][0][1]["main"]
print(f"next step's first marker: {t0.elapsed_time(nxt):.3f} ms")
