"""How long does the HOST take to queue one training step (no read-back), against the GPU's time for it?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from gan_variant_research_amd import cut as C
dev = torch.device("cuda:0")
cfg = bench.default_config()
C.set_seed(42)
G, D = C.build_models(cfg, dev)
tr = C.CutTrainer(G, D, cfg, 16, 256, device=dev, amp=True)
g = torch.Generator().manual_seed(1)
ph = (torch.rand(16, 3, 256, 256, generator=g) * 2 - 1).to(dev)
mo = (torch.rand(16, 3, 256, 256, generator=g) * 2 - 1).to(dev)
for s in range(1, 6):
    tr.train_step(s, ph, mo)
torch.cuda.synchronize()
for s in range(17, 20):       # one step at a time from an idle GPU: pure host cost of queueing it
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train_step(s, ph, mo, sync=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"single step: host {1e3 * (t1 - t0):.2f} ms, until the GPU is done {1e3 * (time.perf_counter() - t0):.2f} ms")
n = 10
t0 = time.perf_counter()
for s in range(17, 17 + n):
    tr.train_step(s, ph, mo, sync=False)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host queues a step in {(t1 - t0) / n * 1e3:.2f} ms; the GPU finishes {n} steps in {(t2 - t0) / n * 1e3:.2f} ms per step")
