"""Throughput of the device-side input pipeline (dataio.InputPipeline): 256x256 uint8 photos -> (B,3,256,256) fp32 through the CUT
train transform (crop + bicubic resize, flip, ColorJitter incl. hue, normalise).  usage: bench_input.py [B=16] [iters=200]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from gan_variant_research_amd import dataio

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
imgs = [torch.from_numpy(rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)).to(dev) for _ in range(B)]
tf = dataio.get_train_transforms(256, device=dev, max_batch=B, max_rows=256)
out = torch.empty(B, 3, 256, 256, device=dev)
for _ in range(10):
    tf(imgs)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    jobs = [dataio.train_job(256, 256, 256) for _ in range(B)]
    tf.pipe.run(imgs, jobs, out=out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
# device time alone (jobs drawn once)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(iters):
    tf.pipe.run(imgs, jobs, out=out)
ev1.record(); torch.cuda.synchronize()
dd = ev0.elapsed_time(ev1) / iters
print(f"input pipeline B={B}: {dt*1e3:.3f} ms/batch incl. host draws + table packing = {B/dt:.0f} images/s; device+launch {dd:.3f} ms/batch = {B/dd*1e3:.0f} images/s")
