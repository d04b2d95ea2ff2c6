"""Forward-only generator latency (inference.stylize), eager launches vs one hipGraph replay.  usage: bench_infer.py [size=256]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_variant_research_amd import BF16, cut as C, inference as I  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
C.set_seed(0)
for B in (1, 4, 16):
    x = (torch.rand(B, 3, S, S) * 2 - 1).to(dev)
    res = {}
    for graph in (False, True):
        G = C.ResNetGenerator(3, 3, 64, 9).to(dev).eval()
        G.compute_dtype, G.use_graph = BF16, graph
        for _ in range(3):
            y = I.stylize(G, x)
        torch.cuda.synchronize()
        n = 50
        t0 = time.perf_counter()
        for _ in range(n):
            y = I.stylize(G, x)
        torch.cuda.synchronize()
        res[graph] = (time.perf_counter() - t0) / n * 1e3
    print(f"B={B:2d} {S}x{S} bf16: eager {res[False]:.3f} ms, hipGraph {res[True]:.3f} ms  ({B / res[True] * 1e3:.0f} images/s)")
