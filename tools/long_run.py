"""Sanity run: N CUT steps in bf16 on synthetic data, printing the losses every K steps (finite, no drift to NaN; R1 steps included).
usage: long_run.py [steps=300] [every=50]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import default_config  # noqa: E402
from gan_variant_research_amd import cut as C  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
every = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda:0")
cfg = default_config()
C.set_seed(42)
gen, disc = C.build_models(cfg, dev)
tr = C.CutTrainer(gen, disc, cfg, 16, 256, device=dev, amp=True)
g = torch.Generator().manual_seed(1)
photos = (torch.rand(16, 3, 256, 256, generator=g) * 2 - 1).to(dev)
monets = (torch.rand(16, 3, 256, 256, generator=g) * 2 - 1).to(dev)
for step in range(steps):
    out = tr.train_step(step, photos, monets, sync=(step % every == 0 or step == steps - 1) or "lag")
    if out is not None and (step % every == 0 or step == steps - 1):
        print(step, {k: round(v, 4) for k, v in out.items()}, flush=True)
tr.flush_losses()
print("finite over", steps, "steps; max |G(x)| =", float(tr.generated().abs().max()))
