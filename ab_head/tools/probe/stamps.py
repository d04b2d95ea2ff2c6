import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dev = torch.device("cuda:0")
buf = torch.zeros(256 * 32, dtype=torch.int64, device=dev)
os.environ["GAN_PATCH_STAMPS"] = str(buf.data_ptr() + (int(sys.argv[1]) if len(sys.argv) > 1 else 0))
from gan_variant_research_amd import BF16
from gan_variant_research_amd.convplan import ConvLayer
from gan_variant_research_amd.runtime import Ctx, HipOps
ctx = Ctx(HipOps(dev), dev, BF16)
B = 16
w = torch.randn(256, 256, 3, 3, device=dev) * 0.05
b = torch.zeros(256, device=dev)
layer = ConvLayer(ctx, w, b, torch.zeros_like(w), torch.zeros_like(b), 3, 1, 1)
x = ctx.view(B, 64, 64, 256, 1); x.t.normal_()
y = ctx.view(B, 64, 64, 256, 0)
ops = layer.fwd(x, y)
for o in layer.repack_ops(): o()
for _ in range(5):
    for o in ops: o()
torch.cuda.synchronize()
st = buf.view(256, 32).cpu()
import numpy as np
s = st.numpy().astype(np.int64)
d = np.diff(s[:, :12], axis=1)   # stamps: start, after prologue, 4 slabs, epilogue, 4 slabs, epilogue
names = ["prologue", "slab0", "slab1", "slab2", "slab3", "epilogue", "slab0'", "slab1'", "slab2'", "slab3'", "epilogue'"]
print("s_memtime ticks (100 MHz constant clock?) median over 256 blocks; total", np.median(s[:, 11] - s[:, 0]))
for n, col in zip(names, d.T):
    print(f"{n:10s} median {np.median(col):9.0f}  min {col.min():9.0f}  max {col.max():9.0f}")
print("block start spread:", s[:, 0].max() - s[:, 0].min(), " end spread:", s[:, 11].max() - s[:, 11].min())
