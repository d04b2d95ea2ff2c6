import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_variant_research_amd import F32, BF16
from gan_variant_research_amd.runtime import Ctx, HipOps, ConvCall
dev = torch.device("cuda:0")
for dtype in (F32, BF16):
    ctx = Ctx(HipOps(dev), dev, dtype)
    Cin = 32 if dtype == F32 else 64
    B, H, W = 1, 4, 8   # 32 pixels
    x = ctx.view(B, H, W, Cin, 0)
    xm = torch.zeros(32, Cin)
    for m in range(32):
        xm[m, m % Cin] = 1.0           # pixel m has a one at channel m
    x.t.copy_(xm.reshape(-1).to(x.t.dtype))
    Nw = 16
    w = torch.zeros(Nw, 1, Cin)
    for n in range(Nw):
        for c in range(Cin):
            w[n, 0, c] = n * 100 + c
    wd = w.reshape(-1).to(ctx.tdtype).to(dev)
    y = ctx.view(B, H, W, 16, 0)
    tap = ctx.i32([0])
    ctx.ops.conv_igemm(ConvCall(B, H, W, Cin, 1, Nw, 16, x, 0, 0, 1, 1, tap, wd, None, y, 0, 0, 1, 1))()
    torch.cuda.synchronize()
    out = y.nhwc().float().reshape(32, 16).cpu()
    print("dtype", dtype, "expect y[m][n] = n*100+m")
    print(out[:10, :6])
