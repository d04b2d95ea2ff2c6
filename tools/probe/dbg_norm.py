"""Debug aid: gan_in_stats / gan_in_apply on one shape against the emulator, piece by piece."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gan_variant_research_amd import F32, BF16
from gan_variant_research_amd.runtime import Ctx, HipOps
from tests.emulator import EmuOps
dt = F32 if len(sys.argv) < 2 or sys.argv[1] == "f32" else BF16
B, H, W, C, halo = 2, 16, 16, 64, 1
dev = torch.device("cuda:0")
cg, cc = Ctx(HipOps(dev), dev, dt), Ctx(EmuOps(), "cpu", dt)
torch.manual_seed(0)
xc = cc.view(B, H, W, C, 0); xc.t.copy_(torch.randn(xc.t.shape).to(xc.t.dtype))
xg = cg.view(B, H, W, C, 0); xg.t.copy_(xc.t)
sc, sg = torch.zeros(B * C * 2), torch.zeros(B * C * 2, device=dev)
wc, wg = torch.zeros(B * 96 * C * 2 + B * C * 2), torch.zeros(B * 96 * C * 2 + B * C * 2, device=dev)
cc.ops.in_stats(xc, 1e-5, sc, wc)(); cg.ops.in_stats(xg, 1e-5, sg, wg)(); torch.cuda.synchronize()
print("stats max err", (sg.cpu() - sc).abs().max().item())
for mode in (2, 1):
    yc, yg = cc.view(B, H, W, C, halo), cg.view(B, H, W, C, halo)
    cc.ops.in_apply(xc, sc, 1, None, yc, mode)(); cg.ops.in_apply(xg, sc.to(dev), 1, None, yg, mode)(); torch.cuda.synchronize()
    d = (yg.t.float().cpu() - yc.t.float()).abs().view(B, H + 2 * halo, W + 2 * halo, C)
    print("mode", mode, "apply max err", d.max().item(), "bad pixels per row:", (d.amax(3) > 1e-3).sum(2)[0].tolist())
