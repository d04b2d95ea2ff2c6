"""Feasibility probe: does an HBM-bound InstanceNorm backward overlap with an MFMA-bound weight-gradient kernel on another stream?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gan_variant_research_amd import BF16, _lib
from gan_variant_research_amd.convplan import ConvLayer
from gan_variant_research_amd.runtime import Ctx, HipOps
dev = torch.device("cuda:0")
print("priority range", torch.cuda.Stream.priority_range())
HP = len(sys.argv) > 2 and sys.argv[2] == "hp"
side = torch.cuda.Stream(device=dev, priority=0)
main = torch.cuda.Stream(device=dev, priority=torch.cuda.Stream.priority_range()[1] if False else -1) if HP else torch.cuda.current_stream(dev)
torch.cuda.set_stream(main)
ops_a = HipOps(dev)
ops_b = HipOps(dev, stream=side.cuda_stream)
ca, cb = Ctx(ops_a, dev, BF16), Ctx(ops_b, dev, BF16)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
w = torch.randn(256, 256, 3, 3, device=dev) * 0.05
b = torch.zeros(256, device=dev)
layer = ConvLayer(cb, w, b, torch.zeros_like(w), torch.zeros_like(b), 3, 1, 1)      # its ops go to the side stream
x = cb.view(B, 64, 64, 256, 1); x.t.normal_()
dy = cb.view(B, 64, 64, 256, 2); dy.t.normal_()
wg = layer.wgrad(x, dy, False, bias_too=False)
# conv dgrad on the main stream (MFMA) and an IN backward (HBM)
layer2 = ConvLayer(ca, w, b, torch.zeros_like(w), torch.zeros_like(b), 3, 1, 1)
gp = ca.view(B, 64, 64, 256, 1)
dg = layer2.dgrad(dy, gp, padded_domain=True)
for o in layer2.repack_ops(): o()
raw = ca.view(B, 64, 64, 256, 0); raw.t.normal_()
stats = torch.zeros(B * 256 * 2, device=dev); stats[1::2] = 1.0
ws = torch.zeros(B * 96 * 256 * 2 + B * 256 * 2 + (B * 1024 + 32) * 256, device=dev)
dx = ca.view(B, 64, 64, 256, 2)
inb = [ops_a.in_bwd(raw, stats, _lib.ACT_RELU, gp, True, None, dx, ws)]

def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main)
    for _ in range(n): fn()
    e1.record(main); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

def seq():
    for o in inb: o()
    for o in dg: o()
    side.wait_stream(main)
    for o in wg: o()
    main.wait_stream(side)

def par():
    side.wait_stream(main)
    for o in wg: o()          # side stream: wgrad + reduce
    for o in inb: o()         # main stream: IN backward then dgrad
    for o in dg: o()
    main.wait_stream(side)

def only(ops_list, s=None):
    def f():
        if s is not None: side.wait_stream(main)
        for o in ops_list: o()
        if s is not None: main.wait_stream(side)
    return f
print(f"B={B}: in_bwd {timed(only(inb)):.1f} us, dgrad {timed(only(dg)):.1f} us, wgrad+reduce (side stream) {timed(only(wg, side)):.1f} us")
print(f"sequential {timed(seq):.1f} us   two streams {timed(par):.1f} us")
