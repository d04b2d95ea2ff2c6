"""Probe: does dlopen-ing libmi355x_gan.so before the HIP runtime is initialised break later launches?  usage: load_order.py MODE"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
mode = sys.argv[1]
LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gan-variant-research_amd", "libmi355x_gan.so")
if mode == "init_first":
    torch.cuda.init()
elif mode == "hipinit_first":
    hip = ctypes.CDLL("libamdhip64.so.7")
    print("hipInit ->", hip.hipInit(0))
elif mode == "devcount_first":
    print("device_count", torch.cuda.device_count())
lib = ctypes.CDLL(LIB)
print("loaded; now torch init")
x = torch.zeros(1 << 16, device="cuda:0")
torch.cuda.synchronize()
from gan_variant_research_amd.runtime import HipOps
from gan_variant_research_amd import _lib
ops = HipOps(torch.device("cuda:0"))
hip = ctypes.CDLL("libamdhip64.so.7")
print("hipGetLastError before first launch:", hip.hipGetLastError())
n = ctypes.c_int(0); print("hipGetDeviceCount rc", hip.hipGetDeviceCount(ctypes.byref(n)), n.value)
d = ctypes.c_int(-1); print("hipGetDevice rc", hip.hipGetDevice(ctypes.byref(d)), d.value)
import __graft_entry__ as g
try:
    g.smoke()
except Exception as e:
    print("FAILED:", e)
