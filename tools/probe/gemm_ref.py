"""Reference point for DESIGN 3.7: what does the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) deliver on this board, on the same kind of
data, under the same 1400 W limit?  bf16 operands, fp32 accumulation, random N(0,1) and all-zero inputs; board power and firmware clock
from hwmon while each shape runs for ~2 s.  Not part of the product: a yardstick for the range-patch convolution's 1.30-1.35 PFLOP/s.

    python tools/probe/gemm_ref.py
"""
import glob
import threading
import time

import torch

dev = torch.device("cuda:0")


def read(p):
    try:
        with open(p) as f:
            return float(f.read().strip())
    except Exception:
        return float("nan")


nodes = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")
before = {h: read(h + "/power1_input") for h in nodes}
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
t0 = time.time()
while time.time() - t0 < 1.0:
    (a @ a)
torch.cuda.synchronize()
HW = max(nodes, key=lambda h: read(h + "/power1_input") - before[h])
print("sensor", HW, "cap", read(HW + "/power1_cap") / 1e6, "W")


def run(tag, A, Bm, seconds=2.0):
    p, f, stop = [], [], [False]

    def sample():
        while not stop[0]:
            p.append(read(HW + "/power1_input") * 1e-6)
            f.append(read(HW + "/freq1_input") * 1e-6)
            time.sleep(0.01)
    th = threading.Thread(target=sample, daemon=True)
    th.start()
    out = torch.empty(A.shape[0], Bm.shape[1], device=dev, dtype=torch.bfloat16)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time()
    while time.time() - t0 < seconds:
        e0.record()
        for _ in range(20):
            torch.matmul(A, Bm, out=out)
        e1.record()
        torch.cuda.synchronize()
    stop[0] = True
    th.join()
    ms = e0.elapsed_time(e1) / 20
    flop = 2.0 * A.shape[0] * A.shape[1] * Bm.shape[1]
    pw, fq = p[len(p) // 2:], f[len(f) // 2:]
    print(f"{tag:58s} {ms * 1e3:9.1f} us  {flop / ms / 1e9:7.1f} TFLOP/s  {sum(pw) / len(pw):5.0f} W  {sum(fq) / len(fq):5.0f} MHz (firmware)")


for M, N, K, note in ((8192, 8192, 8192, "square"), (131072, 256, 2304, "the residual convolution as a GEMM (32 images)"),
                      (16384, 4096, 4096, ""), (4096, 4096, 16384, "")):
    for data in ("random N(0,1)", "zeros"):
        A = torch.randn(M, K, device=dev, dtype=torch.bfloat16) if data.startswith("random") else torch.zeros(M, K, device=dev, dtype=torch.bfloat16)
        Bm = torch.randn(K, N, device=dev, dtype=torch.bfloat16) * 0.05 if data.startswith("random") else torch.zeros(K, N, device=dev, dtype=torch.bfloat16)
        run(f"matmul {M} x {K} x {N} bf16, {data} {note}", A, Bm)
        del A, Bm
