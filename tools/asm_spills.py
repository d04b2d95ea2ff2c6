"""Where a kernel's register spills sit: python tools/asm_spills.py file.s [name-substring]
For every kernel of a hipcc -S listing: MFMA count, scratch (spill) instructions before / inside / after the MFMA range, and inside
the range how many fall between two barriers that also enclose MFMAs (= in the tap loop) -- a spill in the epilogue is survivable,
one in the loop is not."""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else ""
start = 0
for i, l in enumerate(lines):
    m = re.match(r"\s*\.amdhsa_kernel\s+(\S+)", l)
    if not m:
        continue
    name, body = m.group(1), lines[start:i]
    start = i
    if want not in name:
        continue
    mf = [k for k, x in enumerate(body) if "v_mfma" in x]
    sc = [k for k, x in enumerate(body) if re.search(r"\bscratch_(load|store)", x)]
    bar = [k for k, x in enumerate(body) if "s_barrier" in x]
    if not mf:
        continue
    first, last = mf[0], mf[-1]
    inside = [k for k in sc if first <= k <= last]
    # segments between barriers that contain mfma
    loop_sc = 0
    edges = [0] + bar + [len(body)]
    for a, b in zip(edges[:-1], edges[1:]):
        nm = sum(1 for k in mf if a <= k < b)
        ns = sum(1 for k in sc if a <= k < b)
        if nm > 100:
            loop_sc += ns
    print(f"{name[:90]}\n   mfma {len(mf)}  scratch: before {sum(k < first for k in sc)}, in MFMA range {len(inside)} (in MFMA-dense barrier segments: {loop_sc}), after {sum(k > last for k in sc)}")
