#!/usr/bin/env python3
"""Static instruction mix of the kernels in a gfx950 .s file (hipcc -S --cuda-device-only):
   python tools/isa_count.py file.s [name-substring]
Prints per kernel: VGPRs, spills, LDS bytes, and counts of MFMA / other VALU / SALU / LDS / global instructions."""
import re, sys, collections
path, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
cur, counts, meta = None, {}, {}
for line in open(path):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur = m.group(1); counts[cur] = collections.Counter(); continue
    if cur is None: continue
    s = line.strip()
    if s.startswith(".end_amdhsa_kernel") or s.startswith(".Lfunc_end"):
        pass
    m = re.match(r"^;\s*(NumVgprs|NumAgprs|ScratchSize|LDSByteSize|Occupancy|NumSgprs|TotalNumVgprs):\s*(\d+)", s)
    if m:
        meta.setdefault(cur, {})[m.group(1)] = int(m.group(2)); continue
    if not s or s[0] in ".;" or s.endswith(":"): continue
    op = s.split()[0]
    if op.startswith("v_mfma"): counts[cur]["mfma"] += 1
    elif op.startswith("v_"): counts[cur]["valu"] += 1
    elif op.startswith("s_"): counts[cur]["salu"] += 1
    elif op.startswith("ds_"): counts[cur]["lds"] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): counts[cur]["vmem"] += 1
    else: counts[cur]["other"] += 1
for k, c in counts.items():
    if pat and pat not in k: continue
    if sum(c.values()) == 0: continue
    print(k[:90]); print("   ", dict(c), meta.get(k, {}))
