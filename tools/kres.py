#!/usr/bin/env python3
"""Kernel resource table from `hipcc -Rpass-analysis=kernel-resource-usage` output on stdin (dev tool)."""
import re, sys
cur = {}
rows = []
for line in sys.stdin:
    m = re.search(r"remark: +(.*?)(?: \[-Rpass.*)?$", line.strip())
    if not m: continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}; rows.append(cur)
    elif ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for r in rows:
    if flt in r["name"]:
        n = r["name"].replace("_ZN12_GLOBAL__N_1", "").replace("EEvNS_8GemmArgsE", "")
        print(f'{n:46s} vgpr {r.get("VGPRs","?"):>4s} agpr {r.get("AGPRs","?"):>3s} spill {r.get("VGPRs Spill","?"):>3s} scratch {r.get("ScratchSize [bytes/lane]","?"):>4s} occ {r.get("Occupancy [waves/SIMD]","?")} lds {r.get("LDS Size [bytes/block]","?")}')
