#!/usr/bin/env python3
"""Per-kernel table of the LAST T5 + prefill pass in a rocprofv3 kernel trace of tools/prefill_profile.py:
    python3 tools/prefill_trace_report.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
s = [i for i, r in enumerate(rows) if "t5_embed_kernel" in r["Kernel_Name"]][-1]
e = next(i for i in range(s, len(rows)) if "embed_codebooks" in rows[i]["Kernel_Name"])
seg = rows[s:e]
span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
agg = collections.OrderedDict()
busy = 0.0
for r in seg:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    busy += d
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
    k = (name, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]))
    a = agg.setdefault(k, [0, 0.0, 1e9])
    a[0] += 1
    a[1] += d
    a[2] = min(a[2], d)
print(f"last T5 + prefill pass: {len(seg)} launches, span {span:.1f} us, kernels busy {busy:.1f} us, gaps {span - busy:.1f} us")
for k, (n, t, mn) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:4d} x {t / n:7.1f} us (min {mn:6.1f}) = {t:8.1f} us  grid {k[1]:4d} x {k[2]:3d}  {k[0]}")
