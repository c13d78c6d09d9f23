#!/usr/bin/env python3
"""Per-launch durations of one codec decode from a rocprofv3 kernel trace (in launch order = layer order):
   rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/dac_profile.py 16 2571
   python3 tools/dac_layer_times.py out"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "conv" in r["Kernel_Name"] or "dac_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // 2                     # two decodes (warm-up + timed): keep the second
rows = rows[n:]
tot = 0
for i, r in enumerate(rows):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    tot += d
    name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0]
    print(f"{i:3d} {name:28s} grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>9s} {d:8.3f} ms")
print(f"total {tot:.1f} ms over {len(rows)} launches")
