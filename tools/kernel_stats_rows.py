#!/usr/bin/env python3
"""Rows of a rocprofv3 `--kernel-trace --stats` summary whose kernel name matches a regular expression:
   python3 tools/kernel_stats_rows.py <dir or kernel_stats.csv> 'conv_kernel_h3r|conv_kernel_s1'
prints: calls, total ms, average us, name (kernel names contain commas: cut(1) does not survive them)."""
import csv
import glob
import os
import re
import sys

path = sys.argv[1]
if os.path.isdir(path):
    path = glob.glob(path + "/**/*kernel_stats.csv", recursive=True)[0]
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
tot = 0.0
for r in csv.DictReader(open(path)):
    if pat.search(r["Name"]):
        name = r["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
        tot += int(r["TotalDurationNs"]) / 1e6
        print(f'{int(r["Calls"]):6d} calls {int(r["TotalDurationNs"]) / 1e6:10.3f} ms total {float(r["AverageNs"]) / 1e3:10.1f} us avg  {name}')
print(f"{tot:.3f} ms in the matching kernels")
