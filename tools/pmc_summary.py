#!/usr/bin/env python3
"""Compact per-kernel summary of one `rocprofv3 --pmc ... --kernel-trace --output-format csv -d DIR` run (the raw CSVs of a
decode run are tens of MB: summarise on the GPU box, keep this).   python3 tools/pmc_summary.py DIR OUT.json [min_dispatch_id]

Launches are grouped by (kernel name, grid, workgroup size) - that separates the seven projection / attention launches of a
decoder layer; per group: launches, mean of every counter per launch, mean duration (ns) under the profiler."""
import collections
import csv
import glob
import json
import sys

base, out = sys.argv[1], sys.argv[2]
min_id = int(sys.argv[3]) if len(sys.argv) > 3 else 0
files = glob.glob(f"{base}/**/*counter_collection.csv", recursive=True)
agg = collections.OrderedDict()
for f in files:
    for r in csv.DictReader(open(f)):
        if int(r["Dispatch_Id"]) < min_id:
            continue
        name = r["Kernel_Name"]
        short = name.split("(anonymous namespace)::", 1)[-1][:70]     # the FIRST marker: an argument type may carry it too (FusedArgs)
        key = (short, int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1), int(r["Workgroup_Size"]))
        g = agg.setdefault(key, dict(ids=set(), counters=collections.defaultdict(float), ns=0.0, vgpr=r.get("VGPR_Count"),
                                     agpr=r.get("Accum_VGPR_Count"), sgpr=r.get("SGPR_Count"), lds=r.get("LDS_Block_Size")))
        if r["Dispatch_Id"] not in g["ids"]:
            g["ids"].add(r["Dispatch_Id"])
            g["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        g["counters"][r["Counter_Name"]] += float(r["Counter_Value"])
res = []
for (short, wgs, wg), g in agg.items():
    n = len(g["ids"])
    res.append(dict(kernel=short, workgroups=wgs, wg_size=wg, launches=n, mean_ns=round(g["ns"] / n, 1), vgpr=g["vgpr"], agpr=g["agpr"],
                    sgpr=g["sgpr"], lds=g["lds"], per_launch={k: round(v / n, 2) for k, v in g["counters"].items()}))
res.sort(key=lambda r: -r["mean_ns"] * r["launches"])
json.dump(res, open(out, "w"), indent=1)
print(f"{out}: {len(res)} kernel groups from {sum(r['launches'] for r in res)} launches")
