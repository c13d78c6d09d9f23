"""Per-kernel summary (count, average, total) from a rocprofv3 `*_results.db` (the sqlite output of ROCm 7.x), grouped by
kernel name and launch geometry.  usage: python tools/rocpd_kernels.py gpurun_out/prof/x_results.db [top_n]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute(
    f"select s.kernel_name, d.grid_size_x/d.workgroup_size_x, d.grid_size_y, d.workgroup_size_x, d.group_segment_size, count(*), "
    f"avg(d.end-d.start), sum(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by 1,2,3,4,5 order by 8 desc").fetchall()
total = sum(r[7] for r in rows)
print(f"{'kernel':80s} {'grid':>9s} {'wg':>5s} {'lds':>7s} {'calls':>7s} {'avg us':>8s} {'total ms':>9s} {'%':>5s}")
for name, gx, gy, wg, lds, n, avg, tot in rows[:top]:
    name = re.sub(r"^void |\(.*$|\.kd$", "", name).replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")[:80]
    print(f"{name:80s} {gx:5d}x{gy:<3d} {wg:5d} {lds:7d} {n:7d} {avg/1e3:8.2f} {tot/1e6:9.2f} {100*tot/total:5.1f}")
print(f"total kernel time {total/1e6:.2f} ms over {sum(r[5] for r in rows)} dispatches")
