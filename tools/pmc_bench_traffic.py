#!/usr/bin/env python3
"""HBM traffic of the dominant kernel ON THE BENCH WORKLOAD, from rocprofv3 --pmc passes of bench.py itself.

rocprofv3 --pmc aborts inside librocprofiler-sdk when the decode steps are replayed with hipGraphLaunch (profiles/README.md,
"profiler abort"), so the counter passes run the same kernels launched eagerly (PTTS_NO_GRAPH=1):

    for c in TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum; do
        PTTS_NO_GRAPH=1 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-dac
    done
    python3 tools/pmc_bench_traffic.py /tmp > profiles/pmc_attn_traffic.json          (summarise; the CSVs are ~100 MB each)

FETCH_SIZE is rebuilt from its gfx950 definition (rocprofv3 -L: BUBBLE*128 + (RDREQ - BUBBLE - RDREQ_32B)*64 + RDREQ_32B*32
bytes) and doubled per the gfx950 correction for wide coalesced reads (MI355X_MICROARCH.md, HBM: a 128-B request is
tallied at 64 B).  Self-attention launch j of the pass belongs to decode step j // layers: S = P + 1 + step keys,
algorithmic bytes 2 * B * S * D * sizeof(bf16) (KV_self(S) / L)."""
import csv
import glob
import json
import sys

base = sys.argv[1]
B, D, LAYERS, P = 32, 1024, 24, 16
SOURCE = sys.argv[2] if len(sys.argv) > 2 else "bench.py"
if len(sys.argv) > 4:
    LAYERS, P = int(sys.argv[3]), int(sys.argv[4])
vals = {}
for cname in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_BUBBLE_sum"):
    files = glob.glob(f"{base}/pmc_{cname}/**/*counter_collection.csv", recursive=True)
    if not files:
        continue
    rows = []
    with open(files[0]) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] == cname and "attn_decode_kernel" in r["Kernel_Name"] and "true>" in r["Kernel_Name"].replace(" ", ""):
                rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    rows.sort()
    vals[cname] = [v for _, v in rows]
rd = vals["TCC_EA0_RDREQ_sum"]
n = len(rd)
r32 = vals.get("TCC_EA0_RDREQ_32B_sum") or [0.0] * n
bub = vals.get("TCC_BUBBLE_sum") or [0.0] * n
n = min(n, len(r32), len(bub))
fetch = [bub[i] * 128 + (rd[i] - bub[i] - r32[i]) * 64 + r32[i] * 32 for i in range(n)]
algo = [2 * B * (P + 1 + i // LAYERS) * D * 2 for i in range(n)]
out = {"bf16": {
    "fetchx2_over_algo": round(2 * sum(fetch) / sum(algo), 4),
    "source": (f"rocprofv3 --pmc passes (one counter per pass) of `PTTS_NO_GRAPH=1 python3 tools/pmc_attn.py <L> {LAYERS} {P}`: the "
               f"bench's self-attention kernel at its batch and widths (batch 32, D 1024, bf16, {LAYERS} layers, S up to L - 1, same "
               "library) driven without PyTorch and with eager launches - the configurations under which rocprofv3 --pmc does not "
               "abort inside librocprofiler-sdk on this image (profiles/README.md); summarised by tools/pmc_bench_traffic.py")
              if SOURCE != "bench.py" else
              "rocprofv3 --pmc passes of `PTTS_NO_GRAPH=1 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-dac`",
    "launches": n, "passes_present": sorted(vals),
    "sum_fetch_x2_bytes": 2 * sum(fetch), "sum_algorithmic_bytes": sum(algo),
    "per_launch_avg_fetch_x2_bytes": round(2 * sum(fetch) / n), "per_launch_avg_algorithmic_bytes": round(sum(algo) / n),
    "rdreq_32B_total": sum(r32[:n]), "bubble_total": sum(bub[:n]),
    "ratio_first_tenth": round(2 * sum(fetch[: n // 10]) / sum(algo[: n // 10]), 4),
    "ratio_last_tenth": round(2 * sum(fetch[-(n // 10):]) / sum(algo[-(n // 10):]), 4),
}}
print(json.dumps(out, indent=1))
