#!/usr/bin/env python3
"""Report of the diagnostic build's in-kernel clock stamps (make -C parler_tts_amd/csrc stamps; PTTS_HIP_LIB=...stamps.so;
PTTS_STAMPS_OUT=file.npy python3 tools/pmc_attn.py ...): for every gemm / attention launch of the LAST decode step, where wave 0
of its workgroups spent its cycles, and how the launches sit against each other on the device's 100 MHz real-time counter.

    python3 tools/stamps_report.py stamps.npy [layers]

gemm phases (cycles): entry -> loads issued (kernel arguments, addresses) -> all loads of the first burst landed -> MFMA loop ->
K-split hand-over + barrier -> epilogue stores issued.   attention: entry -> first K burst issued -> scores -> softmax -> P.V -> stored."""
import sys

import numpy as np

a = np.load(sys.argv[1]).astype(np.int64)      # [slots][1024][8]
L = int(sys.argv[2]) if len(sys.argv) > 2 else 24
per_layer = (a.shape[0] - 1) // L
names7 = ["qkv", "self_attn", "out+look", "cross_attn", "out2", "fc1", "fc2"]
names = names7 if per_layer == 7 else [f"k{i}" for i in range(per_layer)]
rows = []
for s in range(a.shape[0]):
    w = a[s]
    live = w[:, 0] != 0
    if not live.any():
        continue
    w = w[live]
    t_first, t_last_start = w[:, 0].min(), w[:, 0].max()
    t_end = w[:, 7].max()
    ph = np.diff(w[:, 1:7], axis=1)                 # 5 phase lengths per workgroup (cycles)
    rows.append(dict(slot=s, wgs=int(live.sum()), start=t_first, ramp=(t_last_start - t_first) * 10, span=(t_end - t_first) * 10,
                     end=t_end, ph_med=np.median(ph, axis=0), ph_max=ph.max(axis=0), cyc=np.median(w[:, 6] - w[:, 1]),
                     cyc_max=(w[:, 6] - w[:, 1]).max()))
print("slot kernel        wgs  gap_ns  ramp_ns  span_ns | median cycles per phase (wave 0)              | wg cycles med / max  -> clock GHz")
prev_end = None
acc = {}
for r in rows:
    nm = "heads" if r["slot"] == a.shape[0] - 1 and per_layer * L + 1 == a.shape[0] else names[r["slot"] % per_layer]
    gap = (r["start"] - prev_end) * 10 if prev_end is not None else 0
    prev_end = r["end"]
    ghz = r["cyc_max"] / max(r["span"], 1)
    if r["slot"] < 2 * per_layer or r["slot"] >= a.shape[0] - per_layer - 1:
        print(f"{r['slot']:4d} {nm:12s} {r['wgs']:4d} {gap:7d} {r['ramp']:8d} {r['span']:8d} | " +
              " ".join(f"{int(x):7d}" for x in r["ph_med"]) + f" | {int(r['cyc']):7d} / {int(r['cyc_max']):7d}  {ghz:5.2f}")
    d = acc.setdefault(nm, dict(n=0, gap=0, span=0, ph=np.zeros(5), ramp=0))
    d["n"] += 1; d["gap"] += gap; d["span"] += r["span"]; d["ph"] += r["ph_med"]; d["ramp"] += r["ramp"]
print("\naverages over the step (ns; phases in cycles of wave 0, median workgroup):")
tot = 0
for nm, d in acc.items():
    n = d["n"]
    tot += d["gap"] + d["span"]
    print(f"  {nm:12s} x{n:3d}: gap before {d['gap'] / n:7.0f}  first->last start {d['ramp'] / n:6.0f}  first start -> last end {d['span'] / n:7.0f} | " +
          " ".join(f"{x / n:7.0f}" for x in d["ph"]))
print(f"sum of gaps + spans over the recorded launches: {tot / 1e3:.1f} us")
