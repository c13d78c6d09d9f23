#!/usr/bin/env python3
"""Tables from the per-kernel counter summaries tools/pmc_summary.py wrote (one JSON per rocprofv3 --pmc pass):
    python3 tools/pmc_report.py decoder DIR     -> the launches of a decode step at the bench shape
    python3 tools/pmc_report.py dac DIR         -> the codec conv kernels
    python3 tools/pmc_report.py prefill DIR     -> T5 encoder + prefill (same table; tools/gpu_pmc.sh OUT prefill)
Byte counts: fabric reads = 32 B x RDREQ_32B + 64 B x RDREQ_64B + 128 B x RDREQ_128B (the three size classes partition
TCC_EA0_RDREQ on gfx950: `other` = 0 in every row), writes = 64 B x WRREQ_64B + 32 B x the rest.  The per-launch durations are
those of the profiled run (counter collection serialises and slows the launches); shares and per-launch counts are what to read."""
import json
import sys

mode, d = sys.argv[1], sys.argv[2].rstrip("/") + "/"


def load(n):
    return {(r["kernel"], r["workgroups"], r["wg_size"]): r for r in json.load(open(d + n))}


if mode == "decoder":
    t1, t2, t3, tp, sq = (load(f"dec_{x}.json") for x in ("tcc1", "tcc2", "tcc3", "tcp", "sq1"))
    role = {("gemm_kernel<PBF16, 8, 1, false, 1, 2, true>", 192): "LN1 + QKV projection (folded)        6.29 MB",
            ("gemm_kernel<PBF16, 8, 1, false, 1, 1, false>", 256): "self out-proj + look-ahead cross-q   6.29 MB",
            ("gemm_kernel<PBF16, 8, 1, false, 1, 1, false>", 128): "cross out-proj                        2.10 MB",
            ("gemm_kernel<PBF16, 8, 1, false, 1, 2, true>", 256): "LN3 + fc1 + GELU (folded)            8.39 MB",
            ("gemm_kernel<PBF16, 16, 1, false, 1, 1, false>", 128): "fc2 + residual                        8.39 MB",
            ("gemm_kernel<PBF16, 8, 1, false, 1, 2, true>", 612): "final LN + 9 LM heads (folded)      20.05 MB",
            ("attn_decode_kernel<PBF16, PBF16, true>", 512): "self-attention, S ~ 1266            165.9 MB",
            ("attn_page_kernel<PBF16, PBF16>", 128): "cross-attention, T = 40               5.24 MB",
            # the final build of round 3: instances with a compile-time epilogue / exact-fit bursts / quarter-row workgroups
            ("gemm_kernel<PBF16, 8, 1, false, 1, 2, true, 3, 0, 0, 4, false>", 192): "LN1 + QKV projection (folded)        6.29 MB",
            ("gemm_kernel<PBF16, 8, 1, false, 1, 1, false, 1, 0, -1, 0, false>", 256): "self out-proj + look-ahead cross-q   6.29 MB",
            ("gemm_kernel<PBF16, 8, 1, false, 1, 1, false, 1, 0, 0, 4, true>", 256): "cross out-proj (quarter rows)         2.10 MB",
            ("gemm_kernel<PBF16, 8, 1, false, 1, 2, true, 2, 1, 0, 4, false>", 256): "LN3 + fc1 + GELU (folded)            8.39 MB",
            ("gemm_kernel<PBF16, 16, 1, false, 1, 1, false, 1, 0, 0, 8, true>", 256): "fc2 + residual (quarter rows)         8.39 MB",
            ("gemm_kernel<PBF16, 8, 1, false, 1, 2, true, 0, 0, 0, 4, false>", 612): "final LN + 9 LM heads (folded)      20.05 MB",
            ("gemm_kernel<PBF16, 8, 3, false, 1, 2, true, 0, 0, 0, 4, false>", 204): "final LN + 9 LM heads (folded, NB = 3) 20.05 MB",
            ("attn_decode_kernel<PBF16, PBF16, true, 0>", 512): "self-attention, S ~ 1266            165.9 MB",
            # round 5: QKV projection + self-attention in ONE launch (csrc/fused.hip): 192 projection + 512 attention workgroups
            ("fused_qkv_attn_kernel<PBF16, 4>", 704): "LN1 + QKV + self-attention, S ~ 1266 172.2 MB"}
    print("kernel (workgroups x threads)                                          launch = algorithmic weight/KV bytes      | fabric rd MB  wr MB | L2 hit % | "
          "L1->L2 read latency (cycles) | wave cycles: issuing % / parked on s_waitcnt+barrier % / issue-stalled % | MFMA busy cycles (summed over the SIMDs) per 100 SQ busy cycles")
    for k, r in sorted(t1.items(), key=lambda x: -x[1]["mean_ns"] * x[1]["launches"]):
        c, c2, c3, cp, cs = (t[k]["per_launch"] for t in (t1, t2, t3, tp, sq))
        n32, n64, n128, tot = (c["TCC_EA0_RDREQ_%s" % x] for x in ("32B_sum", "64B_sum", "128B_sum", "sum"))
        rd = (32 * n32 + 64 * n64 + 128 * n128) / 1e6
        wr = (c3["TCC_EA0_WRREQ_64B_sum"] * 64 + (c3["TCC_EA0_WRREQ_sum"] - c3["TCC_EA0_WRREQ_64B_sum"]) * 32) / 1e6
        hit = c2["TCC_HIT_sum"] / max(c2["TCC_HIT_sum"] + c2["TCC_MISS_sum"], 1)
        lat = cp["TCP_TCC_READ_REQ_LATENCY_sum"] / max(cp["TCP_TCC_READ_REQ_sum"], 1)
        wc = cs["SQ_WAVE_CYCLES"]
        name = k[0].split("(")[0]
        print(f"{name[:66]:66s} {k[1]:4d} x {k[2]:4d}  {role.get((name, k[1]), ''):44s} | {rd:8.2f} {wr:7.3f} | {100 * hit:6.1f}   | {lat:8.0f}"
              f"                     | {100 * cs['SQ_ACTIVE_INST_ANY'] / wc:5.1f} / {100 * cs['SQ_WAIT_ANY'] / wc:5.1f} / {100 * cs['SQ_WAIT_INST_ANY'] / wc:5.1f}"
              f"                        | {100 * cs['SQ_VALU_MFMA_BUSY_CYCLES'] / max(cs['SQ_BUSY_CYCLES'], 1):5.1f}   (other RDREQ {tot - n32 - n64 - n128:.0f})")
else:
    s1, s2, tc, gr = (load(f"dac_{x}.json") for x in ("sq1", "sq2", "tcc", "grbm"))
    print("kernel (workgroups)                                         launches  us/launch(profiled) | fabric rd GB  wr GB | wave cycles: issuing % / parked % / "
          "issue-stalled % / of which LDS % | MFMA pipe busy cycles per CU-busy cycle | MFMA:VALU instr | LDS conflict % of LDS-active | clock GHz")
    for k, r in sorted(s1.items(), key=lambda x: -x[1]["mean_ns"] * x[1]["launches"]):
        if r["mean_ns"] * r["launches"] < (2e4 if mode == "prefill" else 5e4) or k not in s2 or k not in tc or k not in gr:
            continue
        a, b, c, g = (t[k]["per_launch"] for t in (s1, s2, tc, gr))
        wc = a["SQ_WAVE_CYCLES"]
        rd = (c["TCC_EA0_RDREQ_128B_sum"] * 128 + (c["TCC_EA0_RDREQ_sum"] - c["TCC_EA0_RDREQ_128B_sum"]) * 64) / 1e9
        wr = (c["TCC_EA0_WRREQ_64B_sum"] * 64 + (c["TCC_EA0_WRREQ_sum"] - c["TCC_EA0_WRREQ_64B_sum"]) * 32) / 1e9
        ghz = g["GRBM_GUI_ACTIVE"] / 8 / gr[k]["mean_ns"]
        # SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs (4 per CU); SQ_BUSY_CU_CYCLES sums busy CUs
        pipe = a["SQ_VALU_MFMA_BUSY_CYCLES"] / max(4 * b["SQ_BUSY_CU_CYCLES"], 1)
        print(f"{k[0].split('(')[0][:52]:52s} {k[1]:6d} {r['launches']:6d} {r['mean_ns'] / 1e3:10.1f}          | {rd:8.3f} {wr:7.3f} | "
              f"{100 * a['SQ_ACTIVE_INST_ANY'] / wc:5.1f} / {100 * a['SQ_WAIT_ANY'] / wc:5.1f} / {100 * a['SQ_WAIT_INST_ANY'] / wc:5.1f} / {100 * a['SQ_WAIT_INST_LDS'] / wc:4.1f}"
              f"                 | {pipe:6.3f}                                  | {b['SQ_INSTS_MFMA'] / max(b['SQ_INSTS_VALU'], 1):5.2f}"
              f"           | {100 * a['SQ_LDS_BANK_CONFLICT'] / max(b['SQ_LDS_IDX_ACTIVE'], 1):5.2f}                      | {ghz:.2f}")
