#!/usr/bin/env python3
"""Both rooflines of every launch of one codec decode: from the per-launch times of tools/dac_layer_times.py
(one chunk of 16 utterances x 2571 frames of the 44 kHz DAC decoder: strides 8, 8, 4, 2; channels 1536 -> 768 -> 384 -> 192 -> 96)

    python3 tools/dac_two_rooflines.py gpurun_out/X/dac_layer_times.txt > profiles/r05_dac_two_rooflines.txt
    (round 6: ... > profiles/r06_dac_two_rooflines.txt; the family table then carries round 5's column beside the new one)

per launch: its matrix-core time (3 x the fp32-equivalent flops - every product runs as hi.hi + hi.lo + lo.hi on
v_mfma_f32_16x16x32_f16 - divided by the dense fp16 peak, 2.5 PFLOP/s at 2.4 GHz, and by the rate at the ~2.0 GHz these kernels
clock at under load: profiles/r04_pmc_dac.txt), its HBM time (fp32 activation bytes in + out, the residual operand where the
kernel re-reads it, divided by 6.3 TB/s - the part's measured streaming rate, MI355X_MICROARCH.md), the measured time, and the
ratio measured / max(the two).  Reference of the arithmetic: dac_wrapper/modeling_dac.py:98-134 -> the DAC decoder stack."""
import re
import sys

rows = []
for ln in open(sys.argv[1]):
    m = re.match(r"\s*(\d+)\s+(.*?)\s+grid\s+(\d+)\s+([\d.]+) ms", ln)
    if m:
        rows.append((int(m.group(1)), m.group(2).strip(), float(m.group(4))))
U, F = 16, 2571
PEAK, CLK, HBM = 2.5e15, 2.0 / 2.4, 6.3e12
# (name, kernel family, flops (fp32-equivalent), HBM bytes) in launch order, per chunk of U utterances
spec = []
T = U * F
spec.append(("conv0 1024->1536 k7", "ring 7-tap", 2 * 7 * 1024 * 1536 * T, 4 * T * (1024 + 1536)))
C = 1536
for bi, s in enumerate((8, 8, 4, 2)):
    Co = C // 2
    To = T * s
    spec.append((f"block{bi + 1} convT {C}->{Co} stride {s} (k={2 * s})", "ring 2-tap", 2 * 2 * C * Co * To, 4 * (T * C + To * Co)))
    for u in range(3):
        if Co >= 384:
            spec.append((f"block{bi + 1} unit{u + 1} conv7 {Co} dil {3 ** u}", "ring 7-tap", 2 * 7 * Co * Co * To, 4 * To * Co * 2))
            spec.append((f"block{bi + 1} unit{u + 1} conv1 {Co} + residual", "1x1 (s1)", 2 * Co * Co * To, 4 * To * Co * 3))
        else:
            spec.append((f"block{bi + 1} unit{u + 1} conv7 + conv1 {Co} fused + residual", "fused unit", 2 * 8 * Co * Co * To, 4 * To * Co * 2))
    C, T = Co, To
spec.append(("last snake + conv 96->1 k7 + tanh", "last", 2 * 7 * 96 * T, 4 * T * (96 + 1)))
# the conv launches of ONE chunk: skip the small set-up / gather launches in front (everything below 1 ms before the first conv)
big = [r for r in rows if r[2] >= 0.6]      # conv0 runs ~1.06 ms since round 6 (was 2.35); the set-up launches stay below 0.3 ms
n = len(spec)
assert len(big) >= n, (len(big), n)
chunks = [big[i:i + n] for i in range(0, len(big) - n + 1, n)]
print(__doc__.split("\n\n")[0])
print()
print(f"{'launch':58s} {'GFLOP':>8s} {'x3 TF':>6s} {'MFMA ms':>8s} {'@2.0GHz':>8s} {'GB':>6s} {'HBM ms':>7s} {'meas ms':>8s} {'meas/max':>8s}  bound")
tot = dict(mfma=0.0, mfc=0.0, hbm=0.0, meas=0.0, bnd=0.0, bndc=0.0)
fam = {}
for i, (name, family, fl, by) in enumerate(spec):
    meas = sum(c[i][2] for c in chunks) / len(chunks)
    mf = 3 * fl / PEAK * 1e3
    mfc = mf / CLK
    hb = by / HBM * 1e3
    bound = max(mf, hb)
    which = "MFMA" if mf >= hb else "HBM"
    if family == "last":
        mf = mfc = 0.0
        bound, which = hb, "HBM (reduction, no MFMA)"
    print(f"{name:58s} {fl / 1e9:8.0f} {3 * fl / 1e12:6.2f} {mf:8.2f} {mfc:8.2f} {by / 1e9:6.2f} {hb:7.2f} {meas:8.2f} {meas / bound:8.2f}  {which}")
    tot["mfma"] += mf; tot["mfc"] += mfc; tot["hbm"] += hb; tot["meas"] += meas; tot["bnd"] += bound; tot["bndc"] += max(mfc, hb)
    f = fam.setdefault(family, [0.0, 0.0, 0.0])
    f[0] += meas; f[1] += bound; f[2] += max(mfc, hb)
print()
print(f"per chunk of {U} utterances: measured {tot['meas']:.1f} ms | sum of max(MFMA @2.5 PF, HBM @6.3 TB/s) {tot['bnd']:.1f} ms "
      f"({tot['meas'] / tot['bnd']:.2f} x) | with the MFMA rate at the 2.0 GHz the kernels clock at {tot['bndc']:.1f} ms ({tot['meas'] / tot['bndc']:.2f} x)")
print(f"   MFMA alone {tot['mfma']:.1f} ms (@2.0 GHz {tot['mfc']:.1f}), HBM alone {tot['hbm']:.1f} ms; a B = 32 pass = two chunks")
print()
print(f"{'kernel family':14s} {'meas ms':>8s} {'bound ms':>9s} {'meas/bound':>10s} {'@2.0GHz':>8s} {'gap ms':>7s}")
R5 = {"fused unit": 74.5, "ring 7-tap": 58.0, "ring 2-tap": 30.8, "1x1 (s1)": 17.8, "last": 2.4}     # profiles/r05_dac_two_rooflines.txt
for k, (m, b, bc) in sorted(fam.items(), key=lambda kv: -(kv[1][0] - kv[1][2])):
    print(f"{k:14s} {m:8.1f} {b:9.1f} {m / b:10.2f} {m / bc:8.2f} {m - bc:7.1f}   round 5: {R5[k]:5.1f} ms = {R5[k] / bc:4.2f} x the @2.0GHz bound  ({m - R5[k]:+5.1f} ms)")
