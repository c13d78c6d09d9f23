#!/usr/bin/env python3
"""ISA lint for the MFMA -> inline-asm read hazard on gfx950.

hipcc's hazard recognizer pads the wait states between an MFMA (XDL) write of a VGPR and a later read of that VGPR by a
VALU / LDS / VMEM instruction ONLY for instructions it models.  An `asm volatile` statement is opaque to it
(cdna_hip_programming.md section 5.7 item 2: "every pair whose producer OR consumer is inside the string needs its
s_nop inside the string"), so an accumulator handed straight to an inline-asm `ds_write` / `global_store` / VALU
instruction can be read while the matrix pipe is still retiring it: a few lanes of one accumulator register come out
stale, on some launches only.  LLVM's own table for gfx950 (GCNHazardRecognizer, XDL write VGPR -> VALU / memory
read): passes + 3 wait states = 7 for a 4-pass, 11 for an 8-pass and 19 for a 16-pass MFMA.

This tool compiles the given .hip files to gfx950 assembly, builds each function's control-flow graph and walks
BACKWARDS from every inline-asm block over all paths, counting wait states (one per instruction, N + 1 for `s_nop N`),
until it meets an MFMA whose destination overlaps a VGPR the block reads.  Any such pair closer than NEED wait states
is reported (exit code 1).

    python tools/check_mfma_asm_hazard.py [file.hip ...]          (default: every product .hip file)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "parler_tts_amd", "csrc")

# wait states LLVM requires on gfx950 between an XDL write and a non-MFMA read of the same VGPR, by mnemonic shape.
# (16x16x32 / 16x16x16 bf16-f16 and 16x16x4 f32: 8 passes at most -> 11; 32x32: 16 passes -> 19.)


def need_for(mnemonic):
    if "32x32" in mnemonic:
        return 19
    return 11


REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def parse_functions(asm_text):
    """-> {name: [(label or None, text, in_asm)]} one entry per instruction, labels attached to the next instruction"""
    funcs, cur, pending, in_asm = {}, None, [], False
    for line in asm_text.split("\n"):
        t = line.strip()
        if not t:
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        head = t.split(";")[0].strip()
        if not head:
            continue
        if head.endswith(":"):
            name = head[:-1]
            if name.startswith(".LBB") or name.startswith(".Ltmp") or name.startswith(".L"):
                pending.append(name)
            elif not name.startswith("."):
                cur = funcs.setdefault(name, [])
                pending = []
            continue
        if cur is None or head.startswith("."):
            continue
        cur.append((tuple(pending), head, in_asm))
        pending = []
        if head.startswith("s_endpgm"):
            pass
    return funcs


def analyse(name, ins):
    label_at = {}
    for i, (labels, _, _) in enumerate(ins):
        for lb in labels:
            label_at[lb] = i
    preds = [[] for _ in ins]
    for i, (_, text, _) in enumerate(ins):
        op = text.split()[0]
        falls = not (op in ("s_branch", "s_endpgm", "s_setpc_b64") or op.startswith("s_endpgm"))
        if falls and i + 1 < len(ins):
            preds[i + 1].append(i)
        if op.startswith("s_cbranch") or op == "s_branch":
            tgt = text.split()[-1]
            if tgt in label_at:
                preds[label_at[tgt]].append(i)
    found = []
    for i, (_, text, in_asm) in enumerate(ins):
        if not in_asm:
            continue
        op = text.split()[0]
        if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep")):
            continue
        reads = regs_of(text.split(None, 1)[1] if " " in text else "")
        if not reads:
            continue
        # backward walk: (index, wait states accumulated between that instruction and the asm instruction)
        best = {}
        stack = [(p, 0) for p in preds[i]]
        while stack:
            j, dist = stack.pop()
            if dist >= 19 or best.get(j, 99) <= dist:
                continue
            best[j] = dist
            t = ins[j][1]
            o = t.split()[0]
            if o.startswith("v_mfma") or o.startswith("v_smfmac"):
                dst = regs_of(t.split(None, 1)[1].split(",")[0])
                if dst & reads and dist < need_for(o):
                    found.append((name, t, text, dist, need_for(o)))
                if dst & reads:
                    continue
            step = 1
            if o == "s_nop":
                step = int(t.split()[1], 0) + 1
            for p in preds[j]:
                stack.append((p, dist + step))
    return found


def compile_to_asm(src, out):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
           "-I" + CSRC, "-S", "--cuda-device-only", "-o", out, src]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)


def main(paths):
    rc = 0
    for src in paths:
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "k.s")
            compile_to_asm(src, out)
            funcs = parse_functions(open(out).read())
        n_asm, found = 0, []
        for name, ins in funcs.items():
            n_asm += sum(1 for x in ins if x[2])
            found += analyse(name, ins)
        print(f"{os.path.relpath(src, ROOT)}: {len(funcs)} functions, {n_asm} inline-asm instructions, "
              f"{len(found)} MFMA results read by inline asm too early")
        seen = set()
        for k, mf, rd, dist, need in found:
            key = (k, mf, rd)
            if key in seen:
                continue
            seen.add(key)
            dem = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
            print(f"  {dem[:110]}\n      {mf}\n      -> {rd}    ({dist} wait states, {need} required)")
        rc |= 1 if found else 0
    return rc


if __name__ == "__main__":
    default = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    sys.exit(main(sys.argv[1:] or default))
