#!/usr/bin/env python3
"""ISA lint for a gfx950 code-generation trap: MFMA instructions ignore the EXEC mask.  When a uniform-but-VGPR-derived
condition guards a short block, hipcc predicates it with s_and_saveexec and DROPS the s_cbranch_execz skip, so a
"skipped" MFMA still runs - on whatever its operand registers hold.  This lists every v_mfma whose nearest preceding
control instruction is a saveexec without an execz branch.

    python tools/check_mfma_guards.py parler_tts_amd/csrc/gemm.hip [more.hip ...]      (exit code 1 if any is found)
"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASS_THROUGH = ("v_mfma", "s_waitcnt", "s_nop", "ds_read", "v_mov", "v_pk_", "v_accvgpr", "s_mov", "v_lshl", "v_add", "s_add", "s_lshl")


def scan(asm_text):
    found, n_mfma, kernel, clean = [], 0, None, []
    for line in asm_text.split("\n"):
        t = line.strip()
        if not t or t.startswith(";"):
            continue
        head = t.split(";")[0].strip()
        if head.endswith(":") and not head.startswith("."):      # function label
            kernel, clean = head[:-1], []
            continue
        if kernel is None or head.startswith(".") and not head.startswith(".LBB"):
            continue
        clean.append(head)
        if head.startswith("v_mfma"):
            n_mfma += 1
            j = len(clean) - 2
            while j >= 0 and (clean[j].startswith(PASS_THROUGH) or clean[j].endswith(":")):
                j -= 1
            if j >= 0 and "saveexec" in clean[j]:
                found.append((kernel, clean[j], head))
    return n_mfma, found


def main(paths):
    rc = 0
    for src in paths:
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "k.s")
            cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                   "-I" + os.path.join(ROOT, "parler_tts_amd", "csrc"), "-S", "--cuda-device-only", "-o", out, src]
            subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
            n, found = scan(open(out).read())
        print(f"{src}: {n} MFMAs, {len(found)} guarded by EXEC masking only")
        for k, ctl, ins in found[:20]:
            print(f"  {k[:90]}\n      {ctl}\n      {ins}")
        rc |= 1 if found else 0
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:] or [os.path.join(ROOT, "parler_tts_amd", "csrc", f) for f in ("gemm.hip", "conv.hip")]))
