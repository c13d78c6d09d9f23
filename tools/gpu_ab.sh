#!/bin/bash
# Decode-loop (or whole-pass) A/B of variants on ONE GPU box, two alternating rounds:
#   bash tools/gpu_ab.sh OUT [--full] VARIANT [VARIANT ...]
# VARIANT = "-" (the tree's library, no switch) | NAME=VALUE (an environment switch of the library, e.g. PTTS_GEMM_NO_NB3=1)
#         | file.so (another build of the library in parler_tts_amd/csrc, selected with PTTS_HIP_LIB)
# --full keeps the codec in the pass (default: --no-dac, the decode loop alone).  Results: gpurun_out/OUT/ab.txt.
# Stops at the first failing run (no further GPU step after a failure) and exits with its return code.
set -o pipefail
P=$PWD; O=$P/gpurun_out/$1; mkdir -p "$O"; shift
DAC="--no-dac"; if [ "$1" = "--full" ]; then DAC=""; shift; fi
: > "$O/ab.txt"
for rep in 1 2; do
  for v in "$@"; do
    unset PTTS_HIP_LIB
    case "$v" in
      -) env_kv="" ;;
      *.so) env_kv="PTTS_HIP_LIB=$P/parler_tts_amd/csrc/$v" ;;
      *=*) env_kv="$v" ;;
      *) echo "bad variant $v" >&2; exit 64 ;;
    esac
    env $env_kv python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-pass --no-sampling-pass $DAC > "$O/b.json" 2> "$O/b.err"
    rc=$?
    if [ $rc -ne 0 ]; then echo "variant [$v] rep $rep FAILED rc=$rc" | tee -a "$O/ab.txt"; tail -n 5 "$O/b.err"; exit $rc; fi
    python3 - "$v" "$rep" "$O/b.json" <<'PY' | tee -a "$O/ab.txt"
import json, sys
d = json.load(open(sys.argv[3]))
ph = d["phase_ms_per_step_rank0"]
print(f"[{sys.argv[1]}] rep {sys.argv[2]}: value {d['value']} tok/s, decode_ms {ph['decode_ms']}, post_ms {ph['post_ms']}, "
      f"attn_us {d['roofline']['avg_launch_us']}, passes identical {d['passes_bit_identical_rank0']}")
PY
  done
done
exit 0
