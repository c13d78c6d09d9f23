#!/bin/bash
# decode-loop A/B of product-library variants on ONE box: bash tools/r3_bench_ab.sh OUT lib1.so lib2.so ...  (two alternating rounds)
set -o pipefail
P=$PWD; O=$P/gpurun_out/$1; mkdir -p $O; shift
for rep in 1 2; do
for lib in "$@"; do
  PTTS_HIP_LIB=$P/parler_tts_amd/csrc/$lib python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-dac > $O/b.json 2> $O/b.err
  python3 -c "
import json; d=json.load(open('$O/b.json')); print('$lib rep $rep: decode_ms', d['phase_ms_per_step_rank0']['decode_ms'], 'attn_us', d['roofline']['avg_launch_us'], 'identical', d['passes_bit_identical_rank0'])"
done
done
exit 0
