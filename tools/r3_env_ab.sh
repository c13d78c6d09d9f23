#!/bin/bash
# decode-loop A/B of ONE library under an environment switch: bash tools/r3_env_ab.sh OUT VAR  (VAR=1 vs unset, two alternating rounds)
set -o pipefail
P=$PWD; O=$P/gpurun_out/$1; mkdir -p $O; VAR=$2
for rep in 1 2; do
for val in "" 1; do
  if [ -n "$val" ]; then export $VAR=$val; else unset $VAR; fi
  python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-dac > $O/b.json 2> $O/b.err
  python3 -c "
import json; d=json.load(open('$O/b.json')); print('$VAR=[$val] rep $rep: decode_ms', d['phase_ms_per_step_rank0']['decode_ms'], 'attn_us', d['roofline']['avg_launch_us'], 'identical', d['passes_bit_identical_rank0'])"
done
done
exit 0
