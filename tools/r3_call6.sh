#!/bin/bash
# A/B of two stamps libraries on the same box (graph replay), then tests + bench of the product library
set -o pipefail
P=$PWD; O=$P/gpurun_out/${1:-r3c6}; mkdir -p $O
for tag in base new base2 new2; do
  lib=libparler_tts_hip_stamps.so; case $tag in base*) lib=libparler_tts_hip_stamps_base.so;; esac
  PTTS_HIP_LIB=$P/parler_tts_amd/csrc/$lib PTTS_STAMPS_OUT=$O/stamps_$tag.npy python3 tools/pmc_attn.py 40 24 1250 > $O/stamps_$tag.log 2>&1
  python3 tools/stamps_report.py $O/stamps_$tag.npy > $O/stamps_$tag.txt 2>&1
  echo "== $tag"; tail -n 10 $O/stamps_$tag.txt
done
rm -f $O/stamps_base2.npy $O/stamps_new2.npy
if [ "$2" != "nobench" ]; then
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fp32_tokens or bf16_teacher or graph_replay or finished_utterances or baseline_size or long_description or page_attention or paged_kv or variants" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -n 4 $O/pytest.log
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python3 -c "
import json,sys; d=json.load(open('$O/bench.json')); print(d['value'], d['phase_ms_per_step_rank0'], d['roofline']['avg_launch_us'])"
fi
exit 0
