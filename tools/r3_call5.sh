#!/bin/bash
# stamps of the current build (graph replay + eager) and a quick correctness gate
set -o pipefail
P=$PWD; O=$P/gpurun_out/${1:-r3c5}; mkdir -p $O
PTTS_HIP_LIB=$P/parler_tts_amd/csrc/libparler_tts_hip_stamps.so PTTS_STAMPS_OUT=$O/stamps_graph.npy python3 tools/pmc_attn.py 40 24 1250 > $O/stamps_graph.log 2>&1
echo "stamps graph rc=$?"
python3 tools/stamps_report.py $O/stamps_graph.npy > $O/stamps_graph.txt 2>&1; tail -n 11 $O/stamps_graph.txt
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fp32_tokens or bf16_teacher or graph_replay or finished_utterances or baseline_size or long_description or gemm_against or folded_layernorm or partial_bursts or page_attention or paged_kv" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -n 4 $O/pytest.log
python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python3 -c "
import json,sys; d=json.load(open('$O/bench.json')); print(d['value'], d['phase_ms_per_step_rank0'], d['roofline']['avg_launch_us'])"
exit 0
