#!/bin/bash
# round 3, GPU call 3: in-kernel stamps of one decode step at the bench shape (diagnostic library), then the new GPU tests
set -o pipefail
P=$PWD; O=$P/gpurun_out/r3c3; mkdir -p $O
PTTS_HIP_LIB=$P/parler_tts_amd/csrc/libparler_tts_hip_stamps.so PTTS_STAMPS_OUT=$O/stamps_graph.npy python3 tools/pmc_attn.py 40 24 1250 > $O/stamps_graph.log 2>&1
echo "stamps graph rc=$?"; tail -n 2 $O/stamps_graph.log
PTTS_NO_GRAPH=1 PTTS_HIP_LIB=$P/parler_tts_amd/csrc/libparler_tts_hip_stamps.so PTTS_STAMPS_OUT=$O/stamps_eager.npy python3 tools/pmc_attn.py 40 24 1250 > $O/stamps_eager.log 2>&1
echo "stamps eager rc=$?"
python3 tools/stamps_report.py $O/stamps_graph.npy > $O/stamps_graph.txt 2>&1; tail -n 12 $O/stamps_graph.txt
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "variants or fp32_long_horizon or production_kernels or one_second" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -n 8 $O/pytest.log
exit 0
