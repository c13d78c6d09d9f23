#!/bin/bash
# round 3, GPU call 1: production-size codec parity tests, then two rocprofv3 --pmc passes that separate "many launches"
# from "large pool" as the trigger of the profiler abort
set -o pipefail
O=gpurun_out/r3c1; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "production_kernels or one_second_of_audio or dac_decode_kernels" > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.log; tail -n 5 $O/pytest.log
P=$PWD
cd /tmp
PTTS_DUMP_MAPS=$P/$O/A_maps.txt PTTS_DUMP_ALLOCS=$P/$O/A_allocs.txt timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $P/$O/A -- python3 $P/tools/pmc_attn.py 580 1 2000 > $P/$O/A.log 2>&1
echo "A (few launches, 338 MB pool) rc=$?"; tail -n 3 $P/$O/A.log
PTTS_DUMP_MAPS=$P/$O/B_maps.txt PTTS_DUMP_ALLOCS=$P/$O/B_allocs.txt timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $P/$O/B -- python3 $P/tools/pmc_attn.py 1700 1 0 > $P/$O/B.log 2>&1
echo "B (18.7k launches, 222 MB pool) rc=$?"; tail -n 3 $P/$O/B.log
exit 0
