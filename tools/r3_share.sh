#!/bin/bash
# final-build repeatability under GPU sharing (the folded bf16 GEMMs take their LayerNorm statistics from the matrix cores since
# round 3): two processes, 40 greedy passes of 200 steps each, every pass compared with the process's first; then two native
# hammers of the folded instance (prologue 5 = folded, 32-row workgroups; 6 = folded, half-split) side by side
set -o pipefail
mkdir -p gpurun_out
PASSES=40 bash tools/share_bisect.sh r3share bf16
( timeout -k 10 200 python3 tools/gemm_hammer.py 5 32 4096 1024 200000 bf16 > gpurun_out/r3_hammer_A.log 2>&1 & )
timeout -k 10 200 python3 tools/gemm_hammer.py 6 32 1024 1024 200000 bf16 > gpurun_out/r3_hammer_B.log 2>&1
sleep 15
tail -n 3 gpurun_out/r3_hammer_A.log gpurun_out/r3_hammer_B.log
exit 0
