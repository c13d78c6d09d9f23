#!/bin/bash
# two processes share the GPU; each checks its own repeatability.  usage: share_bisect.sh <tag> <dtype> [ENV=V ...]
tag=$1; dt=$2; shift 2
( env "$@" timeout -k 10 300 python tools/repeat_check.py 200 ${PASSES:-6} "$tag/A" $dt > gpurun_out/sb_${tag}_A.log 2>&1 & )
env "$@" timeout -k 10 300 python tools/repeat_check.py 200 ${PASSES:-6} "$tag/B" $dt > gpurun_out/sb_${tag}_B.log 2>&1
sleep 12
grep -h "pass" gpurun_out/sb_${tag}_A.log gpurun_out/sb_${tag}_B.log | grep -v "pass 0" | cut -c1-150
