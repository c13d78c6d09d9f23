#!/bin/bash
set -o pipefail
P=$PWD; O=$P/gpurun_out/r3c9; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
run_pmc() {   # tag, min dispatch id, counters, program...
  local tag=$1 minid=$2 ctr=$3; shift 3
  timeout -k 10 240 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$tag -- "$@" > $O/$tag.log 2>&1
  local rc=$?
  python3 $P/tools/pmc_summary.py /tmp/pmc_$tag $O/$tag.json $minid >> $O/$tag.log 2>&1
  rm -rf /tmp/pmc_$tag
  echo "$tag rc=$rc: $(tail -n 1 $O/$tag.log | cut -c1-160)"
}
DEC="python3 $P/tools/pmc_attn.py 31 24 1250"
run_pmc dec_icache 2600 "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES" $DEC
run_pmc dec_dcache 2600 "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_TC_DATA_READ_REQ SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU" $DEC
exit 0
