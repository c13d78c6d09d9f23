#!/bin/bash
# counter passes of a decode step at the bench shape on the FINAL build of the round (the six decoder passes of tools/r3_call4.sh)
set -o pipefail
P=$PWD; O=$P/gpurun_out/r3pmcf; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
run_pmc() {   # tag, min dispatch id, counters, program...
  local tag=$1 minid=$2 ctr=$3; shift 3
  local t0=$(date +%s)
  timeout -k 10 240 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$tag -- "$@" > $O/$tag.log 2>&1
  local rc=$?
  python3 $P/tools/pmc_summary.py /tmp/pmc_$tag $O/$tag.json $minid >> $O/$tag.log 2>&1
  rm -rf /tmp/pmc_$tag
  echo "$tag rc=$rc $(( $(date +%s) - t0 )) s: $(tail -n 1 $O/$tag.log | cut -c1-160)"
  [ $rc -eq 0 ] || exit 1      # a pass that was killed: no further GPU step in this call
}
DEC="python3 $P/tools/pmc_attn.py 31 24 1250"
run_pmc dec_tcc1 2600 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" $DEC
run_pmc dec_tcc2 2600 "TCC_EA0_RDREQ_DRAM_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" $DEC
run_pmc dec_tcc3 2600 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" $DEC
run_pmc dec_tcp 2600 "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" $DEC
run_pmc dec_sq1 2600 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" $DEC
run_pmc dec_grbm 2600 "GRBM_GUI_ACTIVE" $DEC
du -sh $O
exit 0
