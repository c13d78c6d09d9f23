#!/bin/bash
# round 3, GPU call 2: counter passes on the bench workload's own shape (24 layers, batch 32, S = 1251 .. 1311), each process
# kept below ~13k dispatches (the profiler reads one packet past the end of the 16384-packet AQL ring when a queue wraps:
# profiles/r03_pmc_abort_root_cause.txt); codec counter passes at 2 x 2571 frames; then the r03 baseline bench + kernel stats
set -o pipefail
P=$PWD; O=$P/gpurun_out/r3c2; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
run_pmc() {   # tag, counters, program...
  local tag=$1 ctr=$2; shift 2
  timeout -k 10 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/$tag -- "$@" > $O/$tag.log 2>&1
  echo "$tag rc=$?"; tail -n 2 $O/$tag.log | cut -c1-200
}
DEC="python3 $P/tools/pmc_attn.py 61 24 1250"
run_pmc dec_tcc1 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" $DEC
run_pmc dec_tcc2 "TCC_EA0_RDREQ_DRAM_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" $DEC
run_pmc dec_tcc3 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" $DEC
run_pmc dec_sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" $DEC
DAC="python3 $P/tools/dac_profile.py 2 2571"
run_pmc dac_sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" $DAC
run_pmc dac_sq2 "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES" $DAC
run_pmc dac_tcc "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" $DAC
cd $P
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc=$?"; cut -c1-400 $O/bench_default.json
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $P/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/stats.log 2>&1
echo "stats rc=$?"
exit 0
