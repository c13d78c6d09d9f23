#!/bin/bash
# rocprofv3 --pmc counter passes, one counter group per process (separate passes, as MI355X_MICROARCH.md prescribes), summarised
# ON THE BOX by tools/pmc_summary.py (the raw CSVs exceed what gpurun brings back and are deleted):
#   bash tools/gpu_pmc.sh OUT decoder      the six passes of a decode step at the bench shape (tools/pmc_attn.py 31 24 1250:
#                                          24 layers, batch 32, S = 1251 .. 1281, ~7.6k dispatches - below the profiler's AQL-ring
#                                          wrap, profiles/r03_pmc_abort_root_cause.txt)
#   bash tools/gpu_pmc.sh OUT dac          the four passes of the codec decode at 2 x 2571 frames (tools/dac_profile.py 2 2571)
#   bash tools/gpu_pmc.sh OUT prefill      the same four passes over T5 encoder + prefill at the bench shape (tools/prefill_profile.py)
# Tables: python3 tools/pmc_report.py gpurun_out/OUT.  A pass that fails or is killed ends the call with its return code.
set -o pipefail
P=$PWD; O=$P/gpurun_out/$1; mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp
run_pmc() {   # tag, min dispatch id, counters, program...
  local tag=$1 minid=$2 ctr=$3; shift 3
  local t0=$(date +%s)
  timeout -k 10 240 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$tag -- "$@" > "$O/$tag.log" 2>&1
  local rc=$?
  python3 "$P/tools/pmc_summary.py" /tmp/pmc_$tag "$O/$tag.json" $minid >> "$O/$tag.log" 2>&1
  rm -rf /tmp/pmc_$tag
  echo "$tag rc=$rc $(( $(date +%s) - t0 )) s: $(tail -n 1 "$O/$tag.log" | cut -c1-160)"
  [ $rc -eq 0 ] || exit $rc      # a pass that failed or was killed: no further GPU step in this call
}
case "$2" in
  decoder)
    DEC="python3 $P/tools/pmc_attn.py 31 24 1250"
    run_pmc dec_tcc1 2600 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" $DEC
    run_pmc dec_tcc2 2600 "TCC_EA0_RDREQ_DRAM_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" $DEC
    run_pmc dec_tcc3 2600 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" $DEC
    run_pmc dec_tcp 2600 "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" $DEC
    run_pmc dec_sq1 2600 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" $DEC
    run_pmc dec_grbm 2600 "GRBM_GUI_ACTIVE" $DEC ;;
  dac)
    DAC="python3 $P/tools/dac_profile.py 2 2571"
    run_pmc dac_sq1 0 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" $DAC
    run_pmc dac_sq2 0 "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES" $DAC
    run_pmc dac_tcc 0 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" $DAC
    run_pmc dac_grbm 0 "GRBM_GUI_ACTIVE" $DAC ;;
  prefill)
    # T5 encoder + prefill of the benchmark shape (tools/prefill_profile.py: 4 short generate() calls, ~1.3k dispatches): the MFMA tile
    # GEMM and the attention kernels of the one-off phases; same four passes as the codec
    PF="python3 $P/tools/prefill_profile.py 1"
    run_pmc dac_sq1 0 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" $PF
    run_pmc dac_sq2 0 "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES" $PF
    run_pmc dac_tcc 0 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" $PF
    run_pmc dac_grbm 0 "GRBM_GUI_ACTIVE" $PF ;;
  *) echo "usage: gpu_pmc.sh OUT decoder|dac|prefill" >&2; exit 64 ;;
esac
du -sh "$O"
exit 0
