# PMC passes over three pair-list convolution shapes (alone, 20 launches each): what is busy in the block loop?
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc3
rocprofv3 -L > gpurun_out/pmc3/counters_list.txt 2>&1 || true
grep -c "" gpurun_out/pmc3/counters_list.txt
run_pass() {  # name, counters
  for shape in "0 16 16" "1 32 32" "3 64 64"; do
    tag=$(echo $shape | tr ' ' '_')
    (cd /tmp && timeout -k 10 200 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d /tmp/pmc_$1_$tag -- python3 $GRAFT_REPO_ROOT/tools/bench_pairs_one.py $shape > /tmp/pmc_$1_$tag.log 2>&1) || { echo "pass $1 $shape failed"; tail -3 /tmp/pmc_$1_$tag.log; }
    echo "== pass $1 shape $shape" >> gpurun_out/pmc3/table.txt
    python tools/pmc_table.py /tmp/pmc_$1_$tag 2>/dev/null | grep -A40 "k_gconv_pairs" >> gpurun_out/pmc3/table.txt || true
  done
}
run_pass sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
run_pass sq2 "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM"
run_pass ta "TA_BUSY_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE GRBM_COUNT"
run_pass tcp "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"
run_pass lds "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_MFMA"
cat gpurun_out/pmc3/table.txt | tail -150
