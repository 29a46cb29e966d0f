#!/bin/bash
# round-3 experiment C: what k_hfused waits for -- TA / TCP / UTCL1 / TCC counter passes over tools/sgbm_ab.py (stock library)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export QB_BATCH=30 ROUNDS=1 VARIANTS="HSPLIT=0"
LIB=${1:-base}
pass() { n=$1; shift; rm -rf gpurun_out/pmcc_$n
  V3D_HIP_LIB=$PWD/var_libs/lib_$LIB.so timeout -k 10 200 rocprofv3 --pmc "$@" -d gpurun_out/pmcc_$n -o run --output-format csv -- python3 tools/sgbm_ab.py > gpurun_out/pmcc_$n.log 2>&1 || { echo "pass $n failed"; tail -3 gpurun_out/pmcc_$n.log; return; }
  python3 tools/pmc_sq_summary.py gpurun_out/pmcc_$n/run_counter_collection.csv | grep -E "k_hfused|k_vdd<|k_cost"; rm -rf gpurun_out/pmcc_$n; }
pass a TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum
pass b TCP_TCR_TCP_STALL_CYCLES_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass c TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
pass d TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_avr TCC_REQ_sum
pass e TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum
pass f MemUnitStalled TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum
pass g GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM
