#!/bin/bash
# SQ counter passes over tools/sgbm_ab.py (one variant, one round); summaries land in gpurun_out/pmc_sq_*.txt
# usage (on the GPU box): [PROG=tools/gf_one.py] bash tools/pmc_sq.sh [batch]
cd /tmp && export TMPDIR=/tmp
export QB_BATCH=${1:-30} ROUNDS=1 VARIANTS="${VARIANTS:-HFUSED=1}"
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_sq1 gpurun_out/pmc_sq2
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU \
    -d gpurun_out/pmc_sq1 -o run --output-format csv -- python3 ${PROG:-tools/sgbm_ab.py} > gpurun_out/pmc_sq1.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES \
    -d gpurun_out/pmc_sq2 -o run --output-format csv -- python3 ${PROG:-tools/sgbm_ab.py} > gpurun_out/pmc_sq2.log 2>&1 &&
python3 tools/pmc_sq_summary.py gpurun_out/pmc_sq1/run_counter_collection.csv gpurun_out/pmc_sq2/run_counter_collection.csv > gpurun_out/pmc_sq.txt
