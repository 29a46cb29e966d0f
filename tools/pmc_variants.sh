#!/bin/bash
# LDS counters of k_cost for each experiment build in tools/variants/ (built with -DV3D_COST_DBG=n)
cd /tmp && export TMPDIR=/tmp
export QB_BATCH=${1:-30} ROUNDS=1 VARIANTS="HFUSED=1"
cd "$GRAFT_REPO_ROOT"
: > gpurun_out/pmc_var.txt
for lib in tools/variants/*.so; do
    export V3D_HIP_LIB=$PWD/$lib
    rm -rf gpurun_out/pmc_var
    timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
        -d gpurun_out/pmc_var -o run --output-format csv -- python3 tools/sgbm_ab.py > gpurun_out/pmc_var.log 2>&1 || exit 1
    echo "== $lib" >> gpurun_out/pmc_var.txt
    grep "cost=" gpurun_out/pmc_var.log | sed 's/.*\(cost=[0-9.]*\).*/\1/' >> gpurun_out/pmc_var.txt
    python3 tools/pmc_sq_summary.py gpurun_out/pmc_var/run_counter_collection.csv | grep k_cost >> gpurun_out/pmc_var.txt
done
