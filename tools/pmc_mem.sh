#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (separate passes) of PROG's kernels -> gpurun_out/pmc_mem.txt (KiB per launch; FETCH x2 on gfx950)
cd /tmp && export TMPDIR=/tmp
export QB_BATCH=${1:-30} ROUNDS=1 VARIANTS="${VARIANTS:-HFUSED=1}"
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_m1 gpurun_out/pmc_m2
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_m1 -o run --output-format csv -- python3 ${PROG:-tools/sgbm_ab.py} > gpurun_out/pmc_m1.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_m2 -o run --output-format csv -- python3 ${PROG:-tools/sgbm_ab.py} > gpurun_out/pmc_m2.log 2>&1 &&
python3 tools/pmc_sq_summary.py gpurun_out/pmc_m1/run_counter_collection.csv gpurun_out/pmc_m2/run_counter_collection.csv > gpurun_out/pmc_mem.txt
