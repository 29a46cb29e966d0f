#!/bin/bash
# round-3 experiment B: HBM fetch bytes of k_hfused, stock library vs the 12-bit-C load proxy (one --pmc FETCH_SIZE pass each)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export QB_BATCH=30 ROUNDS=1 VARIANTS="HSPLIT=0"
for v in base c12h; do
  rm -rf gpurun_out/pmc_$v
  V3D_HIP_LIB=$PWD/var_libs/lib_$v.so timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_$v -o run --output-format csv -- python3 tools/sgbm_ab.py > gpurun_out/pmc_$v.log 2>&1 || exit 1
  echo "== $v"; python3 tools/pmc_sq_summary.py gpurun_out/pmc_$v/run_counter_collection.csv | grep -E "k_hfused|k_vdd|k_cost|kernel" 
done
