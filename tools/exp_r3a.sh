#!/bin/bash
# round-3 experiment batch A (one gpurun call = one box): SGBM stage times of the variant libraries in var_libs/
# (built by tools/build_variant.sh: base, c12h/c12v/c12c/c12all = -DV3D_X_C12=1/2/4/7, costchain = -DV3D_X_COSTCHAIN=1)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/exp_r3a; mkdir -p $O
export QB_BATCH=30 ROUNDS=2
run() { echo "== $1 [$2]"; V3D_HIP_LIB=$PWD/var_libs/lib_$1.so VARIANTS="$2" timeout -k 10 200 python3 tools/sgbm_ab.py 2>&1 | grep -v "^$" ; }
{
run base "HSPLIT=0;HSPLIT=1"
run c12h "HSPLIT=0"
run c12v "HSPLIT=0"
run c12c "HSPLIT=0"
run c12all "HSPLIT=0"
run costchain "HSPLIT=0"
run ril4 "HSPLIT=0"
run ril4c12 "HSPLIT=0"
run base "HSPLIT=0"
} > $O/log.txt 2>&1
tail -80 $O/log.txt
