#!/bin/bash
# round-3 experiment D: (1) what bounds k_hfused -- phase 1 reading an L1- / L2-resident window instead of HBM (same load
# instructions; proxies, results garbage); (2) SGBM time per frame against the batch size (wave mix of k_hfused, k_vdd slot fill)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/exp_r3d; mkdir -p $O
{
for v in base p1l1 p1l2 base; do echo "== $v"; V3D_HIP_LIB=$PWD/var_libs/lib_$v.so QB_BATCH=30 ROUNDS=2 VARIANTS="HSPLIT=0" timeout -k 10 200 python3 tools/sgbm_ab.py 2>&1 | grep -E "total"; done
for b in 30 34 45 60 68 90 102; do echo "== batch $b"; QB_BATCH=$b ROUNDS=2 VARIANTS="HSPLIT=0" timeout -k 10 300 python3 tools/sgbm_ab.py 2>&1 | grep -E "total|min"; done
} > $O/log.txt 2>&1
cat $O/log.txt
