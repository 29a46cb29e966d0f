#!/bin/bash
# Experiment builds: recompile ONE source of csrc with extra -D flags and link it with the stock objects.
#   tools/build_variant.sh NAME v3d_sgbm.hip "-DHF_OPT=2" [kernel-name-pattern]
# -> var_libs/lib_NAME.so (git-ignored; travels to the GPU box); select it with V3D_HIP_LIB=$PWD/var_libs/lib_NAME.so.
# Prints VGPRs / scratch / occupancy / LDS of the kernels matching the pattern.
set -e
NAME=$1; SRC=$2; DEFS=$3; PAT=${4:-.}
ROOT=$(cd "$(dirname "$0")/.." && pwd); C=$ROOT/video-3d-pipeline_amd/csrc
mkdir -p $ROOT/var_libs $C/build/var
make -s -C $C > /dev/null
cd $C
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -ffp-contract=off $DEFS \
    -Rpass-analysis=kernel-resource-usage -c $SRC -o build/var/${NAME}.o 2> build/var/${NAME}.log || { grep -E "error" build/var/${NAME}.log; exit 1; }
grep -E "Function Name|VGPRs:|ScratchSize|Occupancy|LDS Size" build/var/${NAME}.log | sed 's/.*remark: *//; s/ *\[-Rpass.*//' | paste - - - - - | grep -E "$PAT" | sort -u | sed "s/^/[$NAME] /"
OBJS=""; for b in v3d_sgbm v3d_guided v3d_pre v3d_corr v3d_blend v3d_api; do if [ "$b.hip" = "$SRC" ] || [ "$b.cpp" = "$SRC" ]; then OBJS="$OBJS build/var/${NAME}.o"; else OBJS="$OBJS build/$b.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/var_libs/lib_${NAME}.so $OBJS 2>&1 | tail -3
