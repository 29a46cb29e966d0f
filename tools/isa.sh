#!/bin/bash
# tools/isa.sh SRC.hip [-Dflags] -> /tmp/<SRC>.s (gfx950 device assembly of one csrc source)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off --cuda-device-only -S "$@" $ROOT/video-3d-pipeline_amd/csrc/$SRC -o /tmp/${SRC%.hip}.s 2>&1 | grep -E "error" | head
echo /tmp/${SRC%.hip}.s
