#!/bin/bash
# Builds scratch_exp/<name>.so = libnfx.so with ONE source recompiled under extra flags (A/B runs: scripts/lib_exp.py).
#   bash scripts/build_variant.sh gabor_noeven k_gabor.hip -DNFX_GB_EVEN3_MAX_CT=1
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; SRC=$2; shift 2
CS=$ROOT/nuclei-feature-extraction_amd/csrc
mkdir -p $ROOT/scratch_exp/_obj_$NAME
make -C $CS -s >/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function --offload-arch=gfx950 "$@" -x hip -c $CS/$SRC -o $ROOT/scratch_exp/_obj_$NAME/$SRC.o
OBJS=$(ls $CS/_obj/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/scratch_exp/$NAME.so $OBJS $ROOT/scratch_exp/_obj_$NAME/$SRC.o -ldl
echo built scratch_exp/$NAME.so
