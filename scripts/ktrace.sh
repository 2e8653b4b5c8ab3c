#!/bin/bash
# kernel-trace stats of one bench invocation (run on the GPU box): scripts/ktrace.sh <name> <bench args...>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
NAME=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/kt_$NAME --output-format csv -- python3 $ROOT/bench.py "$@" --no-cpu-baseline > $ROOT/gpurun_out/kt_$NAME.log 2>&1
f=$(find $ROOT/gpurun_out/kt_$NAME -name "*kernel_stats.csv" | head -1)
cp "$f" $ROOT/gpurun_out/kt_$NAME.csv
cut -d, -f1-4,7 $ROOT/gpurun_out/kt_$NAME.csv | head -20
