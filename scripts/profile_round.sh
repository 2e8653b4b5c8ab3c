#!/bin/bash
# Regenerates the judged evidence for one round on the GPU box (everything lands in gpurun_out/prof/; copy into profiles/):
#   <tag>_bench_default.json / <tag>_config{2,3}_bench.json   bench.py lines (default = configs[4]'s shard streamed, incl. `also`; configs[1]; configs[2])
#   <tag>_default_streamed_kernel_stats.csv                 rocprofv3 --kernel-trace --stats of the DEFAULT command (the streamed shard)
#   <tag>_config{2,3,5}_kernel_stats.csv                   rocprofv3 --kernel-trace --stats of bench.py (nfx kernels)
#   <tag>_config{2,3,5}_pmc.csv                              per-kernel means of FETCH_SIZE, WRITE_SIZE and the SQ counters
#                                                          (separate --pmc passes, as MI355X_MICROARCH.md prescribes)
#   pmc_current.json                                       what bench.py reads: traffic per launch and VALU instructions per
#                                                          nucleus of each workload's dominant kernel + the csrc/ stamp they
#                                                          were taken on (bench.py prints null when the stamp does not match)
# usage: bash scripts/profile_round.sh r03
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03}
OUT=$ROOT/gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT   # (a local gpurun_out/ keeps the files of earlier runs: never summarise a mixture)
cd $ROOT
cd /tmp && export TMPDIR=/tmp
SQ="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY"
for WL in config2 config3 config5; do
  ARGS="--workload $WL --inputs resident --steps 2 --warmup 1 --no-cpu-baseline --no-also"   # (counters are taken on resident inputs: one launch per kernel and step)
  rocprofv3 --kernel-trace --stats -d $OUT/kt_$WL --output-format csv -- python3 $ROOT/bench.py $ARGS > $OUT/kt_$WL.log 2>&1
  cp "$(find $OUT/kt_$WL -name '*kernel_stats.csv' | head -1)" $OUT/${TAG}_${WL}_kernel_stats_full.csv
  # counters only on this engine's kernels: the data generator's thousands of torch dispatches run unprofiled
  rocprofv3 --pmc FETCH_SIZE --kernel-include-regex nfx -d $OUT/pmc_${WL}_fetch --output-format csv -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_${WL}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-include-regex nfx -d $OUT/pmc_${WL}_write --output-format csv -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_${WL}_write.log 2>&1
  rocprofv3 --pmc $SQ --kernel-include-regex nfx -d $OUT/pmc_${WL}_sq --output-format csv -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_${WL}_sq.log 2>&1
  echo "profiled $WL"
done
# the DEFAULT command's own kernel trace: configs[4]'s shard with the slide streamed (13-14 chunk launches per kernel and step, two at a time)
rocprofv3 --kernel-trace -d $OUT/kt_default --output-format csv -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also > $OUT/kt_default.log 2>&1
python3 $ROOT/scripts/kt_summary.py $OUT/kt_default $OUT/${TAG}_default_streamed_kernel_stats.csv > /dev/null
python3 $ROOT/scripts/pmc_summary.py $OUT $TAG
# the bench lines last: they read the pmc_current.json just written (same csrc stamp -> issue / lds / traffic filled in)
cp $OUT/pmc_current.json $ROOT/profiles/pmc_current.json
cd $ROOT
python3 bench.py --steps 5 --warmup 2 > $OUT/${TAG}_bench_default.json 2> $OUT/bench_default.err
python3 bench.py --workload config2 --steps 20 --warmup 3 > $OUT/${TAG}_config2_bench.json 2> $OUT/bench2.err
python3 bench.py --workload config3 --steps 10 --warmup 2 --no-also > $OUT/${TAG}_config3_bench.json 2> $OUT/bench3.err
tail -c 400 $OUT/${TAG}_bench_default.json; echo; tail -c 300 $OUT/${TAG}_config2_bench.json
