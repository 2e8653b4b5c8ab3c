#!/bin/bash
# Regenerates the judged evidence for one round on the GPU box (everything lands in gpurun_out/prof/; copy into profiles/):
#   <tag>_bench_default.json / <tag>_config2_bench.json   bench.py lines (default incl. `also`; configs[1])
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
mkdir -p $OUT
cd $ROOT
cd /tmp && export TMPDIR=/tmp
SQ="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY"
for WL in config2 config3 config5; do
  ARGS="--workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-also"
  rocprofv3 --kernel-trace --stats -d $OUT/kt_$WL --output-format csv -- python3 $ROOT/bench.py $ARGS > $OUT/kt_$WL.log 2>&1
  cp "$(find $OUT/kt_$WL -name '*kernel_stats.csv' | head -1)" $OUT/${TAG}_${WL}_kernel_stats_full.csv
  # counters only on this engine's kernels: the data generator's thousands of torch dispatches run unprofiled
  rocprofv3 --pmc FETCH_SIZE --kernel-include-regex nfx -d $OUT/pmc_${WL}_fetch --output-format csv -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_${WL}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-include-regex nfx -d $OUT/pmc_${WL}_write --output-format csv -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_${WL}_write.log 2>&1
  rocprofv3 --pmc $SQ --kernel-include-regex nfx -d $OUT/pmc_${WL}_sq --output-format csv -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_${WL}_sq.log 2>&1
  echo "profiled $WL"
done
python3 - <<PY
import collections, csv, glob, json, sys
sys.path.insert(0, "$ROOT")
import bench
out, tag = "$OUT", "$TAG"
NUCLEI = {"config2": 100_000, "config3": 1_000_000, "config5": 1_250_000}
DOM = {"config2": "raster_shape", "config3": "glcm", "config5": "gabor"}
LAUNCH_SETS = 3  # warmup 1 + steps 2: every kernel ran three times per class
doc = {"csrc_stamp": bench.csrc_stamp(),
       "source": f"rocprofv3 --pmc (separate passes: FETCH_SIZE | WRITE_SIZE | SQ_*) of bench.py --workload W --steps 2 --warmup 1, "
                 f"scripts/profile_round.sh {tag}; traffic = (2*FETCH_SIZE + WRITE_SIZE) KB summed over every launch of the dominant kernel "
                 "in one step (FETCH_SIZE doubled per MI355X_MICROARCH.md: gfx950 counts half of wide streaming reads; narrower reads are "
                 "uncalibrated), valu = SQ_INSTS_VALU per nucleus"}
for wl in ("config2", "config3", "config5"):
    rows = [r for r in csv.DictReader(open(f"{out}/{tag}_{wl}_kernel_stats_full.csv")) if "nfx" in r["Name"]]
    with open(f"{out}/{tag}_{wl}_kernel_stats.csv", "w") as f:
        f.write("kernel,calls,total_ns,average_ns,min_ns,max_ns\n")
        for r in rows:
            f.write(f"\"{r['Name'].split('(')[0]}\",{r['Calls']},{r['TotalDurationNs']},{r['AverageNs']},{r['MinNs']},{r['MaxNs']}\n")
    acc = collections.defaultdict(lambda: [0.0, 0])   # (counter, kernel) -> [sum, dispatches]
    for sub in ("fetch", "write", "sq"):
        for fn in glob.glob(f"{out}/pmc_{wl}_{sub}/**/*counter_collection.csv", recursive=True):
            per = collections.defaultdict(float)
            for r in csv.DictReader(open(fn)):
                per[(r["Counter_Name"], r["Dispatch_Id"], r["Kernel_Name"].split("(")[0])] += float(r["Counter_Value"])
            for (c, d, k), v in per.items():
                acc[(c, k)][0] += v
                acc[(c, k)][1] += 1
    with open(f"{out}/{tag}_{wl}_pmc.csv", "w") as f:
        f.write("counter,kernel,dispatches,mean_per_dispatch,per_nucleus_per_step\n")
        for (c, k), (v, n) in sorted(acc.items()):
            f.write(f"{c},\"{k}\",{n},{v / n:.6g},{v / LAUNCH_SETS / NUCLEI[wl]:.6g}\n")
    dom = DOM[wl]
    tot = lambda c: sum(v for (cc, k), (v, n) in acc.items() if cc == c and ("k_" + dom) in k) / LAUNCH_SETS
    doc[wl] = {"kernel": "k_" + dom, "traffic_bytes_per_launch": int((2 * tot("FETCH_SIZE") + tot("WRITE_SIZE")) * 1024),
               "valu_insts_per_nucleus": round(tot("SQ_INSTS_VALU") / NUCLEI[wl], 1),
               "lds_idx_active_cycles_per_nucleus": round(tot("SQ_LDS_IDX_ACTIVE") / NUCLEI[wl], 1),
               "lds_addr_conflict_over_idx_active": round(tot("SQ_LDS_ADDR_CONFLICT") / max(1.0, tot("SQ_LDS_IDX_ACTIVE")), 4),
               "lds_bank_conflict_over_idx_active": round(tot("SQ_LDS_BANK_CONFLICT") / max(1.0, tot("SQ_LDS_IDX_ACTIVE")), 4)}
json.dump(doc, open(f"{out}/pmc_current.json", "w"), indent=1)
print(json.dumps(doc, indent=1))
PY
# the bench lines last: they read the pmc_current.json just written (same csrc stamp -> issue / lds / traffic filled in)
cp $OUT/pmc_current.json $ROOT/profiles/pmc_current.json
cd $ROOT
python3 bench.py --steps 5 --warmup 2 > $OUT/${TAG}_bench_default.json 2> $OUT/bench_default.err
python3 bench.py --workload config2 --steps 20 --warmup 3 > $OUT/${TAG}_config2_bench.json 2> $OUT/bench2.err
tail -c 400 $OUT/${TAG}_bench_default.json; echo; tail -c 300 $OUT/${TAG}_config2_bench.json
