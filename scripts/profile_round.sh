#!/bin/bash
# Regenerates the judged evidence for one round on the GPU box (everything lands in gpurun_out/prof/):
#   bench JSON lines (config3 default with cpu_baseline, config5, config2), rocprofv3 kernel-trace stats of the
#   default bench command, and the FETCH_SIZE / WRITE_SIZE PMC passes (separate, as MI355X_MICROARCH.md prescribes).
# usage: bash scripts/profile_round.sh r01
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd $ROOT
python3 bench.py --steps 3 --warmup 1 > $OUT/${TAG}_config3_bench.json 2> $OUT/bench3.err
python3 bench.py --workload config5 --steps 2 --warmup 1 > $OUT/${TAG}_config5_bench.json 2> $OUT/bench5.err
python3 bench.py --workload config2 --steps 20 --warmup 3 > $OUT/${TAG}_config2_bench.json 2> $OUT/bench2.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/kt.log 2>&1
cp "$(find $OUT/kt -name '*kernel_stats.csv' | head -1)" $OUT/${TAG}_config3_kernel_stats_full.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C -d $OUT/pmc_$C --output-format csv -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_$C.log 2>&1
done
python3 - <<PY
import csv, glob, json, collections
out = "$OUT"; tag = "$TAG"
# kernel stats: nfx kernels only
rows = [r for r in csv.DictReader(open(f"{out}/{tag}_config3_kernel_stats_full.csv")) if "nfx" in r["Name"]]
with open(f"{out}/{tag}_config3_kernel_stats.csv", "w") as f:
    f.write("kernel,calls,total_ns,average_ns,min_ns,max_ns\n")
    for r in rows: f.write(f"\"{r['Name'].split('(')[0]}\",{r['Calls']},{r['TotalDurationNs']},{r['AverageNs']},{r['MinNs']},{r['MaxNs']}\n")
# PMC passes: mean value per launch per kernel
acc = collections.defaultdict(lambda: [0.0, 0])
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    for fn in glob.glob(f"{out}/pmc_{C}/**/*counter_collection.csv", recursive=True):
        per_dispatch = collections.defaultdict(float)
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] == C and "nfx" in r["Kernel_Name"]:
                per_dispatch[(r["Dispatch_Id"], r["Kernel_Name"].split("(")[0])] += float(r["Counter_Value"])
        for (d, k), v in per_dispatch.items():
            acc[(C, k)][0] += v; acc[(C, k)][1] += 1
with open(f"{out}/{tag}_config3_pmc_fetch_write.csv", "w") as f:
    f.write("counter,kernel,launches,\"mean_value_per_launch (KB; rocprofv3 --pmc, separate passes)\"\n")
    for (C, k), (v, n) in sorted(acc.items()): f.write(f"{C},\"{k}\",{n},{v / n}\n")
# GLCM traffic per step: classify + every size class + finish; FETCH doubled per the gfx950 note
steps = 3  # warmup 1 + steps 2
fetch = sum(v for (C, k), (v, n) in acc.items() if C == "FETCH_SIZE" and "glcm" in k) / steps
write = sum(v for (C, k), (v, n) in acc.items() if C == "WRITE_SIZE" and "glcm" in k) / steps
traffic = {"config3": {"kernel": "k_glcm", "traffic_bytes_per_launch": int((2 * fetch + write) * 1024),
    "source": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes), profiles/{tag}_config3_pmc_fetch_write.csv; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 summed over the k_glcm_classify + k_glcm<10..13> + k_glcm_finish launches of one step; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts half of 16-B/lane streams; the 8-B/lane mask-row and table reads are uncalibrated)"}}
json.dump(traffic, open(f"{out}/{tag}_pmc_traffic.json", "w"), indent=1)
print(open(f"{out}/{tag}_config3_kernel_stats.csv").read()); print(json.dumps(traffic)[:300])
PY
tail -n 1 $OUT/${TAG}_config3_bench.json $OUT/${TAG}_config5_bench.json $OUT/${TAG}_config2_bench.json
