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
for WL in config3; do   # config5 under --pmc spends minutes in the data generator's thousands of small dispatches
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C -d $OUT/pmc_${WL}_$C --output-format csv -- python3 $ROOT/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_${WL}_$C.log 2>&1
  done
done
python3 - <<PY
import csv, glob, json, collections
out = "$OUT"; tag = "$TAG"
# kernel stats: nfx kernels only
rows = [r for r in csv.DictReader(open(f"{out}/{tag}_config3_kernel_stats_full.csv")) if "nfx" in r["Name"]]
with open(f"{out}/{tag}_config3_kernel_stats.csv", "w") as f:
    f.write("kernel,calls,total_ns,average_ns,min_ns,max_ns\n")
    for r in rows: f.write(f"\"{r['Name'].split('(')[0]}\",{r['Calls']},{r['TotalDurationNs']},{r['AverageNs']},{r['MinNs']},{r['MaxNs']}\n")
steps = 3  # warmup 1 + steps 2
traffic = {}
for wl, dom in (("config3", "glcm"),):
    # PMC passes: mean value per launch per kernel
    acc = collections.defaultdict(lambda: [0.0, 0])
    for C in ("FETCH_SIZE", "WRITE_SIZE"):
        for fn in glob.glob(f"{out}/pmc_{wl}_{C}/**/*counter_collection.csv", recursive=True):
            per_dispatch = collections.defaultdict(float)
            for r in csv.DictReader(open(fn)):
                if r["Counter_Name"] == C and "nfx" in r["Kernel_Name"]:
                    per_dispatch[(r["Dispatch_Id"], r["Kernel_Name"].split("(")[0])] += float(r["Counter_Value"])
            for (d, k), v in per_dispatch.items():
                acc[(C, k)][0] += v; acc[(C, k)][1] += 1
    with open(f"{out}/{tag}_{wl}_pmc_fetch_write.csv", "w") as f:
        f.write("counter,kernel,launches,\"mean_value_per_launch (KB; rocprofv3 --pmc, separate passes)\"\n")
        for (C, k), (v, n) in sorted(acc.items()): f.write(f"{C},\"{k}\",{n},{v / n}\n")
    # the dominant kernel's traffic per step (every launch that belongs to it); FETCH doubled per the gfx950 note
    fetch = sum(v for (C, k), (v, n) in acc.items() if C == "FETCH_SIZE" and dom in k) / steps
    write = sum(v for (C, k), (v, n) in acc.items() if C == "WRITE_SIZE" and dom in k) / steps
    traffic[wl] = {"kernel": "k_" + dom, "traffic_bytes_per_launch": int((2 * fetch + write) * 1024),
        "source": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes), profiles/{tag}_{wl}_pmc_fetch_write.csv; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 summed over every k_{dom}* launch of one step; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts half of 16-B/lane streams; narrower reads are uncalibrated)"}
json.dump(traffic, open(f"{out}/{tag}_pmc_traffic.json", "w"), indent=1)
print(open(f"{out}/{tag}_config3_kernel_stats.csv").read()); print(json.dumps(traffic)[:600])
PY
tail -n 1 $OUT/${TAG}_config3_bench.json $OUT/${TAG}_config5_bench.json $OUT/${TAG}_config2_bench.json
