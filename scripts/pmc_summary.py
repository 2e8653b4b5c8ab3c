#!/usr/bin/env python3
"""Turns one scripts/profile_round.sh run (rocprofv3 csv files under <out>/) into the committed evidence:
  <tag>_<workload>_kernel_stats.csv   this engine's kernels out of rocprofv3 --kernel-trace --stats
  <tag>_<workload>_pmc.csv            per-kernel means of FETCH_SIZE, WRITE_SIZE and the SQ counters (separate --pmc passes)
  pmc_current.json                    what bench.py reads: per workload and timing slot (k_glcm, k_gabor, ...) the traffic per step, VALU
                                      instructions and LDS-active cycles per nucleus, stamped with the sha of csrc/ they were taken on
usage: python scripts/pmc_summary.py gpurun_out/prof r03"""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

out, tag = sys.argv[1], sys.argv[2]
NUCLEI = {"config2": 100_000, "config3": 1_000_000, "config5": 1_250_000}
SLOTS = ("k_preprocess", "k_raster_shape", "k_polygon", "k_pack_masks", "k_gather_patches", "k_hue_batch", "k_color", "k_glcm", "k_glrlm", "k_gabor")
LAUNCH_SETS = 3  # warmup 1 + steps 2: every kernel ran three times


def slot_of(kernel):
    hits = [s for s in SLOTS if s in kernel]
    return max(hits, key=len) if hits else None


doc = {"csrc_stamp": bench.csrc_stamp(),
       "source": f"rocprofv3 --pmc (separate passes: FETCH_SIZE | WRITE_SIZE | SQ_*) of bench.py --workload W --steps 2 --warmup 1, "
                 f"scripts/profile_round.sh {tag}; traffic = (2*FETCH_SIZE + WRITE_SIZE) KB summed over every launch of a timing slot's kernels "
                 "in one step (FETCH_SIZE doubled per MI355X_MICROARCH.md: gfx950 counts half of wide streaming reads; narrower reads are "
                 "uncalibrated), valu = SQ_INSTS_VALU per nucleus, lds = SQ_LDS_IDX_ACTIVE cycles per nucleus"}
for wl in ("config2", "config3", "config5"):
    full = f"{out}/{tag}_{wl}_kernel_stats_full.csv"
    if not os.path.exists(full):
        continue
    rows = [r for r in csv.DictReader(open(full)) if "nfx" in r["Name"]]
    time_ns = collections.defaultdict(float)
    with open(f"{out}/{tag}_{wl}_kernel_stats.csv", "w") as f:
        f.write("kernel,calls,total_ns,average_ns,min_ns,max_ns\n")
        for r in rows:
            f.write(f"\"{r['Name'].split('(')[0]}\",{r['Calls']},{r['TotalDurationNs']},{r['AverageNs']},{r['MinNs']},{r['MaxNs']}\n")
            s = slot_of(r["Name"])
            if s:
                time_ns[s] += float(r["TotalDurationNs"])
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
    slots = {}
    for s in SLOTS:
        tot = lambda c: sum(v for (cc, k), (v, n) in acc.items() if cc == c and slot_of(k) == s) / LAUNCH_SETS
        if tot("SQ_INSTS_VALU") == 0:
            continue
        slots[s] = {"ms_per_step_under_the_tracer": round(time_ns[s] / LAUNCH_SETS / 1e6, 4),
                    "traffic_bytes_per_launch": int((2 * tot("FETCH_SIZE") + tot("WRITE_SIZE")) * 1024),
                    "valu_insts_per_nucleus": round(tot("SQ_INSTS_VALU") / NUCLEI[wl], 1),
                    "lds_idx_active_cycles_per_nucleus": round(tot("SQ_LDS_IDX_ACTIVE") / NUCLEI[wl], 1),
                    "lds_addr_conflict_over_idx_active": round(tot("SQ_LDS_ADDR_CONFLICT") / max(1.0, tot("SQ_LDS_IDX_ACTIVE")), 4),
                    "lds_bank_conflict_over_idx_active": round(tot("SQ_LDS_BANK_CONFLICT") / max(1.0, tot("SQ_LDS_IDX_ACTIVE")), 4),
                    "wait_any_over_wave_cycles": round(tot("SQ_WAIT_ANY") / max(1.0, tot("SQ_WAVE_CYCLES")), 4)}
    dom = max(slots, key=lambda s: slots[s]["ms_per_step_under_the_tracer"]) if slots else None
    doc[wl] = {"kernel": dom, "slots": slots}
json.dump(doc, open(f"{out}/pmc_current.json", "w"), indent=1)
print(json.dumps(doc, indent=1)[:3000])
