#!/bin/bash
# PMC passes over the GLCM kernels (run on the GPU box): scripts/pmc_glcm.sh <workload> <nuclei>
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
WL=${1:-config3}; N=${2:-100000}
cd /tmp && export TMPDIR=/tmp
i=0
for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_INSTS_SMEM" "SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --pmc $CTRS -d $ROOT/gpurun_out/pmc_$WL/p$i --output-format csv -- python3 $ROOT/bench.py --workload $WL --nuclei $N --sets ${SET:-glcm} --steps 1 --warmup 0 --no-cpu-baseline > $ROOT/gpurun_out/pmc_$WL/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$ROOT/gpurun_out/pmc_$WL/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "${KF:-glcm}" in k: tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
with open("$ROOT/gpurun_out/pmc_$WL/summary.txt", "w") as o:
    for k, d in tot.items():
        o.write(k + "\n")
        for c, v in sorted(d.items()): o.write(f"  {c:28s} {v:.4g}  per-nucleus {v/$N:.1f}\n")
print(open("$ROOT/gpurun_out/pmc_$WL/summary.txt").read())
PY
