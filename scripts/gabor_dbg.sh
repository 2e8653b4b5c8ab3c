#!/bin/bash
# Gabor phase costs: NFX_GABOR_DBG 1 = skip row passes, 2 = skip column passes, 4 = skip all groups (timing only; results are wrong)
for d in 0 1 2 4; do
  echo "dbg=$d"
  NFX_GABOR_DBG=$d timeout -k 10 200 python bench.py --no-cpu-baseline --workload config5 --nuclei 300000 --steps 3 --warmup 1 --sets gabor | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernels_ms_per_step'])" || exit 1
done
