#!/bin/bash
# Gabor phase costs of every scratch_exp/*.so built with -DNFX_EXPERIMENTS (the product build ignores NFX_GABOR_DBG):
# NFX_GABOR_DBG 1 = skip row passes, 2 = skip column passes, 4 = skip all groups (timing only; results are wrong)
#   bash scripts/build_variant.sh gabor_exp k_gabor.hip -DNFX_EXPERIMENTS && bash scripts/gabor_dbg.sh [nuclei]
N=${1:-300000}
for lib in scratch_exp/*.so; do
  for d in 0 1 2 4; do
    echo -n "$lib dbg=$d: "
    NFX_LIB_PATH=$PWD/$lib NFX_GABOR_DBG=$d timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --workload config5 --inputs resident --nuclei $N --steps 3 --warmup 1 --sets gabor | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernels_ms_per_step']['k_gabor'])" || exit 1
  done
done
