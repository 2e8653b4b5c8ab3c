#!/usr/bin/env python3
"""k_gabor beside k_glcm / k_glrlm of the same call (run on the GPU box; one process per variant, the knobs are read once):
    NFX_GABOR_SIDE=1 NFX_GABOR_PAD=8192 python scripts/cosched_exp.py [nuclei] [ps] [sets]
prints the resident step time (rows in HBM) and the per-kernel HIP-event times of the full feature set."""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
nfx = importlib.import_module("nuclei-feature-extraction_amd")
synth = importlib.import_module("nuclei-feature-extraction_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600_000
ps = int(sys.argv[2]) if len(sys.argv) > 2 else 64
names = sys.argv[3].split(",") if len(sys.argv) > 3 else ["all"]
dev = torch.device("cuda", 0)
torch.cuda.init()
ctx = nfx.Context(0, stream=torch.cuda.current_stream().cuda_stream)
seed = synth.SEED_BASE + 4
sets = nfx.parse_feature_sets(names)
off, xy = synth.polygons(seed, 0, n, ps)
d_off, d_xy = torch.from_numpy(off.astype(np.int32)).to(dev), torch.from_numpy(xy).to(dev)
d_p = synth.patches(seed, 0, n, ps, xp=torch, device=dev)
F = sum(nfx.num_columns(s) for s in sets)
out = torch.empty((n, F), dtype=torch.float32, device=dev)
ctx.reserve(sets, n, len(xy), ps)
for _ in range(2):
    ctx.extract(sets, d_off, d_xy, d_p, ps, out=out, want_centroids=False)
torch.cuda.synchronize()
ts = []
for _ in range(4):
    t0 = time.perf_counter()
    ctx.extract(sets, d_off, d_xy, d_p, ps, out=out, want_centroids=False)
    torch.cuda.synchronize()
    ts.append(1e3 * (time.perf_counter() - t0))
ctx.enable_timing(True); ctx.reset_timing()
ctx.extract(sets, d_off, d_xy, d_p, ps, out=out, want_centroids=False)
kt = ctx.kernel_timing()
chk = float(out.nan_to_num().double().sum().item())
print(f"SIDE={os.environ.get('NFX_GABOR_SIDE', '0')} PAD={os.environ.get('NFX_GABOR_PAD', '0')}: {n} nuclei ps {ps} {','.join(names)}: step {min(ts):.2f} ms (runs {', '.join(f'{t:.1f}' for t in ts)}); "
      + ", ".join(f"{k} {v['total_ms']:.2f}" for k, v in kt.items() if v["launches"]) + f"; checksum {chk:.6e}", flush=True)
ctx.set_stream(0)
ctx.close()
