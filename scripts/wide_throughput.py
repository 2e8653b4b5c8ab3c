#!/usr/bin/env python3
"""Throughput of the wide (65..128) and huge (129..512) kernel families per feature set, device-resident inputs (run on the GPU box).
    python scripts/wide_throughput.py [ps] [nuclei]"""
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
ps = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
dev = torch.device("cuda", 0)
seed = synth.SEED_BASE + 90
off, xy = synth.polygons(seed, 0, n, ps)
d_off = torch.from_numpy(off.astype(np.int32)).to(dev)
d_xy = torch.from_numpy(xy).to(dev)
d_p = synth.patches(seed, 0, n, ps, xp=torch, device=dev)
ctx = nfx.Context(0, stream=torch.cuda.current_stream().cuda_stream)
for name in ("geometry", "color", "glcm", "glrlm", "gabor", "all"):
    sets = nfx.parse_feature_sets([name])
    F = sum(nfx.num_columns(s) for s in sets)
    out = torch.empty((n, F), dtype=torch.float32, device=dev)
    ctx.extract(sets, d_off, d_xy, d_p, ps, out=out, want_centroids=False)
    torch.cuda.synchronize()
    t = time.perf_counter()
    reps = 3
    for _ in range(reps):
        ctx.extract(sets, d_off, d_xy, d_p, ps, out=out, want_centroids=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    print(f"ps {ps:3d}  {name:9s} {n / dt:12.0f} nuclei/s  ({dt * 1e3:8.2f} ms for {n} nuclei)", flush=True)
ctx.close()
