#!/usr/bin/env python3
"""Resident-tile entry (nfx_extract_tile, everything in HBM): windows read in place against windows copied into packed patches
(run on the GPU box; one process per variant: NFX_TILE_WINDOWS is read once).
    NFX_TILE_WINDOWS=0|1 python scripts/tile_windows_ab.py [nuclei] [ps] [sets]"""
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
ps = int(sys.argv[2]) if len(sys.argv) > 2 else 64
names = sys.argv[3].split(",") if len(sys.argv) > 3 else ["all"]
dev = torch.device("cuda", 0)
torch.cuda.init()
ctx = nfx.Context(0, stream=torch.cuda.current_stream().cuda_stream)
seed = synth.SEED_BASE + 4
sets = nfx.parse_feature_sets(names)
T = synth.tile_side(n, ps)
cx, cy = synth.centres_slide_order(seed, 0, n, n, ps, T, T)
off, xy = synth.polygons(seed, 0, n, ps, cx, cy)
d_off, d_xy = torch.from_numpy(off.astype(np.int32)).to(dev), torch.from_numpy(xy).to(dev)
cen = ctx.preprocess_polygons(d_off, d_xy)[0].cpu().numpy()
d_p = synth.patches(seed, 0, n, ps, xp=torch, device=dev)
tile = synth.tile_from_patches(torch, d_p, cen[:, 0], cen[:, 1], T, T, ps, seed, dev)
del d_p
torch.cuda.empty_cache()
ctx.reserve(sets, n, len(xy), ps)
for _ in range(2):
    rows, _ = ctx.extract_tile(sets, d_off, d_xy, tile, ps)
torch.cuda.synchronize()
ts = []
for _ in range(4):
    t0 = time.perf_counter()
    rows, _ = ctx.extract_tile(sets, d_off, d_xy, tile, ps)
    torch.cuda.synchronize()
    ts.append(1e3 * (time.perf_counter() - t0))
ctx.enable_timing(True); ctx.reset_timing()
rows, _ = ctx.extract_tile(sets, d_off, d_xy, tile, ps)
kt = ctx.kernel_timing()
import hashlib
h = hashlib.sha256(rows.cpu().numpy().tobytes()).hexdigest()[:16]
print(f"NFX_TILE_WINDOWS={os.environ.get('NFX_TILE_WINDOWS', '1')}: {n} nuclei ps {ps} {','.join(names)} slide {T}^2: resident-tile call {min(ts):.2f} ms (runs {', '.join(f'{t:.1f}' for t in ts)}); "
      + ", ".join(f"{k} {v['total_ms']:.2f}" for k, v in kt.items() if v["launches"]) + f"; rows sha {h}", flush=True)
ctx.set_stream(0)
ctx.close()
