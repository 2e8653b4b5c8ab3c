#!/usr/bin/env python3
"""Where the streamed-tile path spends its time (run on the GPU box): the same slide and nuclei through nfx_extract_tile_streamed with
cheap and with full feature sets and several chunk sizes, against the resident call and the link's plain copy rate.
    python scripts/stream_probe.py [nuclei] [ps] > gpurun_out/stream_probe.txt"""
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
ps = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda", 0)
torch.cuda.init()
ctx = nfx.Context(0, stream=torch.cuda.current_stream().cuda_stream)
seed = synth.SEED_BASE + 4
T = synth.tile_side(n, ps)
cx, cy = synth.centres_slide_order(seed, 0, n, n, ps, T, T)
off, xy = synth.polygons(seed, 0, n, ps, cx, cy)
d_off, d_xy = torch.from_numpy(off.astype(np.int32)).to(dev), torch.from_numpy(xy).to(dev)
cen = ctx.preprocess_polygons(d_off, d_xy)[0].cpu().numpy()
d_p = synth.patches(seed, 0, n, ps, xp=torch, device=dev)
tile = synth.tile_from_patches(torch, d_p, cen[:, 0], cen[:, 1], T, T, ps, seed, dev)
del d_p
h_tile = nfx.pinned_empty((3, T, T), np.uint8)
for r0 in range(0, T, 2048):
    h_tile[:, r0:r0 + 2048] = tile[:, r0:r0 + 2048].cpu().numpy()
print(f"{n} nuclei, ps {ps}, slide {T} x {T} = {3 * T * T / 1e9:.2f} GB", flush=True)
# the link alone: one plain copy of the whole slide out of the same pinned memory, through libnfx's runtime (gather of zero nuclei is not
# possible, so: the streamed call on geometry + colour of 100 nuclei spread over every band -> every band travels, nothing else happens)
for name, sets in (("geometry+color", [0, 1]), ("geometry+color+glcm", [0, 1, 2]), ("all", [0, 1, 2, 3, 4])):
    F = sum(nfx.num_columns(s) for s in sets)
    rows = nfx.pinned_empty((n, F), np.float32)
    d_rows = torch.empty((n, F), dtype=torch.float32, device=dev)
    ctx.reserve(sets, n, len(xy), ps)
    ctx.extract_tile(sets, d_off, d_xy, tile, ps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.extract_tile(sets, d_off, d_xy, tile, ps)
    torch.cuda.synchronize()
    t_res = time.perf_counter() - t0
    for chunk in (50_000, 100_000, 200_000, 400_000):
        ctx.extract_tile_streamed(sets, off, xy, h_tile, ps, chunk_nuclei=chunk, out=rows)
        t0 = time.perf_counter()
        _, _, st = ctx.extract_tile_streamed(sets, off, xy, h_tile, ps, chunk_nuclei=chunk, out=rows)
        dt = time.perf_counter() - t0
        wire = st["image_bytes_h2d"] + st["other_bytes_h2d"] + st["rows_bytes_d2h"]
        print(f"  {name:22s} chunk {chunk:7d}: streamed {dt * 1e3:7.1f} ms ({n / dt / 1e6:5.2f} M/s), resident-tile call {t_res * 1e3:7.1f} ms; "
              f"H2D {st['image_bytes_h2d'] / 1e9:.2f} GB = {st['image_bytes_h2d'] / dt / 1e9:5.1f} GB/s if it were alone, wire {wire / dt / 1e9:5.1f} GB/s; "
              f"chunks {st['chunks']}, bands copied {st['blocks_copied']}/{st['blocks_total']}, pinned {st['image_pinned']}", flush=True)
ctx.set_stream(0)
ctx.close()
