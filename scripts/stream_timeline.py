#!/usr/bin/env python3
"""One streamed call under the tracer (run on the GPU box), for a timeline of its kernels and copies:
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/tl -- python3 scripts/stream_timeline.py [nuclei] [ps] [chunk]
    python scripts/stream_timeline.py --read gpurun_out/tl      # afterwards, anywhere: busy time and gaps per chunk"""
import glob
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def read(d):
    import csv
    ev = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"].split("(")[0][:44], r.get("Queue_Id", "")))
    for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r["Direction"], ""))
    ev.sort()
    # the LAST call: the calls are separated by a host-side pause of 0.2 s, so walk back from the end until an idle stretch of > 50 ms
    t_end = max(e[1] for e in ev)
    cut, nxt = ev[0][0], None
    for e in sorted(ev, key=lambda x: -x[0]):
        if nxt is not None and nxt - e[1] > 50_000_000 and e[1] < nxt:
            cut = nxt
            break
        nxt = e[0] if nxt is None else min(nxt, e[0])
    ev = [e for e in ev if e[0] >= cut]
    t0 = min(e[0] for e in ev)
    print(f"{len(ev)} events over {(t_end - t0) / 1e6:.1f} ms")

    def busy(sel):
        iv = sorted((e[0], e[1]) for e in ev if sel(e))
        tot, cur_s, cur_e = 0, None, None
        for s, e in iv:
            if cur_e is None or s > cur_e:
                if cur_e is not None: tot += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        if cur_e is not None: tot += cur_e - cur_s
        return tot / 1e6
    feat = lambda e: e[2] == "K" and "gather_patches" not in e[3]
    print(f"busy: feature kernels {busy(feat):.1f} ms, gathers {busy(lambda e: 'gather_patches' in e[3]):.1f} ms, "
          f"H2D {busy(lambda e: e[2] == 'C' and 'HOST_TO_DEVICE' in e[3]):.1f} ms, D2H {busy(lambda e: e[2] == 'C' and 'DEVICE_TO_HOST' in e[3]):.1f} ms")
    by = {}
    for e in ev:
        if e[2] == "K":
            a = by.setdefault(e[3], [0, 0.0]); a[0] += 1; a[1] += (e[1] - e[0]) / 1e6
    for k, (c, ms) in sorted(by.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get('TL_TOP', '14'))]:
        print(f"  {k:46s} x{c:4d} {ms:8.2f} ms")
    # what runs beside what: wall time by the set of HEAVY kernels (k_gabor / k_glcm classes / k_glrlm) in flight -- the chunks' kernels run two at a
    # time on two streams, so "k_gabor+k_glcm" is time in which the two share the chip, "k_gabor" time in which it has the chip to itself
    heavy = lambda nm: nm.startswith("void nfx::k_gabor") or nm.startswith("void nfx::k_glcm<") or nm.startswith("void nfx::k_glrlm")
    pts = []
    for e in ev:
        if e[2] == "K":
            pts.append((e[0], 1, e[3])); pts.append((e[1], -1, e[3]))
    pts.sort()
    active, combo, last = {}, {}, t0
    for t, d, nm in pts:
        if t > last:
            names = [k for k, v in active.items() if v > 0]
            h = sorted(set(k.replace("void nfx::", "").split("<")[0] for k in names if heavy(k)))
            key = "+".join(h) if h else ("light kernels only" if names else "no kernel")
            combo[key] = combo.get(key, 0) + (t - last)
        active[nm] = active.get(nm, 0) + d
        last = t
    print("wall time by heavy kernels in flight: " + ", ".join(f"{k} {v / 1e6:.1f} ms" for k, v in sorted(combo.items(), key=lambda kv: -kv[1])))
    # gaps in the feature kernels' union (> 0.2 ms): what the compute stream waited for
    iv = sorted((e[0], e[1], e[3]) for e in ev if feat(e))
    end, last = iv[0][1], iv[0][2]
    for s, e, nm in iv[1:]:
        if s - end > 200_000:
            print(f"  gap {(s - end) / 1e6:6.2f} ms at {(end - t0) / 1e6:7.1f} ms after {last} before {nm}")
        if e > end: end, last = e, nm


if len(sys.argv) > 2 and sys.argv[1] == "--read":
    read(sys.argv[2])
    sys.exit(0)

import torch
sys.path.insert(0, ROOT)
nfx = importlib.import_module("nuclei-feature-extraction_amd")
synth = importlib.import_module("nuclei-feature-extraction_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
ps = int(sys.argv[2]) if len(sys.argv) > 2 else 64
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 0
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
del tile
torch.cuda.empty_cache()
sets = [0, 1, 2, 3, 4]
rows = nfx.pinned_empty((n, sum(nfx.num_columns(s) for s in sets)), np.float32)
import time
if os.environ.get("TL_MODE") == "resident":   # the same shard with the slide resident in HBM: one launch per kernel (what the chunks' launches are compared with)
    d_tile = torch.from_numpy(h_tile).to(dev)
    ctx.reserve(sets, n, len(xy), ps)
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.extract_tile(sets, d_off, d_xy, d_tile, ps)
        torch.cuda.synchronize()
        print(f"resident-tile call {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
        time.sleep(0.2)
else:
    for _ in range(2):
        t0 = time.perf_counter()
        ctx.extract_tile_streamed(sets, off, xy, h_tile, ps, chunk_nuclei=chunk, out=rows)
        print(f"streamed call {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
        time.sleep(0.2)
ctx.set_stream(0)
ctx.close()
