"""Throughput of the strict drop-in entry (`nfx_compute_features_batched`: one feature set, one reference batch of f32
patchs + f32 masks per call -- what the reference's `extract_features` closure would call, main.rs:52-73), with host
tensors (the reference's CPU loader) and with device tensors.  usage: python scripts/trait_entry_bench.py [ps] [batch] [calls]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
torch.cuda.init()
nfx = importlib.import_module("nuclei-feature-extraction_amd")
synth = importlib.import_module("nuclei-feature-extraction_amd.synth")
ps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 100
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 200
ctx = nfx.Context(0)
seed = synth.SEED_BASE + 31
off, xy = synth.polygons(seed, 0, B, ps)
pu8 = synth.patches(seed, 0, B, ps)
cen, ctr = ctx.preprocess_polygons(off, xy)
masks = ctx.rasterize_polygons(off, ctr, ps)
patchs = pu8.astype(np.float32) / np.float32(255.0)
d = torch.device("cuda", 0)
t_off, t_ctr, t_p, t_m = (torch.from_numpy(np.ascontiguousarray(a)).to(d) for a in (off.astype(np.int32), ctr, patchs, masks))
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for s in range(5):
    name = nfx.feature_set_cli_name(s)
    for label, args in (("host", (off, ctr, patchs, masks)), ("device", (t_off, t_ctr, t_p, t_m))):
        for _ in range(5):
            ctx.compute_features_batched(s, *args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(calls):
            ctx.compute_features_batched(s, *args)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / calls
        print(f"{name:9s} {label:6s} ps={ps} batch={B}: {dt * 1e6:8.1f} us per call, {B / dt / 1e3:8.1f} k nuclei/s", flush=True)

# ---- the reference's calling pattern: several worker threads, one context each, host tensors ----------------------
import threading
for s in (2, 4):
    name = nfx.feature_set_cli_name(s)
    for T in (1, 2, 4, 8, 16):
        ctxs = [nfx.Context(0) for _ in range(T)]
        for c in ctxs:
            c.compute_features_batched(s, off, ctr, patchs, masks)
        per = max(20, calls // 2)
        def work(c):
            for _ in range(per):
                c.compute_features_batched(s, off, ctr, patchs, masks)
        th = [threading.Thread(target=work, args=(c,)) for c in ctxs]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        dt = time.perf_counter() - t0
        print(f"{name:9s} host, {T:2d} threads x {per} calls: {T * per * B / dt / 1e3:8.1f} k nuclei/s aggregate", flush=True)
        for c in ctxs: c.close()

# ---- one reference batch through the fast entry (u8 patches, all five sets), eager launches vs one HIP graph ---------
ctx = nfx.Context(0)
d_off_i = torch.from_numpy(off.astype(np.int32)).to(d); d_xy_raw = torch.from_numpy(xy).to(d); d_pu8 = torch.from_numpy(pu8).to(d)
out = torch.empty((B, 418), dtype=torch.float32, device=d)
all_sets = [0, 1, 2, 3, 4]
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for _ in range(3):
    ctx.extract(all_sets, d_off_i, d_xy_raw, d_pu8, ps, batch_size=B, out=out, want_centroids=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(calls):
    ctx.extract(all_sets, d_off_i, d_xy_raw, d_pu8, ps, batch_size=B, out=out, want_centroids=False)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / calls
print(f"fast entry, all sets, batch {B}, eager : {dt * 1e6:8.1f} us per call, {B / dt / 1e3:8.1f} k nuclei/s", flush=True)
g = torch.cuda.CUDAGraph(); side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    ctx.set_stream(side.cuda_stream)
    with torch.cuda.graph(g, stream=side):
        ctx.extract(all_sets, d_off_i, d_xy_raw, d_pu8, ps, batch_size=B, out=out, want_centroids=False)
torch.cuda.current_stream().wait_stream(side)
for _ in range(3): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(calls): g.replay()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / calls
print(f"fast entry, all sets, batch {B}, graph : {dt * 1e6:8.1f} us per call, {B / dt / 1e3:8.1f} k nuclei/s", flush=True)
