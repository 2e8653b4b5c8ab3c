"""Tile-resident entry (`nfx_extract_tile`, what the command line uses) against the packed-patch entry on the same nuclei:
cost of gathering the patches on the GPU.  usage: python scripts/tile_entry_bench.py [nuclei] [ps] [tile side]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
torch.cuda.init()
nfx = importlib.import_module("nuclei-feature-extraction_amd")
synth = importlib.import_module("nuclei-feature-extraction_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ps = int(sys.argv[2]) if len(sys.argv) > 2 else 48
side = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
d = torch.device("cuda", 0)
seed = synth.SEED_BASE + 41
cx, cy = synth.centres(seed, 0, n, ps, side, side)
off, xy = synth.polygons(seed, 0, n, ps, cx, cy)
g = torch.Generator(device=d); g.manual_seed(7)
img = torch.randint(0, 256, (3, side, side), dtype=torch.uint8, device=d, generator=g)
d_off = torch.from_numpy(off.astype(np.int32)).to(d); d_xy = torch.from_numpy(xy).to(d)
ctx = nfx.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
ctx.enable_timing(True)
sets = [0, 1, 2]
def run(fn, reps=3):
    fn(); torch.cuda.synchronize(); ctx.reset_timing()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return dt, {k: round(v["total_ms"] / reps, 3) for k, v in ctx.kernel_timing().items() if v["launches"]}
rows_t, cen = ctx.extract_tile(sets, d_off, d_xy, img, ps)
patches = ctx.gather_patches(cen, img, ps)
dt, kt = run(lambda: ctx.extract_tile(sets, d_off, d_xy, img, ps))
print(f"tile entry   : {dt * 1e3:8.2f} ms  {n / dt / 1e6:6.2f} M nuclei/s  {kt}")
dt, kt = run(lambda: ctx.extract(sets, d_off, d_xy, patches, ps))
print(f"packed entry : {dt * 1e3:8.2f} ms  {n / dt / 1e6:6.2f} M nuclei/s  {kt}")
rows_p, _ = ctx.extract(sets, d_off, d_xy, patches, ps)
print("identical rows:", bool(torch.equal(torch.nan_to_num(rows_t), torch.nan_to_num(rows_p))))
