#!/usr/bin/env python3
"""The streamed-slide call in a process WITHOUT torch against the same call after `import torch` (run on the GPU box).

libnfx.so needs libamdhip64.so.7.  In a process that imported torch first that name is already bound to the runtime the torch
wheel bundles (torch/lib/libamdhip64.so, ROCm 7.0: ONE runtime in the process); in a torch-free process -- the C++ command line,
a Rust host -- it is /opt/rocm's (7.2).  This script says which one is mapped (from /proc/self/maps) and times the same call in both,
rows copied home (`copy`) and written home by the kernels (`direct`).

    python scripts/stream_ab.py gen [nuclei] [ps]            # makes the inputs once (uses torch on the GPU), into /dev/shm/nfx_ab
    python scripts/stream_ab.py run 0|1 [copy|direct] [chunk] [block]
                                                             # 0: torch never imported; 1: torch imported and initialised first
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/ab0 -- python3 scripts/stream_ab.py run 0 copy
    python scripts/stream_timeline.py --read gpurun_out/ab0  # busy times and gaps of the last call
"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHM = "/dev/shm/nfx_ab"


def hip_runtimes():
    libs = set()
    with open("/proc/self/maps") as fh:
        for ln in fh:
            if "libamdhip64" in ln:
                libs.add(ln.split()[-1])
    return sorted(libs)


def gen(n, ps):
    import torch
    nfx = importlib.import_module("nuclei-feature-extraction_amd")
    synth = importlib.import_module("nuclei-feature-extraction_amd.synth")
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
    sets = [0, 1, 2, 3, 4]
    ctx.reserve(sets, n, len(xy), ps)
    ref, _ = ctx.extract_tile(sets, d_off, d_xy, tile, ps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.extract_tile(sets, d_off, d_xy, tile, ps)
    torch.cuda.synchronize()
    t_res = time.perf_counter() - t0
    os.makedirs(SHM, exist_ok=True)
    np.save(os.path.join(SHM, "off.npy"), off)
    np.save(os.path.join(SHM, "xy.npy"), xy)
    np.save(os.path.join(SHM, "ref.npy"), ref.cpu().numpy())
    mm = np.lib.format.open_memmap(os.path.join(SHM, "tile.npy"), mode="w+", dtype=np.uint8, shape=(3, T, T))
    for r0 in range(0, T, 2048):
        mm[:, r0:r0 + 2048] = tile[:, r0:r0 + 2048].cpu().numpy()
    mm.flush()
    with open(os.path.join(SHM, "meta.txt"), "w") as fh:
        fh.write(f"{n} {ps} {T} {t_res * 1e3:.2f}\n")
    print(f"gen: {n} nuclei, ps {ps}, slide {T} x {T} = {3 * T * T / 1e9:.2f} GB; resident-tile call {t_res * 1e3:.1f} ms; runtimes {hip_runtimes()}", flush=True)
    ctx.set_stream(0)
    ctx.close()


def run(with_torch, rows_mode, chunk, block, reps=3):
    if with_torch:
        import torch
        torch.cuda.init()
    nfx = importlib.import_module("nuclei-feature-extraction_amd")
    n, ps, T, t_res = open(os.path.join(SHM, "meta.txt")).read().split()
    n, ps, T = int(n), int(ps), int(T)
    off, xy = np.load(os.path.join(SHM, "off.npy")), np.load(os.path.join(SHM, "xy.npy"))
    ref = np.load(os.path.join(SHM, "ref.npy"), mmap_mode="r")
    ctx = nfx.Context(0)
    ctx.set_option("stream_rows_staged", 1 if rows_mode == "copy" else 0)
    h_tile = nfx.pinned_empty((3, T, T), np.uint8)
    src = np.load(os.path.join(SHM, "tile.npy"), mmap_mode="r")
    for r0 in range(0, T, 2048):
        h_tile[:, r0:r0 + 2048] = src[:, r0:r0 + 2048]
    sets = [0, 1, 2, 3, 4]
    rows = nfx.pinned_empty((n, sum(nfx.num_columns(s) for s in sets)), np.float32)
    times = []
    for _ in range(reps):
        rows[:] = 0
        t0 = time.perf_counter()
        _, _, st = ctx.extract_tile_streamed(sets, off, xy, h_tile, ps, chunk_nuclei=chunk, block=block, out=rows)
        times.append(1e3 * (time.perf_counter() - t0))
        if os.environ.get("AB_NOCOPY_AFTER_FIRST") == "1":   # measurement only, needs a library built with -DNFX_EXPERIMENTS (NFX_LIB_PATH, scripts/build_variant.sh):
            os.environ["NFX_EXP_STREAM_NOCOPY"] = "1"        # later calls find the bands where the first call left them
        time.sleep(0.2)
    same = bool(np.array_equal(rows.view(np.uint32), np.asarray(ref).view(np.uint32)))
    print(f"run torch={with_torch} rows={rows_mode} chunk={chunk or 100000} block={block or 2048}: streamed calls {', '.join(f'{t:.1f}' for t in times)} ms "
          f"(resident-tile call {t_res} ms); rows == resident rows: {same}; chunks {st['chunks']}, bands {st['blocks_copied']}/{st['blocks_total']}, "
          f"pinned {st['image_pinned']}; HIP runtimes mapped: {hip_runtimes()}", flush=True)
    ctx.close()


if __name__ == "__main__":
    if sys.argv[1] == "gen":
        gen(int(sys.argv[2]) if len(sys.argv) > 2 else 1_250_000, int(sys.argv[3]) if len(sys.argv) > 3 else 64)
    else:
        run(int(sys.argv[2]), sys.argv[3] if len(sys.argv) > 3 else "copy", int(sys.argv[4]) if len(sys.argv) > 4 else 0,
            int(sys.argv[5]) if len(sys.argv) > 5 else 0)
