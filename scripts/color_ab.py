#!/usr/bin/env python3
"""A/B of the colour kernels (k_hue_batch, k_color) between libnfx.so builds (run on the GPU box): the default library against every
scratch_exp/*.so, colour set alone on configs[4]'s shard size (64 x 64) and on configs[2]'s (48 x 48); rows of every variant compared with
the default's bit for bit (reported, not required: a variant may round its sums in another order).
    python scripts/color_ab.py [nuclei]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = sys.argv[1] if len(sys.argv) > 1 else "1250000"
libs = [None] + sorted(p for p in os.listdir(os.path.join(ROOT, "scratch_exp")) if p.endswith(".so"))
for wl in ("config5", "config3"):
    for lib in libs:
        env = dict(os.environ)
        if lib:
            env["NFX_LIB_PATH"] = os.path.join(ROOT, "scratch_exp", lib)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-also", "--workload", wl, "--inputs", "resident", "--nuclei", n,
                              "--steps", "5", "--warmup", "2", "--sets", "color"], env=env, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            k = d["roofline"]["kernels_ms_per_step"]
            print(f"{wl} {lib or 'default':28s} step {d['ms_per_step']:8.3f} ms  k_hue_batch {k.get('k_hue_batch')}  k_color {k.get('k_color')}  k_raster_shape {k.get('k_raster_shape')}", flush=True)
        except Exception:
            print(wl, lib, "ERR", out.stderr[-600:], flush=True)
