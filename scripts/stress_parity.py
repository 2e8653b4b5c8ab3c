"""Larger randomised parity sweep against the oracle (not part of the test suite: minutes of CPU oracle time).
usage: python scripts/stress_parity.py [n_fast] [n_slow]        (STRESS_SEEDS=303,404 other seeds; STRESS_SETS=2 one fast group of these sets only)"""
import importlib, os, sys, time, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import _parity as par, _oracle as orc
nfx = importlib.import_module("nuclei-feature-extraction_amd")
n_fast = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
n_slow = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
ctx = nfx.Context(0)
seeds = tuple(int(x) for x in os.environ.get("STRESS_SEEDS", "101,202").split(","))
groups = (([0, 1, 2], n_fast), ([3, 4], n_slow))
if os.environ.get("STRESS_SETS"):
    groups = (([int(x) for x in os.environ["STRESS_SETS"].split(",")], n_fast),)
for ps in (32, 48, 64, 40, 56):
    for sets, n in groups:
        for seed in seeds:
            t0 = time.time()
            off, xy, pu8 = par.make_inputs(seed + ps, n, ps)
            got, gc = ctx.extract(sets, off, xy, pu8, ps, batch_size=100)
            want, wc = orc.extract(sets, off, xy, pu8, ps, batch_size=100, threads=16)
            assert np.array_equal(gc, wc, equal_nan=True)
            worst = par.assert_rows_close(got, want, par.column_scales(nfx, sets, ps), what=f"ps={ps} sets={sets} seed={seed}")
            print(f"ps {ps} sets {sets} n {n} seed {seed}: worst {worst:.3e}  ({time.time() - t0:.1f} s)", flush=True)
print("stress parity ok")
