"""Randomised parity sweep at ODD patch sizes (the byte-wise loaders, strips that do not divide the patch, boxes that touch every border):
glrlm + gabor and geometry + color + glcm against the oracle.  usage: python scripts/stress_odd_sizes.py [n_fast] [n_slow]"""
import importlib, sys, time, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import _parity as par, _oracle as orc
nfx = importlib.import_module("nuclei-feature-extraction_amd")
n_fast = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
n_slow = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
ctx = nfx.Context(0)
for ps in (17, 23, 31, 37, 45, 51, 59, 63):
    for sets, n in (([0, 1, 2], n_fast), ([3, 4], n_slow)):
        t0 = time.time()
        off, xy, pu8 = par.make_inputs(300 + ps, n, ps)
        got, gc = ctx.extract(sets, off, xy, pu8, ps, batch_size=100)
        want, wc = orc.extract(sets, off, xy, pu8, ps, batch_size=100, threads=16)
        assert np.array_equal(gc, wc, equal_nan=True)
        worst = par.assert_rows_close(got, want, par.column_scales(nfx, sets, ps), what=f"ps={ps} sets={sets}")
        got2, _ = ctx.extract(sets, off, xy, pu8, ps, batch_size=100)
        assert np.array_equal(got.view(np.uint32), got2.view(np.uint32)), "rows differ between two runs"
        print(f"ps {ps} sets {sets} n {n}: worst {worst:.3e}, identical between runs  ({time.time() - t0:.1f} s)", flush=True)
print("odd-size stress parity ok")
