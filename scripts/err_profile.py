#!/usr/bin/env python3
"""Error profile of the HIP path against the oracle (run on the GPU box): per feature set and column, the worst absolute
error, the column's median |value| and their ratio -- what a data-driven tolerance floor (tests/_parity.py) can be.
    python scripts/err_profile.py [nuclei] > gpurun_out/err_profile.txt"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as orc   # noqa: E402
import _parity as par   # noqa: E402

nfx = importlib.import_module("nuclei-feature-extraction_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
ctx = nfx.Context(0)
sets = [0, 1, 2, 3, 4]
names = nfx.all_column_names(sets)[1:]
for seed, ps in ((7, 17), (8, 24), (1, 32), (2, 48), (4, 64)):
    off, xy, pu8 = par.make_inputs(seed, n, ps)
    got, _ = ctx.extract(sets, off, xy, pu8, ps, batch_size=100)
    want, _ = orc.extract(sets, off, xy, pu8, ps, batch_size=100, threads=16)
    g, w = got.astype(np.float64), want.astype(np.float64)
    ok = np.isfinite(g) & np.isfinite(w)
    err = np.where(ok, np.abs(g - w), 0.0)
    med = np.array([np.median(np.abs(w[ok[:, c], c])) if ok[:, c].any() else 0.0 for c in range(w.shape[1])])
    with np.errstate(divide="ignore", invalid="ignore"):
        rel_to_med = err.max(0) / med
        pure_rel_all = np.where(ok & (w != 0), err / np.abs(w), 0.0)
        pure_rel = pure_rel_all.max(0)
    col = 0
    try:
        scaled = f"{par.assert_rows_close(got, want, par.column_scales(nfx, sets, ps)):.3e}"
    except AssertionError as e:
        scaled = "FAILS: " + str(e)[:300]
    print(f"== ps {ps}, {n} nuclei; scaled error under tests/_parity.py: {scaled} (limit 1e-5)")
    for s in sets:
        k = nfx.num_columns(s)
        sl = slice(col, col + k)
        order = np.argsort(-rel_to_med[sl])[:4]
        print(f"  set {nfx.feature_set_cli_name(s)}: worst err/median {np.nanmax(rel_to_med[sl]):.3e}, worst pure relative {pure_rel[sl].max():.3e}")
        for j in order:
            c = col + j
            print(f"     {names[c]:55s} max|err| {err[:, c].max():.3e}  median|v| {med[c]:.3e}  err/median {rel_to_med[c]:.3e}  pure rel {pure_rel[c]:.3e}")
        # the same by PURE relative error (no floor): the column, the nucleus, both values -- a sign flip or a 0-vs-tiny on a tiny mask shows here
        for j in np.argsort(-pure_rel[sl])[:3]:
            c = col + j
            if pure_rel[c] == 0.0:
                continue
            i = int(np.argmax(pure_rel_all[:, c]))
            print(f"     by pure relative: {names[c]:45s} nucleus {i:5d}  hip {got[i, c]: .9e}  oracle {want[i, c]: .9e}  |err| {err[i, c]:.3e}  pure rel {pure_rel[c]:.3e}  "
                  f"column median {med[c]:.3e}  mask area {int(round(float(want[i, 0])))}")
        if s == 4:
            for tag, sel in (("mean", slice(col, col + k, 2)), ("variance", slice(col + 1, col + k, 2))):
                e = err[:, sel].max(0)
                j = int(np.argmax(e))
                print(f"     gabor {tag} columns: worst abs err {e.max():.3e} (column median {med[sel][j]:.3e}); worst err/median {np.nanmax(rel_to_med[sel]):.3e}")
            for fi, fr in enumerate(("0.5", "1", "2", "4", "6", "8")):   # variances by frequency: what their floor has to be
                cv = [col + a * 12 + fi * 2 + 1 for a in range(8)]
                print(f"     gabor variance, frequency {fr:>3s}: worst abs err {err[:, cv].max():.3e}  column medians {med[cv].min():.3e}..{med[cv].max():.3e}  "
                      f"worst err/median {np.nanmax(rel_to_med[cv]):.3e}  worst pure relative {pure_rel[cv].max():.3e}")
        col += k
ctx.close()
