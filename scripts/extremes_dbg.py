"""Prints HIP vs oracle values where they disagree on the extreme textures (debug aid for tests/test_texture_extremes_gpu.py)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch; torch.cuda.init()
import _oracle as orc, _parity as par
import test_texture_extremes_gpu as T
nfx = importlib.import_module("nuclei-feature-extraction_amd")
ctx = nfx.Context(0)
ps = int(sys.argv[1]) if len(sys.argv) > 1 else 32
tex = T.textures(ps, np.random.default_rng(ps)); rs = T.rings(ps)
names, ring_list, patches = [], [], []
for nm, p in tex.items():
    for k, r in enumerate(rs):
        names.append(f"{nm} / ring {k}"); ring_list.append(r); patches.append(p)
off = np.zeros(len(ring_list) + 1, np.uint32); np.cumsum([len(r) for r in ring_list], out=off[1:])
xy = np.concatenate(ring_list).astype(np.float32); pu8 = np.stack(patches)
sets = [0, 1, 2, 3, 4]
want, _ = orc.extract(sets, off, xy, pu8, ps, batch_size=7)
got, _ = ctx.extract(sets, off, xy, pu8, ps, batch_size=7)
cols = nfx.all_column_names(sets)[1:]
sc = par.column_scales(nfx, sets, ps)
for i, nm in enumerate(names):
    g, w = got[i].astype(np.float64), want[i].astype(np.float64)
    nan_diff = np.isnan(g) != np.isnan(w)
    err = np.abs(np.nan_to_num(g) - np.nan_to_num(w)) / np.maximum(np.abs(np.nan_to_num(w)), sc)
    bad = np.where(nan_diff | (err > 1e-5))[0]
    if len(bad):
        print(f"== {nm}: {len(bad)} columns")
        for c in bad[:12]:
            print(f"   {cols[c]:55s} hip {got[i, c]!r:>16} oracle {want[i, c]!r:>16}")
