#!/usr/bin/env python3
"""Generates tests/golden/nfx_golden_ps{32,48,64}.npz.

PROVENANCE: the reference (oxabz/nuclei-feature-extraction) ships no tests, fixtures or golden vectors and
cannot be built or imported here (Rust, un-vendored git dependencies).  These vectors are therefore produced
by THIS repository's CPU oracle (oracle/nfx_oracle.c) -- whose in-tree arithmetic is pinned to torch-CPU ATen
by tests/test_torch_twin.py and whose dependency-resident functions are build decisions (DESIGN.md §4,
"parity unpinned").  They freeze the oracle's behaviour so that neither the oracle nor the HIP path can
drift silently.

    python tests/golden/make_golden.py
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as orc      # noqa: E402
import _parity as par      # noqa: E402

SETS = [0, 1, 2, 3, 4]
BATCH = 10


def build(ps, n=14, seed=77):
    off, xy, pu8 = par.make_inputs(seed, n, ps)
    e_off, e_xy = par.edge_case_rings(ps)
    ne = len(e_off) - 1
    off, xy = par.concat_rings((off, xy), (e_off, e_xy))
    pu8 = np.concatenate([pu8, pu8[:ne]])
    rows, cen = orc.extract(SETS, off, xy, pu8, ps, batch_size=BATCH)
    masks = np.stack([orc.raster_polygon(orc.preprocess_polygon(xy[off[i]:off[i + 1]])[1], ps) for i in range(len(off) - 1)])
    # the integers behind the texture columns (layout of nfx_debug_texture_counts): compared == with the HIP kernels' own
    glcm_sums, glrlm_bins = orc.texture_counts(off, xy, pu8, ps)
    return dict(glcm_sums=glcm_sums, glrlm_bins=glrlm_bins, ring_offsets=off, ring_xy=xy, patches_u8=pu8, ps=np.int32(ps), batch_size=np.int32(BATCH),
                sets=np.asarray(SETS, np.int32), rows=rows, centroids=cen, mask_bits=np.packbits(masks.astype(bool), axis=-1))


if __name__ == "__main__":
    for ps in (32, 48, 64):
        d = build(ps)
        path = os.path.join(HERE, f"nfx_golden_ps{ps}.npz")
        np.savez_compressed(path, **d)
        print(path, os.path.getsize(path), "bytes", d["rows"].shape)
