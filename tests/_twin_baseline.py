#!/usr/bin/env python3
"""A libtorch-shaped CPU data point for the geometry path (test / bench infrastructure, never on the product path).

The reference runs `geometry` as rayon `par_chunks(batch_size)` over the nuclei, one batch per worker thread, and inside a batch a
serial per-nucleus loop of ~25 small ATen calls, one LAPACK `geev`, one ellipse rasterisation and >= 10 scalar reads
(main.rs:146-151, shape.rs:66-111).  This script times exactly that SHAPE on torch-CPU: one worker PROCESS per core with
`torch.set_num_threads(1)` (rayon's one-batch-per-thread model; default thread count = cores, args.rs:95-99), batch 100, each nucleus
through tests/_torch_twin.py's op-for-op ATen sequence.  The three dependency-resident pieces the twin cannot issue (their sources
are absent) are stood in for by the C oracle -- polygon and ellipse rasterisation, geometric_features' polygon columns -- which is
CHEAPER than the tensor programs tch-utils runs for them, so the figure flatters the reference.  It is "a restatement of the
reference's libtorch-CPU path", not the reference.

    python tests/_twin_baseline.py --ps 32 --nuclei 100000 --batch 100 --procs 16 --seconds 12   -> one JSON line
Must be started BEFORE the calling process touches a GPU, or as an ordinary child process (it never touches one itself).
"""
import argparse
import importlib
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

_state = {}


def _init():
    import torch
    torch.set_num_threads(1)
    import _oracle as orc
    import _torch_twin as tw
    _state.update(torch=torch, orc=orc, tw=tw, synth=importlib.import_module("nuclei-feature-extraction_amd.synth"))


def _batch(job):
    """one reference batch: loader stand-in (preprocess + rasterise) + ShapeFeatureSet::compute_features_batched"""
    seed, i0, m, ps = job
    torch, orc, tw, synth = _state["torch"], _state["orc"], _state["tw"], _state["synth"]
    off, xy = synth.polygons(seed, i0, m, ps)
    t0 = time.perf_counter()
    half = np.float32(ps) / np.float32(2.0)
    rows = np.empty((m, 12), np.float64)
    for i in range(m):
        _, ctr = orc.preprocess_polygon(xy[off[i]:off[i + 1]])           # utils.rs:54-74
        mask = torch.from_numpy(orc.raster_polygon(ctr, ps)).unsqueeze(0)  # tch_utils::shapes::polygon stand-in
        com = tw.center_of_mass(mask)                                    # shape.rs:71
        res = tw.major_minor_axes_w_angle(mask)                          # shape.rs:76
        major, minor, angle = res[0], res[1], res[2]
        ecc = tw.eccentricity(major, minor) if major == major else float("nan")
        ell = orc.raster_ellipse(ps, np.float32(com[1]) - half, np.float32(com[0]) - half, major, minor, angle)  # shapes::ellipse stand-in
        ed = tw.eliptic_deviation(mask, torch.from_numpy(ell).unsqueeze(0))  # shape.rs:105
        p = orc.polygon_features_f64(ctr)                                # geometric_features::* stand-in (shape.rs:89-97)
        rows[i] = (p[0], major, minor, ecc, angle, p[1], p[2], p[3], ed, p[4], p[5], p[6])
    return m, time.perf_counter() - t0, float(np.nansum(rows[:, 1]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ps", type=int, default=32)
    ap.add_argument("--nuclei", type=int, default=100_000)
    ap.add_argument("--batch", type=int, default=100)
    ap.add_argument("--procs", type=int, default=min(len(os.sched_getaffinity(0)), 16))
    ap.add_argument("--seconds", type=float, default=12.0)
    ap.add_argument("--seed-index", type=int, default=1)
    a = ap.parse_args()
    synth = importlib.import_module("nuclei-feature-extraction_amd.synth")
    seed = synth.SEED_BASE + a.seed_index
    with mp.get_context("fork").Pool(a.procs, initializer=_init) as pool:
        pool.map(_batch, [(seed, 0, 2, a.ps)] * a.procs)   # workers up (torch imported, oracle loaded) before anything is timed
        # calibrate on one batch per worker, then size the sample for the asked wall time
        t0 = time.perf_counter()
        cal = pool.map(_batch, [(seed, k * a.batch, a.batch, a.ps) for k in range(a.procs)])
        dt = time.perf_counter() - t0
        rate = sum(c[0] for c in cal) / dt
        nb = int(max(a.procs, min(a.nuclei // a.batch, rate * a.seconds / a.batch)))
        t0 = time.perf_counter()
        res = pool.map(_batch, [(seed, k * a.batch, a.batch, a.ps) for k in range(nb)], chunksize=1)
        wall = time.perf_counter() - t0
    n = sum(r[0] for r in res)
    print(json.dumps({"value": round(n / wall, 1), "unit": "nuclei/s", "cores": a.procs, "kind": "aten-twin",
                      "sample": f"first {n} nuclei of the workload in {nb} reference batches of {a.batch}, one process per core "
                                f"(torch.set_num_threads(1)), {wall:.1f} s",
                      "what": "tests/_torch_twin.py: the reference's per-nucleus ATen op sequence for ShapeFeatureSet (shape.rs:66-111; "
                              "center_of_mass, covariance mm, linalg_eig, eliptic_deviation) on torch-CPU, batch-per-worker like rayon "
                              "(main.rs:146-151); rasterisers and geometric_features stood in for by the C oracle (cheaper than "
                              "tch-utils' tensor programs): a restatement of the reference's libtorch-CPU path, not the reference",
                      "per_nucleus_us_single_core": round(1e6 * sum(r[1] for r in res) / n, 1)}), flush=True)


if __name__ == "__main__":
    main()
