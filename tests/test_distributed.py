"""N>1 path on CPU: world-size-2 gloo.  Each rank extracts its index shard (the CPU oracle stands in for
the GPU engine here -- this test is about sharding and the gather, not the kernels) and the gathered
matrix must equal the single-process result row for row, including the batch-coupled mean_h."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "nuclei-feature-extraction_amd"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, ps, batch, aligned, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import _oracle as orc
    import _parity as par
    D = importlib.import_module(PKG + ".distributed")
    off, xy, pu8 = par.make_inputs(31, n, ps)
    if aligned:
        i0, i1 = D.shard_range(n, batch, rank, world)
        counts = D.shard_counts(n, batch, world)
    else:  # deliberately cut in the middle of a reference batch
        cut = batch + batch // 2
        i0, i1 = (0, cut) if rank == 0 else (cut, n)
        counts = [cut, n - cut]
    so = (off[i0:i1 + 1] - off[i0]).astype(np.uint32)
    rows, _ = orc.extract([0, 1], so, xy[off[i0]:off[i1]], pu8[i0:i1], ps, batch_size=batch)
    full = D.gather_rows(torch.from_numpy(rows), counts, dst=0)
    if rank == 0:
        q.put(full.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _run(aligned):
    n, ps, batch = 70, 32, 16
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, ps, batch, aligned, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as orc
    import _parity as par
    off, xy, pu8 = par.make_inputs(31, n, ps)
    want, _ = orc.extract([0, 1], off, xy, pu8, ps, batch_size=batch)
    return got, want


def test_sharded_gather_equals_single_process():
    got, want = _run(aligned=True)
    assert got.shape == want.shape
    assert np.array_equal(got, want, equal_nan=True)


def test_shard_cut_inside_a_batch_changes_mean_h():
    """Why shard boundaries must be batch multiples: mean_h (column 12+6) couples a batch."""
    got, want = _run(aligned=False)
    col = 12 + 6
    assert not np.allclose(got[:, col], want[:, col], rtol=1e-6, atol=1e-6, equal_nan=True)
    other = [c for c in range(got.shape[1]) if c not in (col, 12 + 9)]
    assert np.array_equal(got[:, other], want[:, other], equal_nan=True)


def test_shard_ranges_cover_in_order():
    D = importlib.import_module(PKG + ".distributed")
    for n, B, R in [(1000, 100, 8), (1050, 100, 8), (70, 16, 2), (5, 100, 4), (0, 100, 2), (1_000_000, 100, 8)]:
        prev = 0
        for r in range(R):
            i0, i1 = D.shard_range(n, B, r, R)
            assert i0 == prev and i0 % B == 0 or i0 == n
            prev = i1
        assert prev == n
        assert sum(D.shard_counts(n, B, R)) == n
