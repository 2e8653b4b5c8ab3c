"""Streamed extraction: host-resident inputs fed to the GPU in chunks, H2D copies overlapped with compute.

The reference's pipeline stage `move_tensors_to_device` (utils.rs:211-224) copies every batch's two f32
tensors synchronously from the rayon worker that is about to compute on them.  Here the packed u8 patches
(4x fewer bytes than one f32 tensor, 5.3x fewer than patch + mask) stay in pinned host memory and are
copied by a side stream into one of two device buffers while the engine's stream works on the other; with
`host_rows` the finished rows of a chunk travel back to pinned host memory on a third stream while the next
chunk is computed (the link is full duplex):

    copy stream     : H2D(chunk 0) H2D(chunk 1)         H2D(chunk 2)         ...
    compute stream  :              extract(chunk 0)     extract(chunk 1)     ...
    rows stream     :                                   D2H(rows 0)          D2H(rows 1)  ...

Chunk boundaries are multiples of the reference batch size, so the batch-coupled hue statistics
(color.rs:144-155) are unchanged: the result is bit-identical to one resident call.
PyTorch is used for pinned memory, streams and events only.
"""
from __future__ import annotations

from typing import Sequence

import numpy as np


def extract_streamed(ctx, sets: Sequence[int], ring_offsets: np.ndarray, ring_xy: np.ndarray, patches_u8_host, ps: int,
                     batch_size: int = 100, chunk_nuclei: int = 100_000, out=None, host_rows=None, want_centroids: bool = True):
    """ring_offsets [n+1] / ring_xy [V,2] numpy (small: uploaded once); patches_u8_host: pinned torch uint8
    tensor [n,3,ps,ps] (or numpy, pinned here); host_rows: optional pinned f32 tensor [n,F] that receives the rows
    (the call then returns only when the last row is on the host).
    Returns (rows [n,F] cuda tensor, centroids [n,2] cuda tensor or None)."""
    import importlib
    import torch

    nfx = importlib.import_module(__package__)
    n = len(ring_offsets) - 1
    F = sum(nfx.num_columns(s) for s in sets)
    dev = torch.device("cuda", ctx.device)
    if not isinstance(patches_u8_host, torch.Tensor):
        patches_u8_host = torch.from_numpy(np.ascontiguousarray(patches_u8_host))
    if not patches_u8_host.is_pinned():
        patches_u8_host = patches_u8_host.pin_memory()
    chunk = max(batch_size, (chunk_nuclei // batch_size) * batch_size)
    compute = torch.cuda.current_stream(dev)
    copy = torch.cuda.Stream(dev)
    ctx.set_stream(compute.cuda_stream)
    d_off_all = torch.from_numpy(np.ascontiguousarray(ring_offsets).astype(np.int64)).to(dev)
    d_xy = torch.from_numpy(np.ascontiguousarray(ring_xy, np.float32)).to(dev)
    rows = out if out is not None else torch.empty((n, F), dtype=torch.float32, device=dev)
    back = torch.cuda.Stream(dev) if host_rows is not None else None
    bufs = [torch.empty((min(chunk, n), 3, ps, ps), dtype=torch.uint8, device=dev) for _ in range(2)]
    ready = [torch.cuda.Event(), torch.cuda.Event()]   # copy of buffer b finished
    freed = [torch.cuda.Event(), torch.cuda.Event()]   # compute on buffer b finished
    starts = list(range(0, n, chunk))

    def issue_copy(k):
        b = k & 1
        i0 = starts[k]
        m = min(chunk, n - i0)
        with torch.cuda.stream(copy):
            if k >= 2:
                copy.wait_event(freed[b])
            bufs[b][:m].copy_(patches_u8_host[i0:i0 + m], non_blocking=True)
            ready[b].record(copy)

    if starts:
        issue_copy(0)
    for k, i0 in enumerate(starts):
        b = k & 1
        m = min(chunk, n - i0)
        if k + 1 < len(starts):
            issue_copy(k + 1)
        compute.wait_event(ready[b])
        v0, v1 = int(ring_offsets[i0]), int(ring_offsets[i0 + m])
        off = (d_off_all[i0:i0 + m + 1] - v0).to(torch.int32)
        ctx.extract(sets, off, d_xy[v0:v1], bufs[b][:m], ps, batch_size=batch_size, out=rows[i0:i0 + m], want_centroids=False)
        freed[b].record(compute)
        if back is not None:
            back.wait_event(freed[b])
            with torch.cuda.stream(back):
                host_rows[i0:i0 + m].copy_(rows[i0:i0 + m], non_blocking=True)
    cen = None
    if want_centroids:   # centroids for the key column: one cheap pass over all rings
        cen, _ = ctx.preprocess_polygons(d_off_all.to(torch.int32), d_xy)
    if back is not None:
        back.synchronize()
    return rows, cen
