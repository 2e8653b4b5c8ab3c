"""Streamed H2D + compute overlap (BASELINE.json configs[4]: "streamed tile H2D overlapped with compute"):
chunked, double-buffered extraction from pinned host memory must equal one resident call bit for bit,
including the batch-coupled mean_h."""
import importlib

import numpy as np
import pytest

import _parity as par

pytestmark = pytest.mark.gpu


def test_streamed_extraction_equals_resident(nfx):
    import torch
    pipeline = importlib.import_module("nuclei-feature-extraction_amd.pipeline")
    ps, n, batch = 32, 1030, 50
    off, xy, pu8 = par.make_inputs(61, n, ps)
    sets = [0, 1, 2, 3, 4]
    ctx = nfx.Context(0)
    want, wcen = ctx.extract(sets, off, xy, pu8, ps, batch_size=batch)
    rows, cen = pipeline.extract_streamed(ctx, sets, off, xy, pu8, ps, batch_size=batch, chunk_nuclei=250)   # 5 chunks, last ragged
    torch.cuda.synchronize()
    assert np.array_equal(rows.cpu().numpy(), want, equal_nan=True)
    assert np.array_equal(cen.cpu().numpy(), wcen, equal_nan=True)
    ctx.set_stream(0)
    ctx.close()


def test_device_entry_can_be_captured_in_a_hip_graph(nfx, synth):
    """NFX_MEM_DEVICE calls only enqueue work (no allocation once the workspace has its size, no synchronisation; the side
    stream of the polygon kernel forks from and joins the calling stream), so a whole extraction replays as one graph."""
    import torch
    ps, n = 48, 500
    seed = synth.SEED_BASE + 61
    off, xy = synth.polygons(seed, 0, n, ps)
    pa, pb = synth.patches(seed, 0, n, ps), synth.patches(seed + 1, 0, n, ps)
    sets = [0, 1, 2, 3, 4]
    dev = torch.device("cuda", 0)
    d_off = torch.from_numpy(off.astype(np.int32)).to(dev)
    d_xy = torch.from_numpy(xy).to(dev)
    d_p = torch.from_numpy(pa).to(dev)
    out = torch.empty((n, 418), dtype=torch.float32, device=dev)
    ctx = nfx.Context(0)
    try:
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx.extract(sets, d_off, d_xy, d_p, ps, out=out, want_centroids=False)      # warm-up: tables, workspace
        torch.cuda.synchronize()
        want_a = out.clone()
        d_p.copy_(torch.from_numpy(pb).to(dev))
        ctx.extract(sets, d_off, d_xy, d_p, ps, out=out, want_centroids=False)
        torch.cuda.synchronize()
        want_b = out.clone()
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ctx.set_stream(side.cuda_stream)
            with torch.cuda.graph(g, stream=side):
                ctx.extract(sets, d_off, d_xy, d_p, ps, out=out, want_centroids=False)
        torch.cuda.current_stream().wait_stream(side)
        for patches, want in ((pa, want_a), (pb, want_b), (pa, want_a)):
            d_p.copy_(torch.from_numpy(patches).to(dev))
            out.zero_()
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(torch.nan_to_num(out), torch.nan_to_num(want))
    finally:
        ctx.set_stream(0)
        ctx.close()
