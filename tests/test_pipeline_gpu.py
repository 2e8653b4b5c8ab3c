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
