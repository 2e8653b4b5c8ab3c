"""Parity at BASELINE.json's FULL sizes through a size-independent property.

Nuclei are independent of each other except inside a reference batch (the batch-coupled circular hue mean,
color.rs:144-155).  So a workload made of ONE batch of 100 nuclei repeated 10 000 (12 500) times must give, in every
repetition, the rows of that batch -- and the rows of one batch are small enough for the CPU oracle.  This checks the
million-nucleus launches (persistent grids, size-class lists, 64-bit indexing, workspace sizing) against the oracle,
not just against themselves."""
import numpy as np
import pytest

import _oracle as orc
import _parity as par

pytestmark = pytest.mark.gpu

BATCH = 100


def _tiled(torch, off, xy, pu8, reps):
    """device copies of the batch repeated `reps` times (offsets rebased per repetition)"""
    n, nv = len(off) - 1, int(off[-1])
    d_xy = torch.from_numpy(xy).cuda().repeat(reps, 1)
    base = torch.from_numpy(off[:-1].astype(np.int64)).cuda()
    shifts = (torch.arange(reps, device="cuda", dtype=torch.int64) * nv).unsqueeze(1)
    d_off = torch.cat([(base.unsqueeze(0) + shifts).reshape(-1), torch.tensor([reps * nv], device="cuda", dtype=torch.int64)]).to(torch.int32)
    d_p = torch.from_numpy(pu8).cuda().repeat(reps, 1, 1, 1)
    return d_off, d_xy, d_p, n * reps


@pytest.mark.parametrize("ps,reps,sets", [(48, 10_000, [0, 1, 2]),            # configs[2]: 1 M nuclei, geometry+color+glcm
                                           (64, 12_500, [0, 1, 2, 3, 4])])     # configs[4] shard: 1.25 M nuclei, all
def test_every_repetition_of_a_batch_equals_the_oracle(nfx, ps, reps, sets):
    import torch
    off, xy, pu8 = par.make_inputs(901 + ps, BATCH, ps)
    want, wcen = orc.extract(sets, off, xy, pu8, ps, batch_size=BATCH)
    ctx = nfx.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    d_off, d_xy, d_p, n = _tiled(torch, off, xy, pu8, reps)
    got, cen = ctx.extract(sets, d_off, d_xy, d_p, ps, batch_size=BATCH)
    torch.cuda.synchronize()
    F = got.shape[1]
    assert got.shape == (n, F) and n == BATCH * reps
    g3 = got.view(reps, BATCH, F)
    first = g3[0]
    # every repetition is bit-identical to the first one (NaNs in the same places)
    same = (g3 == first.unsqueeze(0)) | (torch.isnan(g3) & torch.isnan(first).unsqueeze(0))
    assert bool(same.all()), f"{int((~same).sum())} values differ between repetitions"
    c3 = cen.view(reps, BATCH, 2)
    assert bool(((c3 == c3[0].unsqueeze(0)) | (torch.isnan(c3) & torch.isnan(c3[0]).unsqueeze(0))).all())
    # ... and the first one is the oracle's batch
    assert np.array_equal(c3[0].cpu().numpy(), wcen, equal_nan=True)
    par.assert_rows_close(first.cpu().numpy(), want, par.column_scales(nfx, sets, ps), what=f"full size ps={ps}")
    ctx.close()
