"""Patch sizes above 64 (args.rs:93-94 takes any u32; the reference's `-p 128` works): the "wide" kernel family (65..128: one
workgroup per nucleus, mask rows of two words, lists and maps in LDS) and the "huge" one (129..512: rows of up to eight words,
level maps and dense count tables in device scratch memory, Gabor on up to 15 x 15 tiles) against the oracle, which is written
for any patch size.  Same bar as the row-word kernels: masks, geometry's mask-derived columns and the texture counts bit-exact,
floats at 1e-5."""
import numpy as np
import pytest

import _oracle as orc
import _parity as par

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(nfx):
    c = nfx.Context(0)
    yield c
    c.close()


def _rings(ps, n, seed):
    off, xy, _ = par.make_inputs(seed, n, ps, with_patches=False)
    return par.concat_rings((off, xy), par.edge_case_rings(ps))


@pytest.mark.parametrize("ps", [65, 96, 128, 129, 200, 257, 512])
def test_wide_masks_bit_exact(ctx, ps):
    off, xy = _rings(ps, 40 if ps <= 200 else 12, 51)
    n = len(off) - 1
    cen, ctr = ctx.preprocess_polygons(off, xy)
    masks, rows = ctx.rasterize_polygons(off, ctr, ps, rows=True)
    W = (ps + 63) // 64
    assert rows.shape == (n, ps, W)
    for i in range(n):
        want = orc.raster_polygon(orc.preprocess_polygon(xy[off[i]:off[i + 1]])[1], ps)
        assert np.array_equal(masks[i, 0], want), f"mask {i} differs in {(masks[i, 0] != want).sum()} pixels"
        bits = np.concatenate([(rows[i, :, w][:, None] >> np.arange(64, dtype=np.uint64)[None, :]) & np.uint64(1) for w in range(W)], axis=1)[:, :ps]
        assert np.array_equal(bits.astype(np.float32), want)
    assert masks.sum() > 0


@pytest.mark.parametrize("ps", [80, 128])
def test_wide_all_sets_match_the_oracle(nfx, synth, ctx, ps):
    n = 26
    off, xy, pu8 = par.make_inputs(52, n, ps)
    e_off, e_xy = par.edge_case_rings(ps)
    ne = len(e_off) - 1
    off, xy = par.concat_rings((off, xy), (e_off, e_xy))
    pu8 = np.concatenate([pu8, pu8[:ne]])
    sets = [0, 1, 2, 3, 4]
    want, wcen = orc.extract(sets, off, xy, pu8, ps, batch_size=10, threads=16)
    got, gcen = ctx.extract(sets, off, xy, pu8, ps, batch_size=10)
    assert np.array_equal(gcen, wcen, equal_nan=True)
    par.assert_rows_close(got, want, par.column_scales(nfx, sets, ps), exact_cols=(1, 2, 3, 8), what=f"wide ps={ps}")
    # device memory, CLI order with a repeated set
    import torch
    d_off = torch.from_numpy(off.astype(np.int32)).cuda()
    dev, _ = ctx.extract([2, 0, 0], d_off, torch.from_numpy(xy).cuda(), torch.from_numpy(pu8).cuda(), ps, batch_size=10)
    ctx.synchronize()
    dev = dev.cpu().numpy()
    assert np.array_equal(dev[:, :224], got[:, 30:254], equal_nan=True) and np.array_equal(dev[:, 224:236], got[:, :12], equal_nan=True)
    assert np.array_equal(dev[:, 236:], got[:, :12], equal_nan=True)


def test_wide_texture_counts_are_exact(ctx):
    ps, n = 96, 10
    off, xy, pu8 = par.make_inputs(53, n, ps)
    c = 900.0
    full = np.array([[c - 2 * ps, c - 2 * ps], [c + 2 * ps, c - 2 * ps], [c + 2 * ps, c + 2 * ps], [c - 2 * ps, c + 2 * ps], [c - 2 * ps, c - 2 * ps]], np.float32)
    off, xy = par.concat_rings((off, xy), (np.array([0, 5], np.uint32), full))    # a mask that covers the whole patch: 96 x 95 pairs
    pu8 = np.concatenate([pu8, pu8[:1]])
    got = ctx.debug_texture_counts(off, xy, pu8, ps)
    want = orc.texture_counts(off, xy, pu8, ps)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert int(want[0][:, 3, 0, 0].max()) > 4032    # more pairs than any 64 x 64 patch holds (beyond the wave kernel's tables)


@pytest.mark.parametrize("set_id", range(5))
def test_wide_trait_entry(nfx, ctx, set_id):
    """compute_features_batched: f32 patchs + f32 masks of one reference batch (features/mod.rs:21-27) at ps = 72."""
    ps, n = 72, 9
    off, xy, pu8 = par.make_inputs(54, n, ps)
    patchs = (pu8.astype(np.float32) / np.float32(255.0)).astype(np.float32)
    cen, ctr = ctx.preprocess_polygons(off, xy)
    masks = ctx.rasterize_polygons(off, ctr, ps)
    want = orc.compute_features_batched(set_id, off, ctr, patchs, masks[:, 0])
    got = ctx.compute_features_batched(set_id, off, ctr, patchs, masks)
    par.assert_rows_close(got, want, par.column_scales(nfx, [set_id], ps), what=f"wide trait {nfx.feature_set_cli_name(set_id)}")


def test_wide_tile_entry(nfx, synth, ctx):
    ps, n, H, W = 100, 30, 400, 420
    s = synth.SEED_BASE + 55
    cx, cy = synth.centres(s, 0, n, ps, W, H)
    cx[:3] = [1.5, W - 2.0, 50.5]
    cy[:3] = [2.5, H - 1.0, 0.0]
    off, xy = synth.polygons(s, 0, n, ps, cx, cy)
    img = synth.tile(s, H, W, off, xy, ps)
    sets = [0, 1, 3]
    rows, cen = ctx.extract_tile(sets, off, xy, img, ps, batch_size=8)
    patches = np.stack([orc.crop_patch(img, c[0], c[1], ps) for c in cen])
    assert np.array_equal(ctx.gather_patches(cen, img, ps), patches)
    want, wcen = orc.extract(sets, off, xy, patches, ps, batch_size=8)
    assert np.array_equal(wcen, cen, equal_nan=True)
    par.assert_rows_close(rows, want, par.column_scales(nfx, sets, ps), what="wide tile entry")


@pytest.mark.parametrize("ps,n,sets", [
    (129, 10, [0, 1, 2, 3, 4]),     # the first size of the huge family: three words per row, 4 x 4 Gabor tiles
    (200, 6, [0, 1, 2, 3, 4]),      # four words per row, rows 0..199 on threads 0..199
    (300, 3, [0, 1, 2, 3]),         # rows t and t + 256 on one thread, five words per row (Gabor: the oracle's dense 30 x 30 bank is too slow here)
    (512, 2, [0, 1, 2, 3]),         # the limit: eight words per row, 2^18 pixels per level map
])
def test_huge_sets_match_the_oracle(nfx, ctx, ps, n, sets):
    off, xy, pu8 = par.make_inputs(57, n, ps)
    e_off, e_xy = par.edge_case_rings(ps)
    ne = min(len(e_off) - 1, 4)
    off, xy = par.concat_rings((off, xy), (e_off[:ne + 1], e_xy[:e_off[ne]]))
    c = 5000.0   # ... and a ring that covers the whole patch: ps x (ps - 1) pairs per offset
    full = np.array([[c - 2 * ps, c - 2 * ps], [c + 2 * ps, c - 2 * ps], [c + 2 * ps, c + 2 * ps], [c - 2 * ps, c + 2 * ps], [c - 2 * ps, c - 2 * ps]], np.float32)
    off, xy = par.concat_rings((off, xy), (np.array([0, 5], np.uint32), full))
    pu8 = np.concatenate([pu8] + [pu8[:1]] * (ne + 1))
    want, wcen = orc.extract(sets, off, xy, pu8, ps, batch_size=4, threads=16)
    got, gcen = ctx.extract(sets, off, xy, pu8, ps, batch_size=4)
    assert np.array_equal(gcen, wcen, equal_nan=True)
    par.assert_rows_close(got, want, par.column_scales(nfx, sets, ps), exact_cols=(1, 2, 3, 8), what=f"huge ps={ps}")
    # the integers behind the texture columns, ==
    k = min(3, n)
    sel_off, sel_xy = off[:k + 1], xy[:off[k]]
    gs, gb = ctx.debug_texture_counts(sel_off, sel_xy, pu8[:k], ps)
    ws, wb = orc.texture_counts(sel_off, sel_xy, pu8[:k], ps)
    assert np.array_equal(gs, ws) and np.array_equal(gb, wb)


def test_gabor_on_more_than_64_tiles(nfx, ctx):
    """above 280 pixels a bounding box holds more than 64 candidate Gabor tiles, so k_gabor_plan's tile list is written by
    several waves (cross-wave prefix over the ballots) and the scratch holds up to 225 tiles per nucleus -- never checked below ps 300.
    Two nuclei at 300 x 300 (81 candidate tiles), one ring covering the whole patch, one typical: the 96 Gabor columns against the
    oracle's dense 30 x 30 bank (a few seconds on 16 threads for two nuclei)."""
    ps = 300
    off, xy, pu8 = par.make_inputs(63, 1, ps)
    c = 5000.0
    full = np.array([[c - 2 * ps, c - 2 * ps], [c + 2 * ps, c - 2 * ps], [c + 2 * ps, c + 2 * ps], [c - 2 * ps, c + 2 * ps], [c - 2 * ps, c - 2 * ps]], np.float32)
    off, xy = par.concat_rings((np.array([0, 5], np.uint32), full), (off, xy))
    pu8 = np.concatenate([pu8, pu8])
    want, _ = orc.extract([4], off, xy, pu8, ps, batch_size=2, threads=16)
    got, _ = ctx.extract([4], off, xy, pu8, ps, batch_size=2)
    assert np.isfinite(want).all() and np.abs(want).max() > 1.0
    par.assert_rows_close(got, want, par.column_scales(nfx, [4], ps), what="gabor on 81 tiles")


def test_huge_trait_and_tile_entries(nfx, synth, ctx):
    """ps = 160 through compute_features_batched (f32 patchs + masks) and through the tile entry."""
    ps, n = 160, 5
    off, xy, pu8 = par.make_inputs(58, n, ps)
    patchs = (pu8.astype(np.float32) / np.float32(255.0)).astype(np.float32)
    cen, ctr = ctx.preprocess_polygons(off, xy)
    masks = ctx.rasterize_polygons(off, ctr, ps)
    for set_id in range(5):
        want = orc.compute_features_batched(set_id, off, ctr, patchs, masks[:, 0])
        got = ctx.compute_features_batched(set_id, off, ctr, patchs, masks)
        par.assert_rows_close(got, want, par.column_scales(nfx, [set_id], ps), what=f"huge trait {nfx.feature_set_cli_name(set_id)}")
    H, W = 500, 520
    s = synth.SEED_BASE + 59
    cx, cy = synth.centres(s, 0, n, ps, W, H)
    cx[:2] = [1.5, W - 2.0]
    cy[:2] = [2.5, H - 1.0]
    off, xy = synth.polygons(s, 0, n, ps, cx, cy)
    img = synth.tile(s, H, W, off, xy, ps)
    sets = [0, 1, 3]
    rows, cen = ctx.extract_tile(sets, off, xy, img, ps, batch_size=8)
    patches = np.stack([orc.crop_patch(img, c[0], c[1], ps) for c in cen])
    assert np.array_equal(ctx.gather_patches(cen, img, ps), patches)
    want, wcen = orc.extract(sets, off, xy, patches, ps, batch_size=8)
    par.assert_rows_close(rows, want, par.column_scales(nfx, sets, ps), what="huge tile entry")


def test_patch_size_limit_is_512(nfx, ctx):
    off, xy, pu8 = par.make_inputs(56, 2, 32)
    with pytest.raises(nfx.NfxError, match="patch size > 512"):
        ctx.extract([0], off, xy, None, 513)
