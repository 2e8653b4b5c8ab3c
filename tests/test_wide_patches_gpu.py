"""Patch sizes above 64 (args.rs:93-94 takes any u32; the reference's `-p 128` works): the "wide" kernel family -- one
workgroup per nucleus, mask rows of two words -- against the oracle, which is written for any patch size.  Same bar as the
row-word kernels: masks, geometry's mask-derived columns and the texture counts bit-exact, floats at 1e-5."""
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


@pytest.mark.parametrize("ps", [65, 96, 128])
def test_wide_masks_bit_exact(ctx, ps):
    off, xy = _rings(ps, 40, 51)
    n = len(off) - 1
    cen, ctr = ctx.preprocess_polygons(off, xy)
    masks, rows = ctx.rasterize_polygons(off, ctr, ps, rows=True)
    assert rows.shape == (n, ps, 2)
    for i in range(n):
        want = orc.raster_polygon(orc.preprocess_polygon(xy[off[i]:off[i + 1]])[1], ps)
        assert np.array_equal(masks[i, 0], want), f"mask {i} differs in {(masks[i, 0] != want).sum()} pixels"
        bits = np.concatenate([(rows[i, :, w][:, None] >> np.arange(64, dtype=np.uint64)[None, :]) & np.uint64(1) for w in range(2)], axis=1)[:, :ps]
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


def test_patch_size_limit_is_128(nfx, ctx):
    off, xy, pu8 = par.make_inputs(56, 2, 32)
    with pytest.raises(nfx.NfxError, match="patch size > 128"):
        ctx.extract([0], off, xy, None, 129)
