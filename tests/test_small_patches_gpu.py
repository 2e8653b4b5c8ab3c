"""Patch sizes below everything the other tests use (the reference's `-p` takes any u32 >= 1, args.rs:93-94; its default is 64): 1 .. 15 take the
byte-wise loaders and strips far narrower than a filter, with masks of a handful of pixels -- or none.  Same bar as everywhere: masks and the
texture integers `==` the oracle's, every column at 1e-5 with the floors of tests/_parity.py, NaN patterns identical."""
import numpy as np
import pytest

import _oracle as orc
import _parity as par

pytestmark = pytest.mark.gpu


def _inputs(synth, ps, n, seed):
    """rings a few pixels across (the synthetic generator's radii scale with ps and vanish below ~6) + uniformly random bytes"""
    rng = np.random.default_rng(seed * 1000 + ps)
    rings, off = [], [0]
    for i in range(n):
        k = int(rng.integers(3, 9))
        ang = np.sort(rng.uniform(0, 2 * np.pi, k))
        rad = rng.uniform(0.2, 0.75) * ps * rng.uniform(0.6, 1.0, k)
        c = rng.uniform(100.0, 2000.0, 2)
        pts = np.stack([c[0] + rad * np.cos(ang), c[1] + rad * np.sin(ang)], 1)
        pts = np.concatenate([pts, pts[:1]]).astype(np.float32)
        rings.append(pts)
        off.append(off[-1] + len(pts))
    return np.asarray(off, np.uint32), np.concatenate(rings), rng.integers(0, 256, (n, 3, ps, ps), dtype=np.uint8)


@pytest.mark.parametrize("ps", [1, 2, 3, 4, 5, 7, 8, 11, 13, 15, 16])
def test_tiny_patch_sizes_match_the_oracle(nfx, synth, ps):
    n, B = 160, 32
    off, xy, pu8 = _inputs(synth, ps, n, 7)
    sets = [0, 1, 2, 3, 4]
    ctx = nfx.Context(0)
    cen, centred = ctx.preprocess_polygons(off, xy)
    masks = np.stack([orc.raster_polygon(centred[off[i]:off[i + 1]], ps) for i in range(n)]).reshape(n, ps, ps)
    got_masks = np.asarray(ctx.rasterize_polygons(off, centred, ps))
    assert np.array_equal(got_masks.reshape(n, ps, ps) != 0, masks != 0), "mask bits differ from the oracle's"
    got, gc = ctx.extract(sets, off, xy, pu8, ps, batch_size=B)
    want, wc = orc.extract(sets, off, xy, pu8, ps, batch_size=B)
    assert np.array_equal(gc, wc, equal_nan=True)
    par.assert_rows_close(got, want, par.column_scales(nfx, sets, ps), what=f"ps={ps}")
    gs, rb = ctx.debug_texture_counts(off, xy, pu8, ps)
    ws, wb = orc.texture_counts(off, xy, pu8, ps)
    assert np.array_equal(gs, ws) and np.array_equal(rb, wb), "texture counts differ from the oracle's"
    if ps >= 3:
        assert (masks != 0).any() and (~np.isnan(got[:, 12])).any(), "the scene must hold non-empty masks"
    # the same nuclei through the tile entry (gather at the byte path) on a small image
    H = W = max(64, 8 * ps)
    img = np.random.default_rng(ps).integers(0, 256, (3, H, W), dtype=np.uint8)
    xy2 = xy.copy()
    for i in range(n):       # move every ring into the image (some windows hang over its border)
        sl = slice(off[i], off[i + 1])
        xy2[sl] += (np.float32(3.0 + (i * 37) % (W - 4)) - cen[i, 0], np.float32(2.0 + (i * 53) % (H - 3)) - cen[i, 1])
    rows_t, cen_t = ctx.extract_tile(sets, off, xy2, img, ps, batch_size=B)
    patches = np.stack([orc.crop_patch(img, c[0], c[1], ps) for c in cen_t])
    want_t, _ = orc.extract(sets, off, xy2, patches, ps, batch_size=B)
    par.assert_rows_close(rows_t, want_t, par.column_scales(nfx, sets, ps), what=f"tile entry ps={ps}")
    ctx.close()
