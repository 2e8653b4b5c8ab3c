"""Parity of the rasteriser and the geometry columns on rings built to sit on the decision boundaries: vertices and edges
exactly on pixel centres (the fill rule's strict / non-strict comparisons), clockwise and counter-clockwise order, repeated
vertices, zero-area and sub-pixel rings, self-intersections under the even-odd rule, slide-scale coordinates (f32 centroid
sums), rings longer than the polygon kernel's LDS staging, hulls with collinear points."""
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


def close(r):
    r = np.asarray(r, np.float64)
    return np.concatenate([r, r[:1]])


def rings(ps):
    out = {}
    for c in (512.0, 512.5, 1_000_003.25):   # pixel-centred, half-pixel and slide-scale centroids
        tag = f"@{c:g}"
        h = ps // 4
        sq = np.array([[-h, -h], [h, -h], [h, h], [-h, h]], np.float64)
        out["square ccw " + tag] = close(sq + c)
        out["square cw " + tag] = close(sq[::-1] + c)
        out["square half-integer " + tag] = close(sq + 0.5 + c)
        out["square unclosed " + tag] = sq + c
        out["repeated vertices " + tag] = close(np.repeat(sq, 3, axis=0) + c)
        out["diamond through centres " + tag] = close(np.array([[0, -h], [h, 0], [0, h], [-h, 0]], np.float64) + c)
        out["horizontal sliver on a centre row " + tag] = close(np.array([[-h, 0], [h, 0], [h, 1e-3], [-h, 1e-3]], np.float64) + c)
        out["zero area (back and forth) " + tag] = close(np.array([[-h, -h], [h, h]], np.float64) + c)
        out["sub-pixel " + tag] = close(np.array([[0.1, 0.1], [0.4, 0.1], [0.4, 0.4], [0.1, 0.4]], np.float64) + c)
        k = np.arange(5)
        star = np.stack([0.4 * ps * np.sin(2 * np.pi * 2 * k / 5), -0.4 * ps * np.cos(2 * np.pi * 2 * k / 5)], 1)
        out["pentagram " + tag] = close(star + c)
        cshape = np.array([[-h, -h], [h, -h], [h, -h / 2], [-h / 2, -h / 2], [-h / 2, h / 2], [h, h / 2], [h, h], [-h, h]], np.float64)
        out["C shape " + tag] = close(cshape + c)
        t = np.linspace(0, 2 * np.pi, 3001)[:-1]
        out["3000 vertices " + tag] = close(np.stack([0.35 * ps * np.cos(t) * (1 + 0.1 * np.cos(40 * t)), 0.3 * ps * np.sin(t)], 1) + c)
        grid = np.array([[x, -h] for x in range(-h, h + 1)] + [[h, y] for y in range(-h + 1, h + 1)] +
                        [[x, h] for x in range(h - 1, -h - 1, -1)] + [[-h, y] for y in range(h - 1, -h, -1)], np.float64)
        out["square with collinear points " + tag] = close(grid + c)
        out["larger than the patch " + tag] = close(sq * 8 + c)
        out["touching two corners " + tag] = close(np.array([[-ps / 2, -ps / 2], [ps / 2 - 1, ps / 2 - 1], [-ps / 2, ps / 2 - 1]], np.float64) + c)
    return out


@pytest.mark.parametrize("ps", [32, 47, 64])
def test_boundary_rings_match_the_oracle(nfx, ctx, ps):
    rs = rings(ps)
    names = list(rs)
    ring_list = [rs[k].astype(np.float32) for k in names]
    off = np.zeros(len(ring_list) + 1, np.uint32)
    np.cumsum([len(r) for r in ring_list], out=off[1:])
    xy = np.concatenate(ring_list)
    cen, ctr = ctx.preprocess_polygons(off, xy)                      # utils.rs:54-74
    got_masks = ctx.rasterize_polygons(off, ctr, ps)[:, 0]
    want_masks = []
    for i, nm in enumerate(names):
        wc, wctr = orc.preprocess_polygon(ring_list[i])
        assert np.array_equal(wc, cen[i]) and np.array_equal(wctr, ctr[off[i]:off[i + 1]]), f"ps={ps} {nm}: centroid / centred ring differ"
        want_masks.append(orc.raster_polygon(wctr, ps))
        assert np.array_equal(got_masks[i], want_masks[i]), f"ps={ps} {nm}: mask differs in {(got_masks[i] != want_masks[i]).sum()} pixels"
    assert any(m.sum() == 0 for m in want_masks) and any(m.sum() == ps * ps for m in want_masks)
    rng = np.random.default_rng(ps)
    pu8 = rng.integers(0, 256, (len(names), 3, ps, ps), dtype=np.uint8)
    sets = [0, 1, 2, 3, 4]
    want, wcen = orc.extract(sets, off, xy, pu8, ps, batch_size=16)
    got, gcen = ctx.extract(sets, off, xy, pu8, ps, batch_size=16)
    assert np.array_equal(gcen, wcen, equal_nan=True)
    par.assert_rows_close(got, want, par.column_scales(nfx, sets, ps), exact_cols=(1, 2, 3, 8), what=f"ps={ps}", labels=names)


@pytest.mark.parametrize("ps", [32, 48, 64])
def test_random_rings_of_any_shape(nfx, ctx, ps):
    """Rings nobody would draw: 0-80 uniformly random vertices in a box twice the patch (self-intersecting, mostly outside,
    many crossings per scanline), a third of them on integer or half-integer coordinates (ties), every feature set."""
    rng = np.random.default_rng(1234 + ps)
    n = 240
    ring_list = []
    for i in range(n):
        V = int(rng.integers(0, 81))
        c = rng.uniform(200.0, 5000.0, 2)
        pts = c + rng.uniform(-ps, ps, (V, 2)) * rng.uniform(0.05, 1.0)
        if i % 3 == 0:
            pts = np.round(pts * 2.0) / 2.0
        if V and i % 2 == 0:
            pts = np.concatenate([pts, pts[:1]])
        ring_list.append(pts.astype(np.float32))
    off = np.zeros(n + 1, np.uint32)
    np.cumsum([len(r) for r in ring_list], out=off[1:])
    xy = np.concatenate(ring_list).astype(np.float32)
    cen, ctr = ctx.preprocess_polygons(off, xy)
    got_masks = ctx.rasterize_polygons(off, ctr, ps)[:, 0]
    for i in range(n):
        _, wctr = orc.preprocess_polygon(ring_list[i])
        want = orc.raster_polygon(wctr, ps)
        assert np.array_equal(got_masks[i], want), f"ps={ps} ring {i} (V={len(ring_list[i])}): mask differs in {(got_masks[i] != want).sum()} pixels"
    pu8 = rng.integers(0, 256, (n, 3, ps, ps), dtype=np.uint8)
    sets = [0, 1, 2, 3, 4]
    want, wcen = orc.extract(sets, off, xy, pu8, ps, batch_size=50, threads=8)
    got, gcen = ctx.extract(sets, off, xy, pu8, ps, batch_size=50)
    assert np.array_equal(gcen, wcen, equal_nan=True)
    par.assert_rows_close(got, want, par.column_scales(nfx, sets, ps), exact_cols=(1, 2, 3, 8), what=f"ps={ps} random rings",
                          labels=[f"ring {i}, V={len(ring_list[i])}" for i in range(n)])
