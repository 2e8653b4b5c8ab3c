"""Independent cross-checks of the oracle's BUILD DECISIONS (the arithmetic of tch-utils and
geometric-features is not in the reference tree, so parity there is unpinned).  These do not pin parity -- nothing here
comes from the reference -- they catch restatement slips: every dependency-resident function of the oracle is compared
with a second implementation written from the published formula or taken from a standard library, not from
oracle/nfx_oracle.c:

    hsv_from_rgb          colorsys.rgb_to_hsv (CPython standard library)
    hed_from_rgb          scikit-image's rgb2hed formula in numpy f64 (Ruifrok & Johnston stain matrix)
    polygon area/perim.   numpy shoelace / edge lengths;  convex hull: scipy.spatial.ConvexHull (Qhull)
    shapes::polygon       numpy crossing-number test over the whole pixel grid at once
    glcm + Haralick-14    dense numpy matrix via np.add.at and the textbook sums
    glrlm + 17 features   run extraction by walking every line of the direction in Python, textbook sums
    gabor bank + stats    numpy outer-product bank, scipy.signal.correlate2d, masked mean / variance
CPU only.
"""
import colorsys
import ctypes as C

import numpy as np
import pytest

import _oracle as orc
import _parity as par


def _f3(fn, r, g, b):
    out = np.zeros(3, np.float32)
    fn(C.c_float(r), C.c_float(g), C.c_float(b), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def test_hsv_against_colorsys():
    rng = np.random.default_rng(1)
    cols = rng.integers(0, 256, (2000, 3))
    cols[:6] = [[0, 0, 0], [255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [17, 17, 17]]
    for r8, g8, b8 in cols:
        r, g, b = (np.float32(v) / np.float32(255.0) for v in (r8, g8, b8))
        h, s, v = _f3(orc.lib().orc_hsv_pixel, r, g, b)
        wh, ws, wv = colorsys.rgb_to_hsv(float(r), float(g), float(b))
        assert abs(float(s) - ws) <= 1e-6 and abs(float(v) - wv) <= 1e-6
        dh = abs(float(h) - 360.0 * wh)
        assert min(dh, 360.0 - dh) <= 2e-4, (r8, g8, b8, h, 360.0 * wh)       # degrees; f32 quotient of two differences


def test_hed_against_the_skimage_formula():
    # scikit-image rgb2hed (skimage/color/colorconv.py): stains = (log(max(rgb, 1e-6)) / log(1e-6)) @ inv(rgb_from_hed), >= 0
    rgb_from_hed = np.array([[0.65, 0.70, 0.29], [0.07, 0.99, 0.11], [0.27, 0.57, 0.78]])
    hed_from_rgb = np.linalg.inv(rgb_from_hed)
    rng = np.random.default_rng(2)
    cols = rng.integers(0, 256, (2000, 3))
    cols[:3] = [[0, 0, 0], [255, 255, 255], [1, 128, 254]]
    for r8, g8, b8 in cols:
        rgb = np.array([r8, g8, b8], np.float64) / 255.0
        want = np.maximum((np.log(np.maximum(rgb, 1e-6)) / np.log(1e-6)) @ hed_from_rgb, 0.0)
        got = _f3(orc.lib().orc_hed_pixel, *(np.float32(v) / np.float32(255.0) for v in (r8, g8, b8)))
        assert np.allclose(got, want, rtol=1e-5, atol=2e-6), (r8, g8, b8, got, want)


def _random_rings(rng, n, ps):
    for _ in range(n):
        V = int(rng.integers(3, 50))
        t = np.sort(rng.uniform(0, 2 * np.pi, V))
        rad = ps * rng.uniform(0.1, 0.45) * (1 + 0.4 * rng.uniform(-1, 1, V))
        ring = np.stack([rad * np.cos(t), rad * np.sin(t)], 1)
        yield np.concatenate([ring, ring[:1]]).astype(np.float32)     # closed, centred on the origin


def test_polygon_columns_against_numpy_and_qhull():
    scipy_spatial = pytest.importorskip("scipy.spatial")
    rng = np.random.default_rng(3)
    for ring in _random_rings(rng, 300, 64):
        got = orc.polygon_features_f64(ring)
        p = ring.astype(np.float64)
        x, y = p[:, 0], p[:, 1]
        area = 0.5 * abs(np.sum(x * np.roll(y, -1) - np.roll(x, -1) * y))          # shoelace over the listed (closed) ring
        perim = np.sum(np.hypot(np.roll(x, -1) - x, np.roll(y, -1) - y))
        hull = scipy_spatial.ConvexHull(p[:-1])                                     # 2-D: .volume is the area, .area the perimeter
        want = [area, perim, 2 * np.sqrt(np.pi * area), 4 * np.pi * area / perim ** 2, hull.volume, (hull.volume - area) / hull.volume, hull.area]
        assert np.allclose(got, want, rtol=1e-9, atol=1e-9), (got, want)


def test_polygon_raster_against_a_vectorised_crossing_number():
    rng = np.random.default_rng(4)
    ps = 48
    for ring in _random_rings(rng, 60, ps):
        got = orc.raster_polygon(ring, ps)
        # even-odd rule at integer pixel coordinates (c - ps/2, r - ps/2): count edge crossings of the ray to +x
        py, px = np.mgrid[0:ps, 0:ps].astype(np.float64) - ps / 2.0
        x1, y1 = ring[:-1, 0].astype(np.float64), ring[:-1, 1].astype(np.float64)
        x2, y2 = ring[1:, 0].astype(np.float64), ring[1:, 1].astype(np.float64)
        inside = np.zeros((ps, ps), bool)
        for a, b, c, d in zip(x1, y1, x2, y2):
            straddles = (b > py) != (d > py)
            with np.errstate(divide="ignore", invalid="ignore"):
                xint = (c - a) * (py - b) / (d - b) + a
            inside ^= straddles & (px < xint)
        assert np.array_equal(got.astype(bool), inside)


def _gray_levels(patch_u8, L):
    f = patch_u8.astype(np.float32) / np.float32(255.0)
    gray = ((f[0] + f[1]) + f[2]) / np.float32(3.0)
    return np.minimum(np.floor(gray * np.float32(L)).astype(np.int64), L - 1)


def _haralick(P):
    L = P.shape[0]
    i, j = np.mgrid[0:L, 0:L]
    px = P.sum(1)
    mu = (np.arange(L) * px).sum()
    var = (((np.arange(L) - mu) ** 2) * px).sum()
    k_sum = np.arange(2 * L - 1)
    pxy = np.bincount((i + j).ravel(), weights=P.ravel(), minlength=2 * L - 1)
    pd = np.bincount(np.abs(i - j).ravel(), weights=P.ravel(), minlength=L)
    xlogx = lambda p: np.where(p > 0, p * np.log(np.where(p > 0, p, 1.0)), 0.0)   # noqa: E731
    ent = -xlogx(P).sum()
    sa = (k_sum * pxy).sum()
    da = (np.arange(L) * pd).sum()
    pp = np.outer(px, px)
    lg = np.where(pp > 0, np.log(np.where(pp > 0, pp, 1.0)), 0.0)
    hxy1, hxy2, hx = -(P * lg).sum(), -(pp * lg).sum(), -xlogx(px).sum()
    return np.array([((i * j * P).sum() - mu * mu) / var, (np.arange(L) ** 2 * pd).sum(), da, ent, (P * P).sum(), sa,
                     ((k_sum - sa) ** 2 * pxy).sum(), -xlogx(pxy).sum(), var, (P / (1.0 + (i - j) ** 2)).sum(), da,
                     ((np.arange(L) - da) ** 2 * pd).sum(), (ent - hxy1) / hx, np.sqrt(max(0.0, 1.0 - np.exp(-2.0 * (hxy2 - ent))))])


def test_glcm_against_a_dense_numpy_matrix():
    ps, n = 32, 6
    off, xy, pu8 = par.make_inputs(41, n, ps)
    got, _ = orc.extract([2], off, xy, pu8, ps)
    for i in range(n):
        mask = orc.raster_polygon(orc.preprocess_polygon(xy[off[i]:off[i + 1]])[1], ps).astype(bool)
        col = 0
        for L in (32, 64, 128, 254):
            q = _gray_levels(pu8[i], L)
            for dy, dx in ((0, 1), (1, 1), (1, 0), (1, -1)):
                r, c = np.mgrid[0:ps, 0:ps]
                r2, c2 = r + dy, c + dx
                ok = (r2 >= 0) & (r2 < ps) & (c2 >= 0) & (c2 < ps)
                ok[ok] &= mask[r[ok], c[ok]] & mask[r2[ok], c2[ok]]
                G = np.zeros((L, L))
                np.add.at(G, (q[r[ok], c[ok]], q[r2[ok], c2[ok]]), 1.0)
                G = G + G.T
                want = _haralick(G / G.sum())
                assert np.allclose(got[i, col:col + 14], want, rtol=2e-5, atol=1e-6), (i, L, dy, dx, got[i, col:col + 14], want)
                col += 14


def _runs(q, mask, dx, dy):
    """maximal runs of equal level along (dx, dy) inside the mask: list of (level, length)"""
    ps = q.shape[0]
    out = []
    for r in range(ps):
        for c in range(ps):
            if not mask[r, c]:
                continue
            rp, cp = r - dy, c - dx
            if 0 <= rp < ps and 0 <= cp < ps and mask[rp, cp] and q[rp, cp] == q[r, c]:
                continue
            length, rn, cn = 1, r + dy, c + dx
            while 0 <= rn < ps and 0 <= cn < ps and mask[rn, cn] and q[rn, cn] == q[r, c]:
                length, rn, cn = length + 1, rn + dy, cn + dx
            out.append((int(q[r, c]), length))
    return out


def test_glrlm_against_python_run_walks():
    ps, n = 32, 4
    off, xy, pu8 = par.make_inputs(42, n, ps)
    got, _ = orc.extract([3], off, xy, pu8, ps)
    for i in range(n):
        mask = orc.raster_polygon(orc.preprocess_polygon(xy[off[i]:off[i + 1]])[1], ps).astype(bool)
        q = _gray_levels(pu8[i], 24)
        for d, (dx, dy) in enumerate(((1, 0), (1, 1), (0, 1), (-1, 1))):
            R = np.zeros((24, 16))
            for lvl, length in _runs(q, mask, dx, dy):
                R[lvl, min(length, 16) - 1] += 1
            g = np.arange(1, 25, dtype=np.float64)[:, None]
            l = np.arange(1, 17, dtype=np.float64)[None, :]
            nr, npix = R.sum(), mask.sum()
            e = (g - 12.5) ** 2                                  # "extreme" weight; "mid" weight 1 / (1 + e)  
            mean = (R * l).sum() / nr
            want = [(R / l ** 2).sum() / nr, (R * l ** 2).sum() / nr, (R.sum(1) ** 2).sum() / nr, (R.sum(0) ** 2).sum() / nr,
                    (R / g ** 2).sum() / nr, (R * g ** 2).sum() / nr, (R / (l ** 2 * g ** 2)).sum() / nr, (R * g ** 2 / l ** 2).sum() / nr,
                    (R * l ** 2 / g ** 2).sum() / nr, (R * l ** 2 * g ** 2).sum() / nr, (R / (l ** 2 * (1 + e))).sum() / nr,
                    (R * l ** 2 / (1 + e)).sum() / nr, (R * e / l ** 2).sum() / nr, (R * l ** 2 * e).sum() / nr, nr / npix, mean,
                    (R * (l - mean) ** 2).sum() / nr]
            assert np.allclose(got[i, 17 * d:17 * d + 17], want, rtol=1e-5, atol=1e-7), (i, d, got[i, 17 * d:17 * d + 17], want)


def test_gabor_against_scipy_correlate():
    signal = pytest.importorskip("scipy.signal")
    ps, n = 32, 3
    off, xy, pu8 = par.make_inputs(43, n, ps)
    got, _ = orc.extract([4], off, xy, pu8, ps)
    t = -1.0 + 2.0 * np.arange(30) / 29.0
    yy, xx = np.meshgrid(t, t, indexing="ij")
    env = np.exp(-(xx ** 2 + yy ** 2) / (2 * 0.45 ** 2))
    for i in range(n):
        mask = orc.raster_polygon(orc.preprocess_polygon(xy[off[i]:off[i + 1]])[1], ps).astype(bool)
        f = pu8[i].astype(np.float32) / np.float32(255.0)
        gray = (((f[0] + f[1]) + f[2]) / np.float32(3.0)).astype(np.float64)
        for a in range(8):
            for fi, freq in enumerate((0.5, 1.0, 2.0, 4.0, 6.0, 8.0)):
                th = a * np.pi / 4
                w = env * np.cos(2 * np.pi * freq * (xx * np.cos(th) + yy * np.sin(th)))
                padded = np.zeros((ps + 29, ps + 29))
                padded[15:15 + ps, 15:15 + ps] = gray                               # 15 before, 14 after
                resp = signal.correlate2d(padded, w, mode="valid")
                vals = resp[mask]
                j = a * 6 + fi
                assert abs(got[i, 2 * j] - vals.mean()) <= 1e-5 * max(abs(vals.mean()), 0.5), (i, a, freq, got[i, 2 * j], vals.mean())
                assert abs(got[i, 2 * j + 1] - vals.var()) <= 1e-5 * max(vals.var(), 5.0), (i, a, freq)
