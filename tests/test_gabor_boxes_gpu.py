"""Gabor parity on masks whose bounding boxes sit where the kernel's strip geometry has its corner cases.

`k_gabor` cuts the mask's bounding box into four column strips (one per wave, placed so that they tile the plane's 64 columns
without overlap) and lays row strips over the lanes; the 0-degree filters run on the transposed nucleus.  The synthetic nuclei
of the other tests are star shapes around the patch centre, so this file feeds masks directly (the trait entry,
features/mod.rs:21-27: f32 patchs + f32 masks) that touch every border and corner, are one pixel thin, much wider than high
and the reverse, full-patch, scattered, or fewer columns wide than there are waves -- at patch sizes that are and are not
multiples of 16.  (The byte loaders of `nfx_extract` meet the same geometry code; they are covered by the star-shaped sets.)
"""
import numpy as np
import pytest

import _oracle as orc
import _parity as par

pytestmark = pytest.mark.gpu

GABOR = 4


def box_masks(ps, rng):
    m = {}

    def rect(r0, r1, c0, c1):
        a = np.zeros((ps, ps), np.float32)
        a[max(r0, 0):min(r1, ps), max(c0, 0):min(c1, ps)] = 1.0
        return a

    m["full"] = rect(0, ps, 0, ps)
    m["top row"] = rect(0, 1, 0, ps)
    m["bottom row"] = rect(ps - 1, ps, 0, ps)
    m["left column"] = rect(0, ps, 0, 1)
    m["right column"] = rect(0, ps, ps - 1, ps)
    m["right two columns"] = rect(3, ps - 5, ps - 2, ps)
    for nm, (r, c) in {"corner tl": (0, 0), "corner tr": (0, ps - 1), "corner bl": (ps - 1, 0), "corner br": (ps - 1, ps - 1)}.items():
        m[nm] = rect(r, r + 1, c, c + 1)
    h, w = max(2, ps // 12), (2 * ps) // 3
    m["wide bar top-right"] = rect(0, h, ps - w, ps)
    m["wide bar bottom-left"] = rect(ps - h, ps, 0, w)
    m["tall bar top-left"] = rect(0, w, 0, h)
    m["tall bar bottom-right"] = rect(ps - w, ps, ps - h, ps)
    q = ps // 4 + 1
    m["square br"] = rect(ps - q, ps, ps - q, ps)
    m["square tl"] = rect(0, q, 0, q)
    m["half right"] = rect(0, ps, ps // 2, ps)
    m["half bottom"] = rect(ps // 2, ps, 0, ps)
    m["three columns"] = rect(5, ps - 3, ps // 3, ps // 3 + 3)      # fewer columns than two waves' minimum strips
    m["nine columns"] = rect(2, ps - 2, ps - 9, ps)                 # the last wave's strip starts past the box
    yy, xx = np.mgrid[0:ps, 0:ps]
    m["diagonal"] = (yy == xx).astype(np.float32)
    m["anti-diagonal band"] = (np.abs(yy + xx - (ps - 1)) <= 1).astype(np.float32)
    m["checkerboard"] = ((yy + xx) & 1).astype(np.float32)
    m["frame"] = rect(0, ps, 0, ps) - rect(2, ps - 2, 2, ps - 2)
    m["two far pixels"] = rect(0, 1, 0, 1) + rect(ps - 1, ps, ps - 1, ps)
    for k in range(12):                                              # random boxes, randomly filled
        r0, c0 = rng.integers(0, ps - 1, 2)
        r1, c1 = rng.integers(r0 + 1, ps + 1), rng.integers(c0 + 1, ps + 1)
        a = rect(r0, r1, c0, c1) * (rng.random((ps, ps)) < rng.uniform(0.2, 1.0))
        if a.sum() == 0:
            a[r0, c0] = 1.0
        m[f"random box {k} [{r0}:{r1}, {c0}:{c1}]"] = a.astype(np.float32)
    m["empty"] = np.zeros((ps, ps), np.float32)                     # 0/0 -> NaN in texture.rs:340-344
    return m


@pytest.mark.parametrize("ps", [64, 48, 33, 20, 16])
def test_gabor_on_border_thin_and_lopsided_masks(nfx, ps):
    rng = np.random.default_rng(4000 + ps)
    masks = box_masks(ps, rng)
    names = list(masks)
    n = len(names)
    mk = np.stack([masks[k] for k in names])[:, None]                                       # [n,1,ps,ps]
    yy, xx = np.mgrid[0:ps, 0:ps]
    smooth = 96 + 60 * np.sin(xx / 3.1)[None] * np.cos(yy / 4.3)[None] + rng.integers(0, 40, (n, 3, ps, ps))
    u8 = np.clip(smooth, 0, 255).astype(np.uint8)
    u8[::5] = rng.integers(0, 256, u8[::5].shape, dtype=np.uint8)                           # every fifth patch is white noise
    patchs = (u8.astype(np.float32) / np.float32(255.0)).astype(np.float32)
    off = np.arange(0, 4 * n + 1, 4, dtype=np.uint32)                                       # the Gabor set ignores the rings
    ring = np.array([[-1, -1], [1, -1], [1, 1], [-1, -1]], np.float32)
    ctr = np.tile(ring, (n, 1))
    want = orc.compute_features_batched(GABOR, off, ctr, patchs, mk[:, 0])
    ctx = nfx.Context(0)
    try:
        got = ctx.compute_features_batched(GABOR, off, ctr, patchs, mk)
        par.assert_rows_close(got, want, par.column_scales(nfx, [GABOR], ps), what=f"gabor boxes ps={ps}", labels=names)
    finally:
        ctx.close()
    e = names.index("empty")
    assert np.isnan(got[e]).all()
