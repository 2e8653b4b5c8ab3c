"""Parity on textures chosen to hit the limits of the counting structures rather than typical H&E statistics:

  constant patches      every co-occurring pair falls into ONE diagonal GLCM cell (rank up to ps(ps-1)-1, the end of the
                        rank tables), one GLRLM run per row clipped at the maximum run length, zero Gabor variance,
                        zero saturation (hue is defined as 0), log(0) in the stain separation for black
  white noise           as many distinct GLCM cells as pairs (hash tables at their design load), runs of length 1
  0/255 checkerboards   two cells only, maximal contrast, alternating runs
  one-level ramps       neighbouring levels differ by exactly one quantisation step along x
each under a full-patch mask (the largest size class), a typical nucleus and a one-pixel mask.
"""
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


def textures(ps, rng):
    yy, xx = np.mgrid[0:ps, 0:ps]
    out = {}
    for v in (0, 1, 127, 254, 255):
        out[f"constant {v}"] = np.full((3, ps, ps), v, np.uint8)
    out["noise"] = rng.integers(0, 256, (3, ps, ps), dtype=np.uint8)
    out["gray noise"] = np.repeat(rng.integers(0, 256, (1, ps, ps), dtype=np.uint8), 3, 0)
    out["checkerboard"] = np.repeat((((yy + xx) & 1) * 255).astype(np.uint8)[None], 3, 0)
    out["stripes"] = np.repeat(((yy & 1) * 255).astype(np.uint8)[None], 3, 0)
    out["ramp x"] = np.repeat((xx * 255 // (ps - 1)).astype(np.uint8)[None], 3, 0)
    out["ramp y, one channel"] = np.stack([(yy * 255 // (ps - 1)).astype(np.uint8), np.zeros((ps, ps), np.uint8), np.full((ps, ps), 255, np.uint8)])
    out["two levels"] = np.repeat(np.where(xx < ps // 2, 100, 101).astype(np.uint8)[None], 3, 0)
    return out


def rings(ps):
    c = 700.0
    t = np.linspace(0, 2 * np.pi, 33)
    return [
        np.array([[c - 2 * ps, c - 2 * ps], [c + 2 * ps, c - 2 * ps], [c + 2 * ps, c + 2 * ps], [c - 2 * ps, c + 2 * ps], [c - 2 * ps, c - 2 * ps]], np.float32),  # full mask
        np.stack([c + 0.3 * ps * np.cos(t), c + 0.22 * ps * np.sin(t)], 1).astype(np.float32),                                                               # typical
        np.array([[c - 0.5, c - 0.5], [c + 0.5, c - 0.5], [c + 0.5, c + 0.5], [c - 0.5, c + 0.5], [c - 0.5, c - 0.5]], np.float32),                          # one pixel
    ]


@pytest.mark.parametrize("ps", [32, 48, 64])
def test_extreme_textures_match_the_oracle(nfx, ctx, ps):
    rng = np.random.default_rng(ps)
    tex = textures(ps, rng)
    rs = rings(ps)
    names, ring_list, patches = [], [], []
    for nm, p in tex.items():
        for k, r in enumerate(rs):
            names.append(f"{nm} / ring {k}")
            ring_list.append(r)
            patches.append(p)
    off = np.zeros(len(ring_list) + 1, np.uint32)
    np.cumsum([len(r) for r in ring_list], out=off[1:])
    xy = np.concatenate(ring_list).astype(np.float32)
    pu8 = np.stack(patches)
    sets = [0, 1, 2, 3, 4]
    batch = 7  # ragged batches mix textures inside the batch-coupled hue mean
    want, wcen = orc.extract(sets, off, xy, pu8, ps, batch_size=batch)
    got, gcen = ctx.extract(sets, off, xy, pu8, ps, batch_size=batch)
    assert np.array_equal(gcen, wcen, equal_nan=True)
    par.assert_rows_close(got, want, par.column_scales(nfx, sets, ps), what=f"ps={ps}", labels=names)  # failures name the texture
    # the counting features of a constant patch under the full mask are known in closed form: one cell, p = 1
    cols = nfx.all_column_names(sets)[1:]
    i = names.index("constant 127 / ring 0")
    for L in (32, 64, 128, 254):
        assert got[i, cols.index(f"angular_second_moment_0_1_{L}")] == 1.0
        assert got[i, cols.index(f"entropy_0_1_{L}")] == 0.0
        assert got[i, cols.index(f"contrast_0_1_{L}")] == 0.0


@pytest.mark.parametrize("ps", [32, 64])
def test_extreme_textures_through_the_trait_entry(nfx, ctx, ps):
    """The same textures as f32 tensors through `compute_features_batched` (one set, one batch of f32 patchs + f32 masks):
    the kernels take their f32 loaders here (no byte staging, no exact-division shortcuts), and the batch-coupled hue
    mean sees all textures in ONE batch, as a reference batch would."""
    rng = np.random.default_rng(100 + ps)
    tex = textures(ps, rng)
    rs = rings(ps)
    names, ring_list, patches = [], [], []
    for nm, p in tex.items():
        for k, r in enumerate(rs):
            names.append(f"{nm} / ring {k}")
            ring_list.append(r)
            patches.append(p)
    off = np.zeros(len(ring_list) + 1, np.uint32)
    np.cumsum([len(r) for r in ring_list], out=off[1:])
    xy = np.concatenate(ring_list).astype(np.float32)
    patchs = (np.stack(patches).astype(np.float32) / np.float32(255.0)).astype(np.float32)
    patchs[1::7] = np.clip(patchs[1::7] * np.float32(1.0001), 0, 1)   # some values that are not k / 255
    cen, ctr = ctx.preprocess_polygons(off, xy)
    masks = ctx.rasterize_polygons(off, ctr, ps)
    for set_id in range(5):
        want = orc.compute_features_batched(set_id, off, ctr, patchs, masks[:, 0])
        got = ctx.compute_features_batched(set_id, off, ctr, patchs, masks)
        par.assert_rows_close(got, want, par.column_scales(nfx, [set_id], ps), what=f"trait {nfx.feature_set_cli_name(set_id)} ps={ps}", labels=names)


@pytest.mark.parametrize("seed_idx,ps,i,column,dy,dx", [(7, 17, 1628, "correlation_1_1_32", 1, 1), (8, 24, 2391, "correlation_1_-1_32", 1, -1)])
def test_an_exactly_uncorrelated_glcm_gives_zero_up_to_f64_noise(nfx, synth, ctx, seed_idx, ps, i, column, dy, dx):
    """The column behind "worst pure relative 2.0 at ps 17 / 1.0 at ps 24" of the GLCM error profiles (named by scripts/err_profile.py,
    profiles/r05_err_profile.txt): `correlation` at 32 levels of a small nucleus whose numerator sum(ij P) - mu^2 is ZERO in exact
    arithmetic (checked here on the integer matrix: T sum(ij G) == (sum(i G))^2).  The oracle evaluates the cancellation to +7.8e-15
    (resp. -4.4e-15), the kernel to -7.8e-15 (resp. 0): one ulp of mu^2 over the variance either way -- a true 0-vs-epsilon fourteen
    orders below the column's values (median 0.2), not a sign error.  Every other GLCM column of the nucleus is within 1e-5."""
    seed = synth.SEED_BASE + seed_idx
    off, xy = synth.polygons(seed, i, 1, ps)
    pu8 = synth.patches(seed, i, 1, ps)
    want, _ = orc.extract([2], off, xy, pu8, ps, batch_size=1)
    got, _ = ctx.extract([2], off, xy, pu8, ps, batch_size=1)
    cols = nfx.all_column_names([2])[1:]
    c = cols.index(column)
    mask = orc.raster_polygon(orc.preprocess_polygon(xy)[1], ps)
    G = orc.glcm_counts(pu8[0].astype(np.float32) / np.float32(255.0), mask, 32, dy, dx).astype(object)    # exact integers
    T = int(G.sum())
    lv = np.arange(32, dtype=object)
    s_ij = int((G * np.outer(lv, lv)).sum())
    s_i = int((G.sum(1) * lv).sum())
    assert T > 0 and T * s_ij == s_i * s_i, "the numerator of this nucleus's correlation is exactly zero"
    mu = s_i / T
    var = float(((lv.astype(np.float64) - mu) ** 2 * G.sum(1).astype(np.float64)).sum() / T)
    noise = 4 * np.finfo(np.float64).eps * mu * mu / var                # a few ulps of mu^2, scaled as the division scales them
    assert abs(float(want[0, c])) <= noise and abs(float(got[0, c])) <= noise, (want[0, c], got[0, c], noise)
    other = [j for j in range(len(cols)) if j != c]
    par.assert_rows_close(got[:, other], want[:, other], np.asarray(par.column_scales(nfx, [2], ps))[other], what=f"glcm of nucleus {i} at ps {ps}")
