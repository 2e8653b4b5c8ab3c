"""The HIP trait entry against numbers ATen itself produced -- no oracle in between.

tests/golden/twin_ps{32,48,64}.npz (made by tests/golden/make_twin_golden.py in the build container) hold, for two reference
batches per patch size, the outputs of the reference's IN-TREE arithmetic issued op for op on torch-CPU (same ATen CPU kernels
and MKL LAPACK that `tch` reaches): shape.rs:137-226, color.rs:117-155, texture.rs:332-344.  The dependency-resident inputs of
that arithmetic (masks, ellipse masks, hsv / hed channels, filter bank) are data in the fixture (see the generator's header).

Tolerances, stated per column group (north_star: 1e-5 relative on floats, == on counts):
  * eliptic_deviation: integer XOR count / integer area, one f32 division: ==.
  * major_axis, minor_axis: 1e-6 relative (ATen's f32 `mm` sums the covariance in BLAS order, the kernel rounds the exact
    integer moments once: a few last places of the covariance are allowed, nothing more; measured 3.4e-7).
  * eccentricity = sqrt(M^2 - m^2) / M amplifies a relative error eps of the axes to eps M^2 / (M^2 - m^2): tolerance
    1e-5 x max(1, M^2 / (M^2 - m^2)) relative; orientation (atan2 of an eigenvector row) likewise by the covariance's
    conditioning max(p, r) / max(|p - r|, 2|q|) -- both factors come from the fixture's own covariance, not from the result.
  * colour and Gabor columns: 1e-5 relative with tests/_parity.py's floors (batch median; absolute for mean_h and the Gabor means).
"""
import glob
import os

import numpy as np
import pytest

import _parity as par

pytestmark = pytest.mark.gpu

FIXTURES = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "twin_ps*.npz")))
RTOL = 1e-5


def _load(path):
    d = np.load(path)
    ps, B = int(d["ps"]), int(d["batch_size"])
    off = d["poly_offsets"].astype(np.uint32)
    masks = np.unpackbits(d["mask_bits"], axis=-1)[..., :ps].astype(np.float32)[:, None]
    patchs = d["patches_u8"].astype(np.float32) / np.float32(255.0)           # utils.rs:172
    return d, ps, B, off, d["poly_xy"], patchs, masks


def _trait(nfx, ctx, set_id, off, xy, patchs, masks, B):
    """the trait entry, one reference batch per call (features/mod.rs:21-27)"""
    n = len(off) - 1
    out = []
    for b0 in range(0, n, B):
        b1 = min(n, b0 + B)
        o = (off[b0:b1 + 1] - off[b0]).astype(np.uint32)
        out.append(ctx.compute_features_batched(set_id, o, xy[off[b0]:off[b1]], None if patchs is None else patchs[b0:b1], masks[b0:b1]))
    return np.concatenate(out)


def test_fixtures_present():
    assert len(FIXTURES) == 3, "tests/golden/twin_ps{32,48,64}.npz are committed fixtures"


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_shape_columns_against_aten(nfx, path):
    d, ps, B, off, xy, patchs, masks = _load(path)
    ctx = nfx.Context(0)
    got = _trait(nfx, ctx, nfx.GEOMETRY, off, xy, None, masks, B).astype(np.float64)
    ctx.close()
    want = d["twin_shape"].astype(np.float64)          # major, minor, eccentricity, orientation, eliptic_deviation
    p, q, r = d["twin_cov"].astype(np.float64).T
    assert np.array_equal(got[:, 8].astype(np.float32), want[:, 4].astype(np.float32)), "eliptic_deviation: integer counts, one division"
    for col, k in ((1, 0), (2, 1)):
        rel = np.abs(got[:, col] - want[:, k]) / np.abs(want[:, k])
        assert rel.max() <= 1e-6, (col, rel.max())
    M2, m2 = want[:, 0] ** 2, want[:, 1] ** 2
    amp = np.maximum(1.0, M2 / np.maximum(M2 - m2, 1e-30))
    rel = np.abs(got[:, 3] - want[:, 2]) / np.maximum(np.abs(want[:, 2]), 1e-30)
    assert (rel <= RTOL * amp).all(), ("eccentricity", (rel / amp).max())
    cond = np.maximum(p, r) / np.maximum(np.maximum(np.abs(p - r), 2 * np.abs(q)), 1e-30)
    dang = np.abs(got[:, 4] - want[:, 3])
    dang = np.minimum(dang, np.abs(dang - 2 * np.pi))
    assert (dang <= RTOL * np.maximum(1.0, cond)).all(), ("orientation", (dang / np.maximum(1.0, cond)).max())
    assert cond.max() < 1e3, "the fixture holds no isotropic nucleus whose angle would go unchecked"


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_colour_columns_against_aten(nfx, path):
    d, ps, B, off, xy, patchs, masks = _load(path)
    ctx = nfx.Context(0)
    got = _trait(nfx, ctx, nfx.COLOR, off, xy, patchs, masks, B)
    # the batch partition is part of the result (color.rs:144-155 couples the hue mean of a batch's nuclei): one call over both
    # batches gives a different mean_h, the partition the fixture was made with gives the fixture's
    whole = ctx.compute_features_batched(nfx.COLOR, off, xy, patchs, masks)
    ctx.close()
    worst = par.assert_rows_close(got, d["twin_color"], par.column_scales(nfx, [nfx.COLOR], ps), what=f"colour vs ATen ps={ps}")
    assert worst <= 1.0 * RTOL
    assert np.abs(whole[:, 6] - d["twin_color"][:, 6]).max() > 1e-3, "mean_h must depend on the batch partition"


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_gabor_statistics_against_aten_conv2d(nfx, path):
    d, ps, B, off, xy, patchs, masks = _load(path)
    ctx = nfx.Context(0)
    got = _trait(nfx, ctx, nfx.GABOR, off, xy, patchs, masks, B)
    ctx.close()
    par.assert_rows_close(got, d["twin_gabor"], par.column_scales(nfx, [nfx.GABOR], ps), what=f"gabor vs ATen conv2d ps={ps}")
