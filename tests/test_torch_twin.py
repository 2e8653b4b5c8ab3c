"""Pins the oracle's restatement of the reference's IN-TREE arithmetic to what libtorch computes:
the twin (tests/_torch_twin.py) issues the ATen op sequence of shape.rs / color.rs / texture.rs on
torch-CPU.  Tolerance: 1e-5 relative (BASELINE.json north_star), exact for integer counts."""
import math

import numpy as np
import pytest
import torch

import _oracle as orc
import _torch_twin as tw

torch.set_num_threads(1)
RTOL = 1e-5


def _masks(synth, n, ps, seed=7):
    off, xy = synth.polygons(synth.SEED_BASE + seed, 0, n, ps)
    out = []
    for i in range(n):
        cen, cxy = orc.preprocess_polygon(xy[off[i]:off[i + 1]])
        out.append((cxy, orc.raster_polygon(cxy, ps)))
    return out


def _angle_close(a, b, tol):
    d = abs(a - b)
    return d <= tol or abs(d - 2 * math.pi) <= tol


@pytest.mark.parametrize("ps", [32, 48, 64])
def test_shape_moments_axes_orientation_vs_aten(synth, ps):
    n_checked = 0
    for cxy, mask in _masks(synth, 150, ps):
        out, aux = orc.shape_one(cxy, mask)
        m = torch.from_numpy(mask).unsqueeze(0)
        com = tw.center_of_mass(m)
        K = aux[0]
        assert com[0] == pytest.approx(aux[1] / K, rel=1e-6) and com[1] == pytest.approx(aux[2] / K, rel=1e-6)
        major, minor, angle, cov = tw.major_minor_axes_w_angle(m)
        # covariance entries the eig is fed with
        assert float(cov[0, 0]) == pytest.approx(aux[4], rel=RTOL)
        assert float(cov[1, 1]) == pytest.approx(aux[6], rel=RTOL)
        assert float(cov[0, 1]) == pytest.approx(aux[5], rel=RTOL, abs=1e-5 * max(aux[4], aux[6]))
        assert out[1] == pytest.approx(major, rel=RTOL)
        assert out[2] == pytest.approx(minor, rel=RTOL)
        assert out[3] == pytest.approx(tw.eccentricity(major, minor), rel=1e-4, abs=1e-4)
        p, q, r = aux[4], aux[5], aux[6]
        # orientation is ill-conditioned for near-isotropic nuclei: condition number ~ 1/hypot(p-r, 2q)
        cond = max(abs(p - r), 2 * abs(q)) / max(p, r)
        if cond > 1e-3:
            assert _angle_close(out[4], angle, 1e-5 / cond + 1e-6), (out[4], angle, p, q, r)
            n_checked += 1
        # elliptic deviation: ellipse rasterised by the oracle's restatement from the TWIN's parameters
        ell = orc.raster_ellipse(ps, com[1] - ps / 2.0, com[0] - ps / 2.0, major, minor, angle)
        ed = tw.eliptic_deviation(m, torch.from_numpy(ell).unsqueeze(0))
        # a 1-ulp difference in the axes can flip boundary pixels: allow 2 pixels
        assert abs(out[8] - ed) * K <= 2.0 + 1e-3
    assert n_checked > 100


def test_eig_closed_form_vs_lapack_random_cov():
    """slanv2 restatement vs torch.linalg.eig (MKL sgeev) on many masks incl. symmetric/degenerate ones."""
    rng = np.random.default_rng(3)
    ps = 32
    cases = []
    for _ in range(400):
        m = np.zeros((ps, ps), np.float32)
        k = rng.integers(1, 5)
        for _ in range(k):
            r0, c0 = rng.integers(0, ps - 2, 2)
            h, w = rng.integers(1, 14, 2)
            m[r0:r0 + h, c0:c0 + w] = 1
        cases.append(m)
    # exactly symmetric / degenerate shapes
    for m in (np.ones((ps, ps), np.float32), np.eye(ps, dtype=np.float32), np.fliplr(np.eye(ps, dtype=np.float32)).copy()):
        cases.append(m)
    line = np.zeros((ps, ps), np.float32); line[5, 3:20] = 1; cases.append(line)
    col = np.zeros((ps, ps), np.float32); col[3:20, 7] = 1; cases.append(col)
    one = np.zeros((ps, ps), np.float32); one[9, 9] = 1; cases.append(one)
    tri = np.zeros((ps, ps), np.float32)
    xy = np.zeros((1, 2), np.float32)
    for m in cases:
        out, aux = orc.shape_one(xy, m)
        res = tw.major_minor_axes_w_angle(torch.from_numpy(m).unsqueeze(0))
        major, minor, angle = res[0], res[1], res[2]
        assert out[1] == pytest.approx(major, rel=RTOL, abs=1e-6, nan_ok=True)
        assert out[2] == pytest.approx(minor, rel=RTOL, abs=2e-3, nan_ok=True)   # sqrt of a ~0 eigenvalue amplifies ulps
        p, q, r = aux[4], aux[5], aux[6]
        cond = max(abs(p - r), 2 * abs(q)) / max(p, r, 1e-30)
        if cond > 1e-3:
            assert _angle_close(out[4], angle, 1e-5 / cond + 1e-6), (out[4], angle, p, q, r)


def test_empty_mask_takes_nan_branch():
    m = np.zeros((32, 32), np.float32)
    out, aux = orc.shape_one(np.zeros((3, 2), np.float32), m)
    res = tw.major_minor_axes_w_angle(torch.from_numpy(m).unsqueeze(0))
    assert all(math.isnan(v) for v in res[:3])
    assert all(math.isnan(out[k]) for k in (1, 2, 3, 4, 8))


def _color_batch(synth, n, ps, seed=11):
    off, xy = synth.polygons(synth.SEED_BASE + seed, 0, n, ps)
    pu8 = synth.patches(synth.SEED_BASE + seed, 0, n, ps)
    masks = np.stack([orc.raster_polygon(orc.preprocess_polygon(xy[off[i]:off[i + 1]])[1], ps) for i in range(n)])
    patches = pu8.astype(np.float32) / np.float32(255.0)
    return off, xy, patches, masks


@pytest.mark.parametrize("n,ps", [(7, 32), (1, 48), (23, 64)])
def test_color_set_vs_aten(synth, n, ps):
    off, xy, patches, masks = _color_batch(synth, n, ps)
    got = orc.compute_features_batched(1, off, xy, patches, masks)
    p = torch.from_numpy(patches)
    m = torch.from_numpy(masks).unsqueeze(1)
    want = tw.color_features(p, m, tw.hsv_from_rgb_spec(p), tw.hed_from_rgb_spec(p)).numpy()
    scale = np.array([1, 1, 1, 1, 1, 1, 360, 1, 1, 360, 1, 1, 1, 1, 1, 1, 1, 1], np.float64)
    err = np.abs(got.astype(np.float64) - want) / (np.abs(want) + 1e-2 * scale)
    # std_h of a sharply peaked hue is the one place f32 accumulation order shows: 1e-4
    tol = np.full(18, 2e-5); tol[9] = 2e-4; tol[6] = 2e-5
    assert (err <= tol).all(), (np.argwhere(err > tol), err.max())


def test_color_mean_h_is_batch_coupled(synth):
    """color.rs:50-51,144-155: nucleus i's hue mean sums the hue of EVERY patch of its batch."""
    off, xy, patches, masks = _color_batch(synth, 6, 32)
    full = orc.compute_features_batched(1, off, xy, patches, masks)
    alone = orc.compute_features_batched(1, off[:2], xy[:off[1]], patches[:1], masks[:1])
    assert abs(full[0, 6] - alone[0, 6]) > 1e-3            # coupled
    np.testing.assert_allclose(full[0, [0, 1, 2, 3, 4, 5, 7, 8, 10, 11]], alone[0, [0, 1, 2, 3, 4, 5, 7, 8, 10, 11]], rtol=1e-6)


@pytest.mark.parametrize("ps", [32, 48])
def test_gabor_stats_vs_aten_conv2d(synth, ps):
    n = 3
    off, xy, patches, masks = _color_batch(synth, n, ps, seed=13)
    got = orc.compute_features_batched(4, off, xy, patches, masks)
    p = torch.from_numpy(patches)
    m = torch.from_numpy(masks).unsqueeze(1)
    gs = tw.gray(p)
    bank = torch.from_numpy(orc.gabor_bank()).unsqueeze(1)          # [48,1,30,30]
    filtered = torch.nn.functional.conv2d(torch.nn.functional.pad(gs, (15, 14, 15, 14)), bank)
    assert filtered.shape[-2:] == (ps, ps)
    want = tw.gabor_stats(filtered, m).numpy()
    amp = np.abs(orc.gabor_bank()).sum(axis=(1, 2))                 # natural scale of a response
    scale = np.repeat(amp, 2); scale[1::2] = scale[1::2] ** 2
    err = np.abs(got - want) / (np.abs(want) + 1e-3 * scale)
    assert err.max() <= 2e-4, err.max()   # f32 conv2d accumulates 900 terms in f32; oracle in f64


def test_gray_is_f32_sum_then_div(synth):
    pu8 = synth.patches(synth.SEED_BASE + 1, 0, 4, 32)
    p = torch.from_numpy(pu8.astype(np.float32) / np.float32(255.0))
    g = tw.gray(p).numpy()[:, 0]
    x = p.numpy()
    mine = ((x[:, 0] + x[:, 1]) + x[:, 2]) / np.float32(3.0)
    assert np.array_equal(g, mine)


@pytest.mark.parametrize("n,ps", [(5, 32), (3, 64)])
def test_dense_texture_programs_follow_the_oracles_definitions(synth, n, ps):
    """The libtorch-shaped baseline's tensor programs (dense [N,L,L] GLCM + 14 reductions, dense [N,24,16] GLRLM, conv2d Gabor) compute
    what the oracle defines: counts `==`, features to f32-dense-arithmetic accuracy."""
    off, xy, patches, masks = _color_batch(synth, n, ps, seed=17)
    p = torch.from_numpy(patches)
    m = torch.from_numpy(masks).unsqueeze(1)
    gs = tw.gray(p)
    for L, (dy, dx) in ((32, (0, 1)), (254, (1, -1)), (128, (1, 1))):
        P = tw.glcm_dense(gs, (dy, dx), L, m, True)
        for i in range(n):
            G = orc.glcm_counts(patches[i], masks[i], L, dy, dx)
            np.testing.assert_allclose(P[i].numpy() * G.sum(), G, atol=1e-3)
    for d, (dx, dy) in enumerate(tw.GLRLM_DIRECTIONS):
        R = tw.glrlm_dense(gs, 24, 16, (dx, dy), m)
        for i in range(n):
            assert np.array_equal(R[i].numpy(), orc.glrlm_counts(patches[i], masks[i], dx, dy))
    for set_id, got in ((2, tw.glcm_set(p, m)), (3, tw.glrlm_set(p, m))):
        want = orc.compute_features_batched(set_id, off, xy, patches, masks)
        err = np.abs(got.numpy() - want) / (np.abs(want) + 1e-3 * np.abs(want).max(axis=0, keepdims=True) + 1e-6)
        assert np.nanmax(err) <= 5e-3, (set_id, np.nanmax(err), np.unravel_index(np.nanargmax(err), err.shape))
