"""Count-level parity of the texture sets (north_star: "bit-exact for integer/count features").

The GLCM and GLRLM columns are f32 functions of integer counts (pair counts of tch_utils::glcm::glcm, run counts of
tch_utils::glrlm::glrlm; call sites texture.rs:40-46,193-194).  The kernels keep those integers exactly -- per (level, offset):
the number of pairs, sum (a+b), sum (a+b)^2, sum |a-b|, sum (a-b)^2 and sum G^2 of the symmetric matrix; per direction: all 384
run-length bins -- and `nfx_debug_texture_counts` (test hook of the C ABI) returns them.  Here they are compared `==` with
the same integers derived from the oracle's dense matrices (orc_glcm_counts / orc_glrlm_counts) on the committed golden
sets (incl. their edge-case rings), on the extreme textures, and on one reference batch of every BASELINE config.
"""
import glob
import os

import numpy as np
import pytest

import _oracle as orc
import _parity as par
from test_texture_extremes_gpu import rings as extreme_rings
from test_texture_extremes_gpu import textures as extreme_textures

pytestmark = pytest.mark.gpu

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nfx_golden_ps*.npz")))


@pytest.fixture(scope="module")
def ctx(nfx):
    c = nfx.Context(0)
    yield c
    c.close()


def _assert_counts_equal(got, want, what):
    gg, gr = got
    wg, wr = want
    bad = np.argwhere(gg != wg)
    assert bad.size == 0, f"{what}: GLCM integers differ at [nucleus, level, offset, sum] {bad[:5].tolist()}: got {gg[tuple(bad[0])]} want {wg[tuple(bad[0])]}"
    bad = np.argwhere(gr != wr)
    assert bad.size == 0, f"{what}: GLRLM bins differ at [nucleus, direction, level, length-1] {bad[:5].tolist()}: got {gr[tuple(bad[0])]} want {wr[tuple(bad[0])]}"


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_counts_match_the_committed_golden_integers(ctx, path):
    g = np.load(path)
    got = ctx.debug_texture_counts(g["ring_offsets"], g["ring_xy"], g["patches_u8"], int(g["ps"]))
    _assert_counts_equal(got, (g["glcm_sums"], g["glrlm_bins"]), os.path.basename(path))
    # sanity of the fixture itself: a nucleus with pixels has pairs and runs
    assert int(g["glcm_sums"][0, :, :, 0].min()) > 0 and int(g["glrlm_bins"][0].sum()) > 0


@pytest.mark.parametrize("cfg,ps", [(1, 32), (2, 48), (4, 64)])
def test_counts_of_one_reference_batch_of_every_baseline_config(ctx, cfg, ps):
    off, xy, pu8 = par.make_inputs(cfg, 100, ps)     # seed index = BASELINE config index
    got = ctx.debug_texture_counts(off, xy, pu8, ps)
    _assert_counts_equal(got, orc.texture_counts(off, xy, pu8, ps), f"config seed {cfg} ps {ps}")


@pytest.mark.parametrize("ps", [32, 48, 64])
def test_counts_of_extreme_textures(ctx, ps):
    rng = np.random.default_rng(ps)
    ring_list, patches = [], []
    for _, p in extreme_textures(ps, rng).items():
        for r in extreme_rings(ps):
            ring_list.append(r)
            patches.append(p)
    off = np.zeros(len(ring_list) + 1, np.uint32)
    np.cumsum([len(r) for r in ring_list], out=off[1:])
    xy = np.concatenate(ring_list).astype(np.float32)
    pu8 = np.stack(patches)
    got = ctx.debug_texture_counts(off, xy, pu8, ps)
    want = orc.texture_counts(off, xy, pu8, ps)
    _assert_counts_equal(got, want, f"extreme textures ps {ps}")
    # constant patch under the full mask: ps (ps - 1) pairs in one diagonal cell, one run per row clipped at 16
    n_pairs = ps * (ps - 1)
    assert got[0][0, 0, 0].tolist() == want[0][0, 0, 0].tolist()
    assert int(got[0][0, 3, 0, 0]) == n_pairs and int(got[0][0, 3, 0, 5]) == (2 * n_pairs) ** 2
    assert int(got[1][0, 0].sum()) == ps and int(got[1][0, 0, :, 15].sum()) == ps


@pytest.mark.parametrize("ps", [64, 63, 48, 33, 17])
def test_counts_of_blocky_images_under_ragged_masks(ctx, ps):
    """Run structure in every direction: images of a few gray levels in blocks of 1..7 pixels (runs of every length from 1 up to
    the clip at 16, along rows, columns and both diagonals, across the word halves and -- at 64 -- round the wrap of the diagonal
    lines), under masks with many boundaries: stars, combs, and the full patch.  Every GLRLM bin and GLCM integer `==` the oracle's."""
    rng = np.random.default_rng(1000 + ps)
    c = ps / 2.0
    rings, patches = [], []
    def star(k, r0, r1, phase):
        a = np.arange(2 * k) * np.pi / k + phase
        r = np.where(np.arange(2 * k) % 2 == 0, r1, r0)
        return np.stack([c + r * np.cos(a), c + r * np.sin(a)], 1)
    def comb(teeth):
        pts = [(1.0, 1.0)]
        w = (ps - 2.0) / (2 * teeth)
        for t in range(teeth):
            x0 = 1.0 + 2 * t * w
            pts += [(x0, ps - 1.5), (x0 + w, ps - 1.5), (x0 + w, 4.0), (x0 + 2 * w, 4.0)]
        pts += [(ps - 1.0, 1.0)]
        return np.array(pts)
    shapes = [star(5, 0.18 * ps, 0.49 * ps, 0.3), star(9, 0.3 * ps, 0.5 * ps, 1.1), comb(3), comb(max(2, ps // 8)),
              np.array([[-ps, -ps], [2 * ps, -ps], [2 * ps, 2 * ps], [-ps, 2 * ps]], float)]
    for shape in shapes:
        for block in (1, 2, 3, 5, 7):
            levels = rng.integers(0, 4, size=((ps + block - 1) // block + 1, (ps + block - 1) // block + 1))
            oy, ox = rng.integers(0, block, 2)
            img = np.kron(levels, np.ones((block, block), int))[oy:oy + ps, ox:ox + ps]
            gray = (img * 60 + 20).astype(np.uint8)         # four levels far apart on every quantisation scale
            if block == 3:                                  # ... and one diagonal-striped variant: runs along the diagonals only
                yy, xx = np.mgrid[0:ps, 0:ps]
                gray = ((((yy + xx) // 2) % 3) * 80 + 10).astype(np.uint8)
            if block == 5:
                yy, xx = np.mgrid[0:ps, 0:ps]
                gray = ((((yy - xx) // 3) % 3) * 80 + 10).astype(np.uint8)
            rings.append(np.concatenate([shape, shape[:1]]).astype(np.float32))
            patches.append(np.repeat(gray[None], 3, 0))
    off = np.zeros(len(rings) + 1, np.uint32)
    np.cumsum([len(r) for r in rings], out=off[1:])
    xy = np.concatenate(rings).astype(np.float32)
    pu8 = np.stack(patches)
    got = ctx.debug_texture_counts(off, xy, pu8, ps)
    want = orc.texture_counts(off, xy, pu8, ps)
    _assert_counts_equal(got, want, f"blocky images ps {ps}")
    assert int(want[1].sum()) > 0 and int(want[1][:, :, :, 15].sum()) > 0 and int(want[1][:, 1].sum()) > 0  # clipped runs and diagonal runs are there


def test_count_outputs_are_optional(ctx):
    off, xy, pu8 = par.make_inputs(2, 8, 48)
    g, r = ctx.debug_texture_counts(off, xy, pu8, 48, glrlm=False)
    assert r is None and g.shape == (8, 4, 4, 6)
    g2, r2 = ctx.debug_texture_counts(off, xy, pu8, 48, glcm=False)
    assert g2 is None and r2.shape == (8, 4, 24, 16)
    assert np.array_equal(g, ctx.debug_texture_counts(off, xy, pu8, 48)[0])
