"""Parity of the HIP rasteriser + geometry set against the CPU oracle, through the C ABI.
Integer results (mask bits, eliptic_deviation's XOR count / K) must be bit-exact; floats 1e-5."""
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


def _rings(ps, n=300, seed=2):
    off, xy, _ = par.make_inputs(seed, n, ps, with_patches=False)
    return par.concat_rings((off, xy), par.edge_case_rings(ps))


@pytest.mark.parametrize("ps", [32, 48, 64, 17])
def test_preprocess_and_raster_bit_exact(ctx, ps):
    off, xy = _rings(ps)
    n = len(off) - 1
    cen, ctr = ctx.preprocess_polygons(off, xy)
    masks, rows = ctx.rasterize_polygons(off, ctr, ps, rows=True)
    for i in range(n):
        c0, c1 = orc.preprocess_polygon(xy[off[i]:off[i + 1]])
        assert np.array_equal(c0, cen[i], equal_nan=True), i
        assert np.array_equal(c1, ctr[off[i]:off[i + 1]], equal_nan=True), i
        want = orc.raster_polygon(c1, ps)
        assert np.array_equal(masks[i, 0], want), f"mask {i} differs in {(masks[i,0]!=want).sum()} pixels"
        bits = ((rows[i][:, None] >> np.arange(ps, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(np.float32)
        assert np.array_equal(bits, want)
    assert masks.sum() > 0


@pytest.mark.parametrize("ps", [32, 48, 64])
def test_geometry_set_pipeline_entry(nfx, ctx, ps):
    off, xy = _rings(ps)
    n = len(off) - 1
    dummy = np.zeros((n, 3, ps, ps), np.uint8)
    want, wcen = orc.extract([0], off, xy, dummy, ps)
    got, gcen = ctx.extract([0], off, xy, None, ps)
    assert np.array_equal(gcen, wcen, equal_nan=True)
    scales = par.column_scales(nfx, [0], ps)
    # orientation of near-isotropic nuclei is ill-conditioned but both sides run the same f32 program
    par.assert_rows_close(got, want, scales, exact_cols=(1, 2, 3, 8), what=f"geometry ps={ps}")


def test_geometry_device_memory_entry_matches_host_entry(nfx, ctx):
    import torch
    ps = 64
    off, xy = _rings(ps, n=500)
    host, hcen = ctx.extract([0], off, xy, None, ps)
    d_off = torch.from_numpy(off.astype(np.int32)).cuda()
    d_xy = torch.from_numpy(xy).cuda()
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    dev, dcen = ctx.extract([0], d_off, d_xy, None, ps)
    torch.cuda.synchronize()
    assert np.array_equal(dev.cpu().numpy(), host, equal_nan=True)
    assert np.array_equal(dcen.cpu().numpy(), hcen, equal_nan=True)
    ctx.set_stream(0)


def test_geometry_trait_entry_f32_masks(nfx, ctx):
    """FeatureSet::compute_features_batched(centroids, polygons, patchs, masks) with the reference's
    f32 tensors (features/mod.rs:21-27)."""
    ps = 48
    off, xy = _rings(ps, n=100)
    n = len(off) - 1
    cen, ctr = ctx.preprocess_polygons(off, xy)
    masks = np.stack([orc.raster_polygon(ctr[off[i]:off[i + 1]], ps) for i in range(n)])[:, None]
    patchs = np.zeros((n, 3, ps, ps), np.float32)
    want = orc.compute_features_batched(0, off, ctr, patchs, masks[:, 0])
    fs = nfx.ShapeFeatureSet(ctx)
    assert fs.name() == "geometry"
    frame = fs.compute_features_batched(cen, [ctr[off[i]:off[i + 1]] for i in range(n)], patchs, masks)
    assert list(frame.keys()) == ["centroid"] + nfx.column_names(0)
    assert frame["centroid"][0] == nfx.centroid_key(cen[0, 0], cen[0, 1])
    got = np.stack([frame[c] for c in nfx.column_names(0)], 1)
    par.assert_rows_close(got, want, par.column_scales(nfx, [0], ps), exact_cols=(1, 2, 3, 8), what="trait geometry")


def test_unsupported_patch_size_is_an_error(nfx, ctx):
    off, xy = par.edge_case_rings(64)
    with pytest.raises(nfx.NfxError, match="UNSUPPORTED"):
        ctx.extract([0], off, xy, None, 513)   # 65..128 and 129..512 run on the wide / huge kernel families (tests/test_wide_patches_gpu.py)


def test_empty_call(nfx, ctx):
    got, cen = ctx.extract([0], np.zeros(1, np.uint32), np.zeros((0, 2), np.float32), None, 32)
    assert got.shape == (0, 12)


def test_polygon_columns_at_f64_width(nfx):
    """shape.rs:89-97 pushes geometric_features' own (f64) values into the frame: nfx_polygon_features_f64 returns the seven
    polygon columns unrounded, nfx_feature_set_column_dtype says which columns those are, and the f32 rows hold them rounded once."""
    ctx = nfx.Context(0)
    ps = 64
    off, xy, _ = par.make_inputs(12, 300, ps, with_patches=False)
    e_off, e_xy = par.edge_case_rings(ps)
    off, xy = par.concat_rings((off, xy), (e_off, e_xy))
    cen, ctr = ctx.preprocess_polygons(off, xy)
    p64 = ctx.polygon_features_f64(off, ctr)
    assert p64.dtype == np.float64 and p64.shape == (len(off) - 1, 7)
    rows, _ = ctx.extract([0], off, xy, None, ps)
    dt = nfx.column_dtypes(0)
    cols = [j for j, d in enumerate(dt) if d == "f64"]
    assert [nfx.column_names(0)[j] for j in cols] == ["area", "perimeter", "equivalent_perimeter", "compacity", "convex_hull_area",
                                                     "convex_deffect", "convex_perimeter"]
    assert all(d == "f32" for s in range(1, 5) for d in nfx.column_dtypes(s))
    with np.errstate(invalid="ignore", over="ignore"):
        assert np.array_equal(p64.astype(np.float32), rows[:, cols], equal_nan=True)
    # against the oracle at f64 width: same expression trees, -ffp-contract=off on both sides; sqrt is correctly rounded on
    # both, so only libm-free arithmetic is involved and the doubles agree to the last bits
    want64 = np.stack([orc.polygon_features_f64(ctr[off[i]:off[i + 1]]) for i in range(len(off) - 1)])
    with np.errstate(invalid="ignore", divide="ignore"):
        rel = np.abs(p64 - want64) / np.maximum(np.abs(want64), 1e-300)
    assert np.array_equal(np.isnan(p64), np.isnan(want64))
    assert np.nanmax(np.where(np.isfinite(rel), rel, 0.0)) < 1e-12
    # more digits than f32 holds: the values are not f32 numbers widened
    finite = np.isfinite(p64[:300, 1])
    assert (p64[:300, 1][finite] != p64[:300, 1][finite].astype(np.float32).astype(np.float64)).mean() > 0.9
    # device memory, same numbers
    import torch
    d64 = ctx.polygon_features_f64(torch.from_numpy(off.astype(np.int32)).cuda(), torch.from_numpy(ctr).cuda())
    ctx.synchronize()
    assert np.array_equal(d64.cpu().numpy(), p64, equal_nan=True)
    ctx.close()


@pytest.mark.parametrize("ps", [32, 64])
def test_both_polygon_kernels_give_the_same_bits(nfx, ps):
    """The seven polygon columns come from two kernels: the group-parallel one when the columns are on the critical path (geometry
    alone, the trait entry, nfx_polygon_features_f64) and the serial one on the side stream beside pixel sets.  They share one
    summation order (edge k -> partial k mod 8, butterfly of the partials), so a nucleus does not change in its last f32 bit with
    the other sets that were requested -- the command line mixes both in one file (f64 columns from the first, f32 rows from the
    second)."""
    ctx = nfx.Context(0)
    off, xy, pu8 = par.make_inputs(41, 500, ps)
    e_off, e_xy = par.edge_case_rings(ps)                  # incl. the 150-vertex ring (> PG_MAXV: the shared serial program) and degenerate rings
    off, xy = par.concat_rings((off, xy), (e_off, e_xy))
    pu8 = np.concatenate([pu8, pu8[:len(e_off) - 1]])
    alone, _ = ctx.extract([0], off, xy, None, ps)                       # k_polygon<8>
    beside, _ = ctx.extract([0, 1], off, xy, pu8, ps, batch_size=50)     # k_polygon_serial on the side stream
    assert np.array_equal(alone.view(np.uint32), beside[:, :12].copy().view(np.uint32)), \
        np.argwhere(alone.view(np.uint32) != beside[:, :12].copy().view(np.uint32))[:5]
    _, ctr = ctx.preprocess_polygons(off, xy)
    p64 = ctx.polygon_features_f64(off, ctr)
    cols = [j for j, d in enumerate(nfx.column_dtypes(0)) if d == "f64"]
    with np.errstate(invalid="ignore", over="ignore"):
        assert np.array_equal(p64.astype(np.float32), beside[:, cols], equal_nan=True)
    ctx.close()
