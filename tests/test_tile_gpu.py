"""§8f-1 patch addressing on the GPU (load_image_dataset, utils.rs:141-206): crops gathered from a
resident [3,H,W] u8 image must equal the oracle's restatement byte for byte, incl. the zero-padding
cases at the image border, and the tile entry must equal the packed-patch entry on those crops."""
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


def _scene(synth, ps, n=120, H=300, W=340, seed=17):
    s = synth.SEED_BASE + seed
    cx, cy = synth.centres(s, 0, n, ps, W, H)
    # force border cases: corners, edges, fractional centres
    cx[:8] = [0.2, W - 0.3, 3.7, W - 4.1, ps / 2.0, W - ps / 2.0, 10.5, W / 2.0]
    cy[:8] = [0.4, H - 0.2, H - 2.5, 2.2, ps / 2.0 - 0.5, H - ps / 2.0 + 0.5, H / 2.0, 0.0]
    off, xy = synth.polygons(s, 0, n, ps, cx, cy)
    img = synth.tile(s, H, W, off, xy, ps)
    return off, xy, img


@pytest.mark.parametrize("ps", [32, 33, 47, 48, 64])
def test_gather_patches_matches_reference_addressing(synth, ctx, ps):
    off, xy, img = _scene(synth, ps)
    cen, _ = ctx.preprocess_polygons(off, xy)
    got = ctx.gather_patches(cen, img, ps)
    for i in range(len(cen)):
        want = orc.crop_patch(img, cen[i, 0], cen[i, 1], ps)
        assert np.array_equal(got[i], want), f"patch {i} at {cen[i]} differs in {(got[i] != want).sum()} bytes"
    assert (got == 0).any() and (got != 0).any()


@pytest.mark.parametrize("ps", [32, 64])
def test_slide_addressing_shifts_border_windows_instead_of_padding(synth, ctx, ps):
    """load_slide_dataset (utils.rs:97-122): the region address is `(c - ps/2) as u32`, so a nucleus closer than ps/2 to
    the top/left border gets the window at 0 (not a zero-padded one); bottom/right overhang stays zero."""
    off, xy, img = _scene(synth, ps)
    cen, _ = ctx.preprocess_polygons(off, xy)
    cen = np.concatenate([cen, np.array([[np.nan, 5.0], [-1e9, 1e12], [3e9, 5e9]], np.float32)])   # `as u32` corner cases
    image_mode = ctx.gather_patches(cen, img, ps)
    ctx.set_patch_addressing("slide")
    try:
        got = ctx.gather_patches(cen, img, ps)
    finally:
        ctx.set_patch_addressing("image")
    differs = 0
    for i in range(len(cen)):
        want = orc.crop_patch(img, cen[i, 0], cen[i, 1], ps, slide=True)
        assert np.array_equal(got[i], want), f"patch {i} at {cen[i]} differs in {(got[i] != want).sum()} bytes"
        differs += not np.array_equal(got[i], image_mode[i])
    assert np.array_equal(got[0], np.ascontiguousarray(img[:, :ps, :ps]))       # centre (0.2, 0.4): window at the origin
    assert not got[-1].any()                                                     # saturated address: nothing to read
    assert 0 < differs < len(cen)                                                # only border nuclei see the difference
    with pytest.raises(ValueError):
        ctx.set_patch_addressing("tiled")


@pytest.mark.parametrize("ps,W", [(48, 340), (32, 341), (64, 343), (16, 337), (47, 340)])
def test_tile_entry_equals_packed_entry_and_oracle(nfx, synth, ctx, ps, W):
    """The tile entry gathers every window into a packed patch and runs the packed entry's kernels on it: it must give the packed entry's
    bits -- at every row alignment (odd image widths: the gather's 16-byte loads start anywhere), with windows at the border, and on the
    sizes that take the gather's byte path (47)."""
    off, xy, img = _scene(synth, ps, n=90, W=W)
    sets = [0, 1, 2, 3, 4]
    rows_t, cen_t = ctx.extract_tile(sets, off, xy, img, ps, batch_size=32)
    patches = ctx.gather_patches(cen_t, img, ps)
    rows_p, cen_p = ctx.extract(sets, off, xy, patches, ps, batch_size=32)
    assert np.array_equal(rows_t, rows_p, equal_nan=True) and np.array_equal(cen_t, cen_p)
    want, wcen = orc.extract(sets, off, xy, np.stack([orc.crop_patch(img, c[0], c[1], ps) for c in cen_t]), ps, batch_size=32)
    assert np.array_equal(wcen, cen_t, equal_nan=True)
    par.assert_rows_close(rows_t, want, par.column_scales(nfx, sets, ps), what="tile entry")


def test_tile_entry_device_memory(nfx, synth, ctx):
    import torch
    ps = 32
    off, xy, img = _scene(synth, ps, n=64)
    host, _ = ctx.extract_tile([0, 1], off, xy, img, ps)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    dev, _ = ctx.extract_tile([0, 1], torch.from_numpy(off.astype(np.int32)).cuda(), torch.from_numpy(xy).cuda(), torch.from_numpy(img).cuda(), ps)
    torch.cuda.synchronize()
    ctx.set_stream(0)
    assert np.array_equal(dev.cpu().numpy(), host, equal_nan=True)


def test_image_addressing_of_nan_and_huge_centroids_follows_rust_casts(synth, ctx):
    """utils.rs:159-162 casts with `as i64`: NaN -> 0, +-huge saturate (a C cast would be undefined).  A NaN centroid -- the
    vertex mean of an EMPTY ring is 0/0 -- gives the empty box [0,0) x [0,0): an all-zero patch; saturated boxes lie outside."""
    ps = 32
    off, xy, img = _scene(synth, ps, n=12)
    cen = np.array([[np.nan, np.nan], [np.nan, 40.0], [1e30, 20.0], [-1e30, -1e30], [20.0, 3e38], [np.inf, -np.inf], [50.5, 60.25]], np.float32)
    got = ctx.gather_patches(cen, img, ps)
    for i in range(len(cen)):
        assert np.array_equal(got[i], orc.crop_patch(img, cen[i, 0], cen[i, 1], ps)), f"centroid {cen[i]}"
    assert not got[:6].any() and got[6].any()


def test_tile_entry_with_an_empty_ring(nfx, synth, ctx):
    """An empty ring through the tile entry: NaN centroid, empty mask, all-zero patch -> the row the reference's arithmetic
    gives (NaN where it divides by the empty mask's area), and the neighbours' rows untouched."""
    ps = 48
    off, xy, img = _scene(synth, ps, n=40)
    off2 = np.concatenate([off[:11], off[10:]]).astype(np.uint32)     # nucleus 10 becomes an empty ring; 41 nuclei
    sets = [0, 1, 2, 3, 4]
    rows, cen = ctx.extract_tile(sets, off2, xy, img, ps, batch_size=16)
    assert np.isnan(cen[10]).all()
    patches = np.stack([orc.crop_patch(img, c[0], c[1], ps) for c in cen])
    assert not patches[10].any()
    want, wcen = orc.extract(sets, off2, xy, patches, ps, batch_size=16)
    assert np.array_equal(wcen, cen, equal_nan=True)
    par.assert_rows_close(rows, want, par.column_scales(nfx, sets, ps), what="tile entry, empty ring")
    assert np.isnan(rows[10, 1]) and np.isnan(rows[10, 12])           # major_axis, mean_r: 0/0


@pytest.mark.parametrize("ps,H,W", [(32, 20, 45), (48, 1, 1), (64, 70, 9), (17, 16, 16), (64, 64, 64)])
def test_images_smaller_than_a_patch(nfx, synth, ctx, ps, H, W):
    """load_image_dataset clips the box to the image and pastes what is left at (-min(top,0), -min(left,0)) of a zero patch (utils.rs:164-191): an image
    narrower or lower than the patch -- down to one pixel -- is clipped on BOTH sides of an axis at once.  Byte-exact through the gather, the tile entry
    and the streamed entry (one band lower than its own halo), in both addressing modes."""
    rng = np.random.default_rng(ps * 1000 + H)
    img = rng.integers(1, 256, (3, H, W), dtype=np.uint8)
    n = 40
    cen = np.stack([rng.uniform(-ps, W + ps, n), rng.uniform(-ps, H + ps, n)], 1).astype(np.float32)
    cen[:4] = [[0.0, 0.0], [W / 2.0, H / 2.0], [W - 0.25, H - 0.25], [W / 2.0 + 0.5, -3.0]]
    for mode in ("image", "slide"):
        ctx.set_patch_addressing(mode)
        got = ctx.gather_patches(cen, img, ps)
        for i in range(n):
            want = orc.crop_patch(img, cen[i, 0], cen[i, 1], ps, slide=(mode == "slide"))
            assert np.array_equal(got[i], want), (mode, i, cen[i])
    ctx.set_patch_addressing("image")
    # the same windows under rings: tile entry == packed entry on the oracle's crops, streamed entry == tile entry
    off = np.arange(0, 5 * n + 1, 5, dtype=np.uint32)
    d = np.array([[-3, -2], [4, -3], [5, 4], [-4, 3], [-3, -2]], np.float32) * (ps / 16.0)
    xy = (cen[:, None, :] + d[None] - d[:4].mean(0) * 0).reshape(-1, 2).astype(np.float32)
    sets = [0, 1, 2, 3, 4]
    rows_t, cen_t = ctx.extract_tile(sets, off, xy, img, ps, batch_size=16)
    patches = np.stack([orc.crop_patch(img, c[0], c[1], ps) for c in cen_t])
    rows_p, _ = ctx.extract(sets, off, xy, patches, ps, batch_size=16)
    assert np.array_equal(rows_t.view(np.uint32), rows_p.view(np.uint32))
    rows_s, cen_s, st = ctx.extract_tile_streamed(sets, off, xy, img, ps, batch_size=16, block=64, chunk_nuclei=16)
    assert np.array_equal(cen_s, cen_t, equal_nan=True) and np.array_equal(rows_s.view(np.uint32), rows_t.view(np.uint32))
    assert st["blocks_total"] == -(-H // 64)
