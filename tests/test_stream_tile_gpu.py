"""BASELINE configs[4]: the slide streamed band by band (nfx_extract_tile_streamed_v2) must give the rows of the resident-image call
bit for bit -- same windows (utils.rs:159-191 / 97-122), same reference batches (the hue mean couples them, color.rs:144-155) --
whatever the order of the nuclei, the band height, the chunk size and the number of bands the device cache holds."""
import ctypes as C

import numpy as np
import pytest

import _parity as par

pytestmark = pytest.mark.gpu


def _scene(synth, ps, n, H, W, seed, slide_order):
    s = synth.SEED_BASE + seed
    cx, cy = synth.centres(s, 0, n, ps, W, H)
    k = min(8, n)
    cx[:k] = [0.2, W - 0.3, 3.7, W - 4.1, ps / 2.0, W - ps / 2.0, 10.5, W / 2.0][:k]     # corners, edges: zero padding / shifted windows
    cy[:k] = [0.4, H - 0.2, H - 2.5, 2.2, ps / 2.0 - 0.5, H - ps / 2.0 + 0.5, H / 2.0, 0.0][:k]
    if slide_order:   # what a segmentation tool emits: tile by tile (128-pixel tiles, raster order)
        order = np.lexsort((cx, (cx // 128).astype(np.int64), (cy // 128).astype(np.int64)))
        cx, cy = cx[order], cy[order]
    off, xy = synth.polygons(s, 0, n, ps, cx, cy)
    img = synth.tile(s, H, W, off, xy, ps)
    return off, xy, img


@pytest.mark.parametrize("ps,slide_order,block,chunk,cache_slots,pinned,B", [
    (48, True, 128, 64, 0, True, 16),     # slide order, default cache: every needed band travels once
    (48, False, 64, 8, 3, False, 4),      # random order, four band slots of nine, pageable image: evictions and re-copies
    (32, False, 256, 1000, 0, True, 16),  # one chunk
    (33, True, 64, 96, 40, False, 16),    # odd patch size (byte path of the gather)
    (64, False, 96, 48, 2, True, 3),      # batches of three nuclei in random order over two band slots: some batches are gathered in two passes
    (48, False, 64, 24, 1, True, 8),      # ONE band slot (cache_bytes is honoured down to one band): every batch in as many passes as it has bands
    (136, True, 192, 40, 2, True, 10),    # a patch size of the huge kernel family (129..512): three-word mask rows, halo of 136 rows per band
])
def test_streamed_rows_equal_resident_rows(nfx, synth, ps, slide_order, block, chunk, cache_slots, pinned, B):
    n, H, W = (310, 520, 700) if ps <= 64 else (40, 520, 700)
    off, xy, img = _scene(synth, ps, n, H, W, 23, slide_order)
    e_off, e_xy = par.edge_case_rings(ps)                     # an empty ring among them: NaN centroid, window at the origin
    off, xy = par.concat_rings((off, xy), (e_off, e_xy))
    sets = [0, 1, 2, 3, 4]
    ctx = nfx.Context(0)
    want, wcen = ctx.extract_tile(sets, off, xy, img, ps, batch_size=B)
    image = img
    if pinned:
        image = nfx.pinned_empty(img.shape, np.uint8)
        image[...] = img
    slot_bytes = (3 * min(H, block + ps) * W + 255) & ~255      # a band: `block` rows + a patch-size halo, full width
    got, cen, st = ctx.extract_tile_streamed(sets, off, xy, image, ps, batch_size=B, block=block, chunk_nuclei=chunk,
                                             cache_bytes=cache_slots * slot_bytes)
    assert np.array_equal(cen, wcen, equal_nan=True)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), np.argwhere(got.view(np.uint32) != want.view(np.uint32))[:5]
    assert st["image_pinned"] == int(pinned) and st["chunks"] >= (len(off) - 1 + max(B, chunk // B * B) - 1) // max(B, chunk // B * B)
    assert st["blocks_total"] == -(-H // block)
    if cache_slots == 0:
        assert st["blocks_copied"] <= st["blocks_total"], "with room for the whole slide no band travels twice"
    else:
        assert st["block_slots"] == min(st["blocks_total"], cache_slots), "cache_bytes is honoured (ADVICE r3: no hidden floor)"
        if cache_slots < min(B, 3) and not slide_order:
            assert st["split_gather_passes"] > 0, "a batch wider than the cache must have been gathered in several passes"
            assert st["blocks_copied"] > st["blocks_total"]
    assert st["image_bytes_h2d"] >= st["blocks_copied"] * 3 * 64 * 64
    assert st["rows_direct"] == 0, "a pageable row matrix is staged on the device and copied home"
    # ROWS STRAIGHT HOME (the default whenever out_rows is pinned: the command line and the bench's top line): the feature kernels write
    # into the caller's pinned matrix over the link -- evictions, multi-pass gathers, odd and huge patch sizes included; k_gabor's REFINED
    # launch reads its shifts back from those rows
    rows_p = nfx.pinned_empty(want.shape, np.float32)
    rows_p[...] = np.float32(-7.0)
    got_p, cen_p, st_p = ctx.extract_tile_streamed(sets, off, xy, image, ps, batch_size=B, block=block, chunk_nuclei=chunk,
                                                   cache_bytes=cache_slots * slot_bytes, out=rows_p)
    assert st_p["rows_direct"] == 1 and got_p is rows_p
    assert np.array_equal(cen_p, wcen, equal_nan=True)
    assert np.array_equal(rows_p.view(np.uint32), want.view(np.uint32)), np.argwhere(rows_p.view(np.uint32) != want.view(np.uint32))[:5]
    # ... and the staged form on the same pinned matrix when the option asks for it (nfx_ctx_set_option, NFX_OPT_STREAM_ROWS_STAGED)
    ctx.set_option("stream_rows_staged", 1)
    rows_p[...] = np.float32(-7.0)
    _, _, st_c = ctx.extract_tile_streamed(sets, off, xy, image, ps, batch_size=B, block=block, chunk_nuclei=chunk,
                                           cache_bytes=cache_slots * slot_bytes, out=rows_p)
    ctx.set_option("stream_rows_staged", 0)
    assert st_c["rows_direct"] == 0
    assert np.array_equal(rows_p.view(np.uint32), want.view(np.uint32))
    # the slide loader's addressing (utils.rs:97-122), rows straight home again
    ctx.set_patch_addressing("slide")
    want_s, _ = ctx.extract_tile(sets, off, xy, img, ps, batch_size=B)
    got_s, _, st_s = ctx.extract_tile_streamed(sets, off, xy, image, ps, batch_size=B, block=block, chunk_nuclei=chunk, cache_bytes=cache_slots * slot_bytes,
                                               out=rows_p)
    ctx.set_patch_addressing("image")
    assert st_s["rows_direct"] == 1
    assert np.array_equal(got_s.view(np.uint32), want_s.view(np.uint32))
    assert not np.array_equal(got_s.view(np.uint32), got.view(np.uint32)), "border nuclei see the two loaders differently"
    ctx.close()


def test_slide_order_needs_few_copies_random_order_many(nfx, synth):
    """The same nuclei in slide order and shuffled, through a cache of four bands out of twelve: rows agree nucleus for nucleus
    with the resident call either way; only the number of band copies differs."""
    ps, n, H, W, B = 32, 600, 768, 768, 4
    off, xy, img = _scene(synth, ps, n, H, W, 29, True)
    sets = [1, 2]
    ctx = nfx.Context(0)
    slot = (3 * (64 + ps) * W + 255) & ~255
    rows, _, st = ctx.extract_tile_streamed(sets, off, xy, img, ps, batch_size=B, block=64, chunk_nuclei=40, cache_bytes=4 * slot)
    want, _ = ctx.extract_tile(sets, off, xy, img, ps, batch_size=B)
    assert np.array_equal(rows.view(np.uint32), want.view(np.uint32))
    rng = np.random.default_rng(3)
    perm = rng.permutation(n)
    rings = [xy[off[i]:off[i + 1]] for i in perm]
    off2 = np.zeros(n + 1, np.uint32)
    np.cumsum([len(r) for r in rings], out=off2[1:])
    xy2 = np.concatenate(rings)
    rows2, _, st2 = ctx.extract_tile_streamed(sets, off2, xy2, img, ps, batch_size=B, block=64, chunk_nuclei=40, cache_bytes=4 * slot)
    want2, _ = ctx.extract_tile(sets, off2, xy2, img, ps, batch_size=B)
    assert np.array_equal(rows2.view(np.uint32), want2.view(np.uint32))
    assert st["block_slots"] == 4 and st["blocks_total"] == 12
    assert st["blocks_copied"] <= 2 * st["blocks_total"] and st2["blocks_copied"] > 4 * st["blocks_copied"], (st, st2)
    ctx.close()


def test_geometry_alone_and_empty_calls_need_no_image(nfx, synth):
    ps = 32
    off, xy, _ = par.make_inputs(3, 50, ps)
    ctx = nfx.Context(0)
    rows, cen, st = ctx.extract_tile_streamed([0], off, xy, np.zeros((3, 8, 8), np.uint8), ps)
    want, wcen = ctx.extract([0], off, xy, None, ps)
    assert np.array_equal(rows, want, equal_nan=True) and np.array_equal(cen, wcen) and st["blocks_copied"] == 0
    rows, cen, _ = ctx.extract_tile_streamed([0, 1], np.zeros(1, np.uint32), np.zeros((0, 2), np.float32), np.zeros((3, 8, 8), np.uint8), ps)
    assert rows.shape == (0, 30)
    # errors: reported, nothing left in flight
    L = nfx.lib()
    s = (C.c_int * 1)(1)
    o = off.astype(np.uint32)
    st_ = L.nfx_extract_tile_streamed_v2(ctx._h, s, 1, 50, int(o[-1]), o.ctypes.data_as(C.c_void_p), xy.ctypes.data_as(C.c_void_p), None, 8, 8, ps, 100, 0, 0, 0,
                                         rows.ctypes.data_as(C.c_void_p), None, None)
    assert st_ == 1 and b"image" in L.nfx_last_error(ctx._h)
    # a struct_size that cannot be a caller's sizeof (an uninitialised first field) is refused before anything is written
    for bad in (0, 8, 20, 8192):
        stats = (C.c_int64 * 16)(bad, *([-1] * 15))
        st_ = L.nfx_extract_tile_streamed_v2(ctx._h, s, 1, 50, int(o[-1]), o.ctypes.data_as(C.c_void_p), xy.ctypes.data_as(C.c_void_p), None, 8, 8, ps, 100, 0, 0, 0,
                                             rows.ctypes.data_as(C.c_void_p), None, stats)
        assert st_ == 1 and b"struct_size" in L.nfx_last_error(ctx._h) and list(stats)[1:] == [-1] * 15
    # the entry of ABI <= 0.4.0 under its old name: an error, never a write into an old caller's shorter struct
    old = L.nfx_extract_tile_streamed
    old.argtypes = L.nfx_extract_tile_streamed_v2.argtypes
    stats = (C.c_int64 * 16)(*([-1] * 16))
    assert old(ctx._h, s, 1, 50, int(o[-1]), o.ctypes.data_as(C.c_void_p), xy.ctypes.data_as(C.c_void_p), None, 8, 8, ps, 100, 0, 0, 0,
               rows.ctypes.data_as(C.c_void_p), None, stats) == 2
    assert b"0.5.0" in L.nfx_last_error(ctx._h) and list(stats) == [-1] * 16
    # options: unknown option / value
    assert L.nfx_ctx_set_option(ctx._h, 99, 1) == 1 and L.nfx_ctx_set_option(ctx._h, 0, 2) == 1 and L.nfx_ctx_set_option(ctx._h, 0, 0) == 0
    ctx.close()
