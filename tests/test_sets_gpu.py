"""Parity of the pixel-consuming feature sets (HIP, through the C ABI) against the CPU oracle on the
same seeded inputs: pipeline entry (u8 patches + raw rings) and trait entry (f32 patchs + f32 masks)."""
import numpy as np
import pytest

import _oracle as orc
import _parity as par

pytestmark = pytest.mark.gpu

SETS_READY = (0, 1, 2, 3, 4)   # grows as kernels land: geometry, color, glcm, glrlm, gabor


@pytest.fixture(scope="module")
def ctx(nfx):
    c = nfx.Context(0)
    yield c
    c.close()


def _inputs(ps, n, seed):
    off, xy, pu8 = par.make_inputs(seed, n, ps)
    e_off, e_xy = par.edge_case_rings(ps)
    ne = len(e_off) - 1
    off, xy = par.concat_rings((off, xy), (e_off, e_xy))
    # edge-case nuclei get patches too (recycled)
    pu8 = np.concatenate([pu8, pu8[:ne]])
    return off, xy, pu8


@pytest.mark.parametrize("set_id", [s for s in SETS_READY if s != 0])
@pytest.mark.parametrize("ps,n,batch", [(32, 130, 100), (48, 70, 32), (64, 45, 100)])
def test_pipeline_entry_matches_oracle(nfx, ctx, set_id, ps, n, batch):
    off, xy, pu8 = _inputs(ps, n, seed=20 + set_id)
    want, wcen = orc.extract([set_id], off, xy, pu8, ps, batch_size=batch)
    got, gcen = ctx.extract([set_id], off, xy, pu8, ps, batch_size=batch)
    assert np.array_equal(gcen, wcen, equal_nan=True)
    par.assert_rows_close(got, want, par.column_scales(nfx, [set_id], ps), what=f"{nfx.feature_set_cli_name(set_id)} ps={ps}")


@pytest.mark.parametrize("ps,n,batch", [(20, 37, 7), (33, 41, 100), (50, 35, 9), (63, 30, 16)])
def test_unaligned_patch_sizes_take_the_scalar_loaders(nfx, ctx, ps, n, batch):
    """Patch sizes whose rows are not whole 16-byte pieces (and whose patches do not start on 16-byte boundaries) leave
    the vector staging / prefetch paths of every kernel; ragged last batches ride along.  All five sets at once."""
    off, xy, pu8 = _inputs(ps, n, seed=70 + ps)
    sets = list(SETS_READY)
    want, wcen = orc.extract(sets, off, xy, pu8, ps, batch_size=batch)
    got, gcen = ctx.extract(sets, off, xy, pu8, ps, batch_size=batch)
    assert np.array_equal(gcen, wcen, equal_nan=True)
    par.assert_rows_close(got, want, par.column_scales(nfx, sets, ps), what=f"all sets ps={ps}")


@pytest.mark.parametrize("set_id", [s for s in SETS_READY if s != 0])
def test_trait_entry_f32_matches_oracle(nfx, ctx, set_id):
    ps, n = 48, 40
    off, xy, pu8 = _inputs(ps, n, seed=40 + set_id)
    n = len(off) - 1
    cen, ctr = ctx.preprocess_polygons(off, xy)
    masks = ctx.rasterize_polygons(off, ctr, ps)
    # arbitrary f32 patch values (not k/255): the trait receives f32 tensors
    rng = np.random.default_rng(5)
    patchs = np.clip(pu8.astype(np.float32) / np.float32(255.0) + rng.uniform(-0.002, 0.002, pu8.shape).astype(np.float32), 0, 1).astype(np.float32)
    patchs[0, :, :4, :4] = 0.0      # exercise max(rgb,1e-6) and v == 0
    want = orc.compute_features_batched(set_id, off, ctr, patchs, masks[:, 0])
    got = ctx.compute_features_batched(set_id, off, ctr, patchs, masks)
    par.assert_rows_close(got, want, par.column_scales(nfx, [set_id], ps), what=f"trait {nfx.feature_set_cli_name(set_id)}")


def test_all_ready_sets_concatenate_in_cli_order(nfx, ctx):
    """main.rs:76-89: frames are concatenated in feature-set order; a repeated set is not de-duplicated."""
    ps = 32
    off, xy, pu8 = _inputs(ps, 60, seed=9)
    sets = [s for s in [1, 0, 1, 2, 3, 4, 0] if s in SETS_READY]
    want, _ = orc.extract(sets, off, xy, pu8, ps, batch_size=25)
    got, _ = ctx.extract(sets, off, xy, pu8, ps, batch_size=25)
    par.assert_rows_close(got, want, par.column_scales(nfx, sets, ps), what="concat")


def test_batch_partition_is_part_of_the_result(nfx, ctx):
    """color.rs:50-51,144-155: mean_h couples the nuclei of a batch -> a different batch size changes it."""
    if 1 not in SETS_READY:
        pytest.skip("color not built")
    ps = 32
    off, xy, pu8 = _inputs(ps, 64, seed=10)
    a, _ = ctx.extract([1], off, xy, pu8, ps, batch_size=100)
    b, _ = ctx.extract([1], off, xy, pu8, ps, batch_size=16)
    valid = ~np.isnan(a[:, 6]) & ~np.isnan(b[:, 6])
    assert np.abs(a[valid, 6] - b[valid, 6]).max() > 1e-3
    np.testing.assert_array_equal(a[:, :6], b[:, :6])


def test_device_memory_entry_equals_host_entry(nfx, ctx):
    import torch
    ps = 48
    off, xy, pu8 = _inputs(ps, 200, seed=12)
    sets = list(SETS_READY)
    host, _ = ctx.extract(sets, off, xy, pu8, ps)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    dev, _ = ctx.extract(sets, torch.from_numpy(off.astype(np.int32)).cuda(), torch.from_numpy(xy).cuda(), torch.from_numpy(pu8).cuda(), ps)
    torch.cuda.synchronize()
    ctx.set_stream(0)
    assert np.array_equal(dev.cpu().numpy(), host, equal_nan=True)


@pytest.mark.parametrize("ps,n", [(32, 30000), (48, 9000)])
def test_glcm_list_hand_out_shapes(nfx, ctx, ps, n):
    """The three ways a GLCM launch walks its class lists (k_glcm.hip): one launch of the top class for a call of a few
    hundred nuclei (every other test), single entries per cursor visit when a list is short against the grid (9 000 nuclei),
    pairs from the class cursor when it is long (30 000 nuclei at 32x32: 25 000 masks in one class, 5 632 waves).  Counts and
    columns against the oracle; the rows must not depend on which wave took which nucleus."""
    off, xy, pu8 = par.make_inputs(300 + ps, n, ps)
    want, _ = orc.extract([2], off, xy, pu8, ps, batch_size=100, threads=16)
    got, _ = ctx.extract([2], off, xy, pu8, ps, batch_size=100)
    par.assert_rows_close(got, want, par.column_scales(nfx, [2], ps), what=f"glcm ps={ps} n={n}")
    again, _ = ctx.extract([2], off, xy, pu8, ps, batch_size=100)
    assert np.array_equal(got, again, equal_nan=True)   # dynamic hand-out, same bits


@pytest.mark.parametrize("ps,n", [(48, 2500), (64, 1500)])
def test_glcm_noise_textures_move_on_to_the_next_size_class(nfx, ctx, ps, n):
    """Every size class but the top one of a patch size keeps a hash table for the TYPICAL nucleus (3/4 slot per mask pixel).
    White noise opens a new level-128 cell with nearly every pair, so the larger nuclei of each class fill that table: the kernel
    abandons them and appends them to the next class's list (k_glcm.hip, FULL).  More nuclei than the single-launch path
    takes (768), a third of them noise, the rest tissue-like; rows against the oracle, and identical between runs although
    which wave meets which nucleus -- and the order of the handed-over nuclei -- differs."""
    off, xy, pu8 = par.make_inputs(500 + ps, n, ps)
    rng = np.random.default_rng(ps)
    noisy = np.arange(n) % 3 == 0
    pu8 = pu8.copy()
    pu8[noisy] = rng.integers(0, 256, (int(noisy.sum()), 3, ps, ps), dtype=np.uint8)
    want, _ = orc.extract([2], off, xy, pu8, ps, batch_size=100, threads=16)
    got, _ = ctx.extract([2], off, xy, pu8, ps, batch_size=100)
    par.assert_rows_close(got, want, par.column_scales(nfx, [2], ps), what=f"glcm noise ps={ps}")
    again, _ = ctx.extract([2], off, xy, pu8, ps, batch_size=100)
    assert np.array_equal(got, again, equal_nan=True)


@pytest.mark.parametrize("ps,n", [(48, 2400), (64, 1300)])
def test_glcm_wide_level_spans_go_to_the_full_range_class(nfx, ctx, ps, n):
    """Outside the top class of a patch size the dense tables and histograms cover 3/4 of the gray scale, counted from the
    nucleus's lowest level (k_glcm.hip, GlGeom).  Tissue-like nuclei span 0.6 of it; here every fourth nucleus gets a black and a
    white pixel at its centre (full span, smooth texture: the hash table does NOT fill), every fifth a span just below / above the
    limit: the kernel hands what does not fit to the top class's list before counting anything.  The integers behind the columns
    == the oracle's for every nucleus, rows against the oracle, same bits run after run."""
    off, xy, pu8 = par.make_inputs(700 + ps, n, ps)
    pu8 = pu8.copy()
    c = ps // 2
    wide = np.arange(n) % 4 == 0
    pu8[wide, :, c, c] = 0
    pu8[wide, :, c, c + 1] = 255
    edge = np.arange(n) % 5 == 1                     # spans of 0.745 .. 0.76 of the scale around the 3/4 limit
    lo = 20
    for k, i in enumerate(np.nonzero(edge)[0]):
        pu8[i] = np.clip(pu8[i], lo, 255)
        pu8[i, :, c, c] = lo
        pu8[i, :, c, c + 1] = lo + 188 + (k % 6)
    want, _ = orc.extract([2], off, xy, pu8, ps, batch_size=100, threads=16)
    got, _ = ctx.extract([2], off, xy, pu8, ps, batch_size=100)
    par.assert_rows_close(got, want, par.column_scales(nfx, [2], ps), what=f"glcm wide spans ps={ps}")
    again, _ = ctx.extract([2], off, xy, pu8, ps, batch_size=100)
    assert np.array_equal(got, again, equal_nan=True)
    m = 160
    gs, _ = ctx.debug_texture_counts(off[:m + 1], xy[:off[m]], pu8[:m], ps, glrlm=False)     # a small call: ONE launch of the full-range class
    ws, _ = orc.texture_counts(off[:m + 1], xy[:off[m]], pu8[:m], ps)
    assert np.array_equal(gs, ws)
    big, _ = ctx.debug_texture_counts(off, xy, pu8, ps, glrlm=False)                         # the size classes, hand-overs included
    assert np.array_equal(big[:m], ws)


@pytest.mark.parametrize("ps", [32, 48, 64])
def test_glcm_rows_do_not_depend_on_the_class_a_nucleus_ran_in(nfx, ctx, ps):
    """A call of up to 768 nuclei runs as one launch of the full-range class, a larger one through the typical-size classes: the
    same nucleus must come out with the same BITS either way (levels are taken relative to its lowest level in every
    instantiation, so its bins meet the same lanes and its sums the same order) -- an index-sharded run on several GPUs hands
    the last rank a short shard."""
    n = 3000
    off, xy, pu8 = par.make_inputs(800 + ps, n, ps)
    big, _ = ctx.extract([2], off, xy, pu8, ps, batch_size=100)
    for i0 in (0, 1100, 2400):
        m = 600
        o = (off[i0:i0 + m + 1] - off[i0]).astype(np.uint32)
        small, _ = ctx.extract([2], o, xy[off[i0]:off[i0 + m]], pu8[i0:i0 + m], ps, batch_size=100)
        assert np.array_equal(small.view(np.uint32), big[i0:i0 + m].view(np.uint32)), np.argwhere(small.view(np.uint32) != big[i0:i0 + m].view(np.uint32))[:5]


@pytest.mark.parametrize("batch", [1, 2, 1000, 317])
def test_colour_at_every_batch_granularity(nfx, ctx, batch):
    """The batch is the reference's par_chunks size (args.rs:107-108): any value >= 1.  mean_h couples the nuclei of a batch (color.rs:144-155), so one
    nucleus per batch, two, one batch larger than the call and a batch that is exactly the call are all different results -- each the oracle's."""
    ps, n = 48, 300
    off, xy, pu8 = par.make_inputs(41, n + (17 if batch == 317 else 0), ps)
    sets = [1, 0]
    want, _ = orc.extract(sets, off, xy, pu8, ps, batch_size=batch)
    got, _ = ctx.extract(sets, off, xy, pu8, ps, batch_size=batch)
    par.assert_rows_close(got, want, par.column_scales(nfx, sets, ps), what=f"batch={batch}")


def test_more_batches_than_one_launch_dimension_holds(nfx, ctx):
    """k_hue_batch walks the batches in slabs of 65 535 (a grid's y dimension): 70 000 batches of one nucleus each take two slabs, and the second
    slab's nuclei must get THEIR batch's sums."""
    ps, n = 32, 70_000
    off, xy, pu8 = par.make_inputs(43, n, ps)
    want, _ = orc.extract([1], off, xy, pu8, ps, batch_size=1, threads=16)
    got, _ = ctx.extract([1], off, xy, pu8, ps, batch_size=1)
    par.assert_rows_close(got, want, par.column_scales(nfx, [1], ps), what="70 000 batches")
    assert np.array_equal(got[65_530:65_545].view(np.uint32), ctx.extract([1], off[65_530:65_546] - off[65_530], xy[off[65_530]:off[65_545]],
                                                                          pu8[65_530:65_545], ps, batch_size=1)[0].view(np.uint32)), "rows across the slab boundary"


@pytest.mark.parametrize("ps", [48, 64, 33])
def test_hostile_f32_pixels_stay_inside_their_own_nucleus(nfx, ctx, ps):
    """The trait entry takes f32 tensors (features/mod.rs:21-27), so nothing guarantees values in [0, 1]: NaN, infinities, negatives, 1e30 and
    denormals in every fourth nucleus's patch.  Levels and table indices are derived from those pixels inside the kernels; whatever the poisoned
    nuclei's own rows become, every OTHER nucleus of the call must come out bit for bit as in the clean call (no index leaves its table, no
    neighbouring wave's bins are touched) -- except mean_h, which the reference itself couples across the batch (color.rs:144-155)."""
    n = 64
    off, xy, pu8 = _inputs(ps, n, seed=77)
    n = len(off) - 1
    cen, ctr = ctx.preprocess_polygons(off, xy)
    masks = ctx.rasterize_polygons(off, ctr, ps)
    clean = (pu8.astype(np.float32) / np.float32(255.0)).astype(np.float32)
    bad = clean.copy()
    rng = np.random.default_rng(11)
    poison = np.array([np.nan, np.inf, -np.inf, -5.0, 1e30, -1e30, 1e-42, 2.0, 255.0], np.float32)
    hostile = np.arange(n) % 4 == 0
    for i in np.nonzero(hostile)[0]:
        k = rng.integers(0, bad[i].size, 200)
        bad[i].reshape(-1)[k] = poison[rng.integers(0, len(poison), 200)]
    for set_id in (1, 2, 3, 4):
        a = ctx.compute_features_batched(set_id, off, ctr, clean, masks)
        b = ctx.compute_features_batched(set_id, off, ctr, bad, masks)
        keep = np.ones(a.shape[1], bool)
        if set_id == 1:
            keep[nfx.column_names(1).index("mean_h")] = False
        same = a[~hostile][:, keep].view(np.uint32) == b[~hostile][:, keep].view(np.uint32)
        assert same.all(), (nfx.feature_set_cli_name(set_id), np.argwhere(~same)[:5])
        c = ctx.compute_features_batched(set_id, off, ctr, clean, masks)          # and nothing lingers in tables or LDS for the next call
        assert np.array_equal(a.view(np.uint32), c.view(np.uint32)), nfx.feature_set_cli_name(set_id)
