// nfx_stream.cpp -- nfx_extract_tile_streamed_v2: the slide streamed band by band, host -> device copies overlapped with compute.
//
// Replaces (reference file:line): the per-batch pipeline stage of main.rs:146-151 on a slide that does not have to be resident --
// load_image_dataset's crop of every nucleus's window out of the image (utils.rs:141-206), move_tensors_to_device (utils.rs:211-224:
// a synchronous copy of two f32 tensors, 16 ps^2 bytes per nucleus) and the feature sets.  BASELINE.json configs[4]: "WSI-scale 10M
// nuclei, 64 x 64 patches, full feature set, streamed tile H2D overlapped with compute".
//
// What travels over the host link is the TILE, cut into horizontal BANDS of 2048 rows by default (full width, each with a
// patch-size halo below, so that every nucleus's window lies inside the band its first row falls in): 3 bytes per slide pixel,
// i.e. (ps/1.5)^2 x 3 bytes per nucleus at the density of the synthetic slides -- 2.25x fewer than packed patches (3 ps^2), 12x
// fewer than the reference's f32 patch + mask tensors.  Bands, not squares: the rows of a band are contiguous in the planar
// [3][H][W] image, so a band is three large contiguous copies at the link's full rate (square 2048^2 blocks are 2112-byte rows at
// // the image's pitch: measured 35 GB/s with an aligned pitch and 0.3 GB/s with an odd one through hipMemcpy2DAsync, against 45-50 GB/s).
// The reference's batch partition is kept (the hue mean couples the nuclei of a batch, color.rs:144-155): nuclei are processed
// in index order in CHUNKS of whole batches; a chunk runs when every band it touches is resident.  Segmentation tools emit
// nuclei tile by tile, so a chunk touches one or two bands; any order is correct (a band that is needed again after it was
// evicted is copied again; the device keeps as many bands as `cache_bytes` allows -- the whole slide, typically -- and never fewer
// than one: a batch that touches more bands than that is gathered in several passes over its bands).
//
//   copy stream    : the rings of the centroid frontier, then the H2D bands of chunks k+1, k+2 (`lead` chunks ahead of the extraction when
//                    the whole slide keeps a cache slot, one chunk ahead otherwise), back to back: the link never waits for a kernel
//   rows stream    : k_preprocess of the frontier's rings; the lists of chunk k+1 and the gather of its windows out of its bands into
//                    patches[(k+1)%3] (waits on the DEVICE for extract(k-2), the buffer's last reader)
//   compute streams: extract(chunk k) from patches[k%3], chunks alternating between TWO streams: the first kernels of chunk k+1 fill the
//                    CUs that the last workgroups of chunk k's kernels leave idle.  The kernels write their rows straight into the
//                    caller's out_rows when it is pinned (posted writes over the link); otherwise into rows[k%3]
//   d2h stream     : the frontier's centroids home; rows[k%3] of chunk k home when out_rows is pageable (or the staged form was asked for)
//
// Results are bit-identical to nfx_extract_tile on the resident image: the same gather kernel fills the same packed patches, the
// same extraction runs on the same batches.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "nfx_internal.h"

using namespace nfx;

namespace {

// (c - ps/2) as the two loaders turn it into the window's first row / column (k_util.hip: same f32 arithmetic, same casts)
int64_t rust_f32_as_i64_host(float v)
{
    if (v != v) return 0;
    const float lim = 1099511627776.0f; // 2^40
    return v >= lim ? (int64_t)1 << 40 : (v <= -lim ? -((int64_t)1 << 40) : (int64_t)v);
}
int64_t window_origin(float c, float half, bool slide)
{
    const float v = c - half;
    if (slide) return v >= 4294967296.0f ? 4294967295ll : (v > 0.0f ? (int64_t)v : 0ll); // `as u32`, utils.rs:104-105
    const int64_t t = rust_f32_as_i64_host(v);                                            // `as i64`, utils.rs:159-162
    return t > 0 ? t : 0;
}

bool is_pinned(const void *p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

struct Pinned { // grow-only pinned staging kept by the context: region 0 = centroids of the call, region 1 = per-chunk lists
    static void *get(nfx_ctx *ctx, int region, size_t bytes)
    {
        void *&p = region == 0 ? ctx->pin_cen : ctx->pin;
        size_t &cap = region == 0 ? ctx->pin_cen_cap : ctx->pin_cap;
        if (bytes <= cap) return p;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); p = nullptr; return nullptr; }
        cap = bytes;
        return p;
    }
};

struct Chunk {
    int64_t i0, i1;                  // nuclei
    std::vector<int64_t> blocks;     // distinct bands its windows start in, ascending
    std::vector<uint32_t> sel;       // chunk-relative nucleus indices grouped by block (same order as `blocks`)
    std::vector<uint32_t> sel_off;   // blocks.size() + 1 offsets into sel
};

} // namespace

extern "C" {

void *nfx_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
void nfx_host_free(void *p) { if (p) (void)hipHostFree(p); }
nfx_status nfx_host_register(void *p, size_t bytes)
{
    if (!p) return NFX_ERR_INVALID_ARGUMENT;
    const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
    if (e == hipSuccess || e == hipErrorHostMemoryAlreadyRegistered) { (void)hipGetLastError(); return NFX_OK; }
    (void)hipGetLastError();
    return NFX_ERR_DEVICE;
}
void nfx_host_unregister(void *p) { if (p && hipHostUnregister(p) != hipSuccess) (void)hipGetLastError(); }

// The entry of ABI 0.3.0 / 0.4.0 under its old name: those callers' nfx_stream_stats has no struct_size (0.3.0) or may be shorter than this
// library's -- they get an error, not memory written past their struct (include/nfx.h, NFX_VERSION 500)
nfx_status nfx_extract_tile_streamed(nfx_ctx *ctx, const int *, int, int64_t, int64_t, const uint32_t *, const float *, const uint8_t *, int64_t, int64_t,
                                     int, int, int, int64_t, int64_t, float *, float *, void *)
{
    const nfx_status st = NFX_ERR_UNSUPPORTED;
    return ctx ? api_fail(ctx, st, "nfx_extract_tile_streamed is the entry of ABI <= 0.4.0: rebuild against include/nfx.h 0.5.0 (nfx_extract_tile_streamed_v2)") : st;
}

nfx_status nfx_extract_tile_streamed_v2(nfx_ctx *ctx, const int *sets, int n_sets, int64_t n, int64_t nv, const uint32_t *ring_offsets,
                                        const float *ring_xy, const uint8_t *image_u8, int64_t H, int64_t W, int ps, int batch_size, int block,
                                        int64_t chunk_nuclei, int64_t cache_bytes, float *out_rows, float *out_centroids_xy, nfx_stream_stats *stats)
{
    nfx_status st = check_common(ctx, n, nv, ps);
    if (st != NFX_OK) return st;
    // the caller says how large ITS nfx_stream_stats is: nothing is written past that (include/nfx.h)
    const int64_t stats_size = stats ? stats->struct_size : 0;
    if (stats && (stats_size < 16 || stats_size > 4096 || (stats_size & 7))) // (an uninitialised first field is caught here, not as a write past the caller's struct)
        return api_fail(ctx, NFX_ERR_INVALID_ARGUMENT, "stats->struct_size must be sizeof(nfx_stream_stats) of the caller's header");
    auto put_stats = [&](const nfx_stream_stats &full) {
        if (!stats) return;
        memcpy((char *)stats + 8, (const char *)&full + 8, (size_t)std::min<int64_t>(stats_size, (int64_t)sizeof(full)) - 8);
    };
    put_stats(nfx_stream_stats{});
    if (!sets || n_sets <= 0) return api_fail(ctx, NFX_ERR_INVALID_ARGUMENT, "no feature set selected");
    const int F = sets_total_columns(sets, n_sets);
    if (F < 0) return api_fail(ctx, NFX_ERR_INVALID_ARGUMENT, "unknown feature set");
    if (batch_size < 1) return api_fail(ctx, NFX_ERR_INVALID_ARGUMENT, "batch_size must be >= 1");
    if (!ring_offsets || (nv && !ring_xy) || !out_rows) return api_fail(ctx, NFX_ERR_INVALID_ARGUMENT, "null input");
    const bool pix = any_pixel_set(sets, n_sets);
    if (n == 0) return NFX_OK;
    if (!pix) // nothing to stream: geometry needs no pixels
        return nfx_extract(ctx, sets, n_sets, n, nv, ring_offsets, ring_xy, nullptr, ps, batch_size, out_rows, out_centroids_xy, NFX_MEM_HOST);
    if (!image_u8 || H < 1 || W < 1) return api_fail(ctx, NFX_ERR_INVALID_ARGUMENT, "null or empty image");
    NFX_HIP(ctx, hipSetDevice(ctx->device));
    StreamScope scope(ctx, NFX_MEM_HOST); // ctx->stream = the context's own stream (the compute stream below), host arena
    if ((st = check_offsets(ctx, ring_offsets, n, nv, "ring_offsets")) != NFX_OK) return st;
    if (!ctx->copy_stream) { // the gathers (and whatever kernels the runtime copies with) go ahead of the extraction's when CUs free up
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); hi = 0; }
        NFX_HIP(ctx, hipStreamCreateWithPriority(&ctx->copy_stream, hipStreamNonBlocking, hi));
    }
    if (!ctx->rows_stream) {
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); hi = 0; }
        NFX_HIP(ctx, hipStreamCreateWithPriority(&ctx->rows_stream, hipStreamNonBlocking, hi));
    }
    // the rows leave on a stream of their own, so that they wait for nothing but their chunk's extraction.  (The runtime carries THIS
    // direction with a copy kernel of its own -- one 512-thread workgroup per CU -- while the bands arrive through the DMA engine: measured,
    // profiles/r03_stream_d2h.txt; forcing the rows onto the engine as well queues them behind the bands and stalls the extraction.)
    if (!ctx->d2h_stream) {
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); hi = 0; }
        NFX_HIP(ctx, hipStreamCreateWithPriority(&ctx->d2h_stream, hipStreamNonBlocking, hi));
    }
    for (int b = 0; b < 3; ++b)
        for (hipEvent_t *e : {&ctx->ev_ready[b], &ctx->ev_done[b], &ctx->ev_rows[b]})
            if (!*e) NFX_HIP(ctx, hipEventCreateWithFlags(e, hipEventDisableTiming));
    for (int b = 0; b < 4; ++b)
        if (!ctx->ev_bands[b]) NFX_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_bands[b], hipEventDisableTiming));
    if (!ctx->ev_pass) NFX_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_pass, hipEventDisableTiming));
    if (!ctx->alt_stream) NFX_HIP(ctx, hipStreamCreateWithFlags(&ctx->alt_stream, hipStreamNonBlocking));
    hipStream_t const compute = ctx->stream, copy = ctx->copy_stream, rows_s = ctx->rows_stream, d2h = ctx->d2h_stream;
    // chunks alternate between two compute streams
#ifdef NFX_EXPERIMENTS // NFX_STREAM_NCS=1: one stream, chunk after chunk
    const bool one_cs = [] { const char *e = getenv("NFX_STREAM_NCS"); return e && atoi(e) == 1; }();
#else
    constexpr bool one_cs = false;
#endif
    hipStream_t const cs[2] = {compute, one_cs ? compute : ctx->alt_stream};
    constexpr int NB = 3; // buffer sets (patches, rows, selection lists): one being filled, two being extracted
    const bool slide = ctx->addressing == NFX_ADDR_SLIDE;
    // Rows straight home.  When out_rows is pinned for this runtime the feature kernels write their columns into it themselves (the
    // mapped host pointer goes in as the rows' address): posted writes over the link, spread over the whole extraction, instead of a
    // device buffer per chunk and a copy behind it -- which the runtime carries with a copy KERNEL of one workgroup per CU while the
    // bands hold the DMA engine (profiles/r03_stream_d2h.txt).  Same kernels, same rows.  nfx_ctx_set_option(NFX_OPT_STREAM_ROWS_STAGED, 1)
    // keeps the staged form.
    float *rows_direct = nullptr;
#ifdef NFX_EXPERIMENTS // MEASUREMENT ONLY (out_rows is never written): the rows stay in a device matrix -- what the kernels' writes over the link cost
    const bool exp_rows_dev = [] { const char *e = getenv("NFX_EXP_STREAM_ROWS_DEVICE"); return e && atoi(e) == 1; }();
#else
    constexpr bool exp_rows_dev = false;
#endif
    {
        const size_t rows_bytes = 4 * (size_t)n * F;
        if (!ctx->opt_stream_rows_staged && rows_bytes && is_pinned(out_rows) && is_pinned((const char *)out_rows + rows_bytes - 1)) {
            void *dp = nullptr;
            if (hipHostGetDevicePointer(&dp, out_rows, 0) == hipSuccess && dp) rows_direct = (float *)dp;
            else (void)hipGetLastError();
        }
    }

    // ---- geometry of the plan: "block" b is the band of rows [b B, b B + B + ps) (clipped to the image), all W columns -----------
    const int64_t B = block > 0 ? (block < 64 ? 64 : block) : 2048;
    const int64_t nblocks = (H + B - 1) / B;
    const int64_t sh = std::min<int64_t>(H, B + ps); // rows of a band incl. its halo
    const size_t slot_bytes = al(3 * (size_t)sh * W);
    const int64_t chunk_max = std::max<int64_t>(batch_size, ((chunk_nuclei > 0 ? chunk_nuclei : 100000) / batch_size) * batch_size);
    const int64_t chunk_cap = std::min<int64_t>(chunk_max, ((n + batch_size - 1) / batch_size) * batch_size);
    int64_t want_slots;
    if (cache_bytes > 0) want_slots = cache_bytes / (int64_t)slot_bytes;
    else { // default: the whole slide if it fits beside everything else (60 % of what is free), else what fits.  The host arena this call
           // is about to reuse (or replace) counts as free: a second call must not plan with less than the first
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = (size_t)16 << 30; }
        free_b += ctx->host_ws.cap;
        want_slots = (int64_t)((free_b / 10 * 6) / slot_bytes);
    }
    // `cache_bytes` is honoured down to ONE band (a batch whose windows start in more bands than the cache holds is gathered in several
    // passes over its bands: the gather needs one band at a time, only the extraction needs all of a batch's patches)
    const int64_t n_slots = std::min<int64_t>(nblocks, std::max<int64_t>(want_slots, 1));

    // ---- stage the rings, centroids of every nucleus (k_preprocess: the arithmetic that addresses the windows) ------------------
    const size_t P3 = 3 * (size_t)ps * ps;
    size_t max_chunk_nv = 0; // vertices of the largest possible chunk: any run of chunk_cap nuclei
    for (int64_t i0 = 0; i0 < n; i0 += batch_size) {
        const int64_t i1 = std::min<int64_t>(n, i0 + chunk_cap);
        max_chunk_nv = std::max<size_t>(max_chunk_nv, (size_t)(ring_offsets[i1] - ring_offsets[i0]));
    }
    const size_t scratch = extract_ws_bytes(sets, n_sets, chunk_cap, (int64_t)max_chunk_nv, ps, batch_size, false, false);
    size_t need = al(4 * ((size_t)n + 1)) + al(8 * (size_t)nv + 8) + al(8 * (size_t)n + 8) + (size_t)n_slots * slot_bytes +
                  NB * al((size_t)chunk_cap * P3) + (rows_direct ? 0 : NB * al(4 * (size_t)chunk_cap * F + 8)) + NB * al(4 * (size_t)chunk_cap + 8) +
                  NB * al(4 * ((size_t)chunk_cap + 1)) + 2 * al(scratch) + 8192 + (exp_rows_dev ? al(4 * (size_t)n * F + 8) : 0);
    if ((st = ws_reserve(ctx, need, NFX_MEM_HOST)) != NFX_OK) return st;
    uint32_t *d_off_all = (uint32_t *)ws_alloc(ctx, 4 * ((size_t)n + 1));
    float *d_xy = (float *)ws_alloc(ctx, 8 * (size_t)nv + 8);
    float *d_cen = (float *)ws_alloc(ctx, 8 * (size_t)n + 8);
    uint8_t *d_slots = (uint8_t *)ws_alloc(ctx, (size_t)n_slots * slot_bytes);
    uint8_t *d_patches[NB]; float *d_rows[NB]; uint32_t *d_sel[NB], *d_off[NB];
    for (int b = 0; b < NB; ++b) {
        d_patches[b] = (uint8_t *)ws_alloc(ctx, (size_t)chunk_cap * P3);
        d_rows[b] = rows_direct ? rows_direct : (float *)ws_alloc(ctx, 4 * (size_t)chunk_cap * F + 8);
        d_sel[b] = (uint32_t *)ws_alloc(ctx, 4 * (size_t)chunk_cap + 8);
        d_off[b] = (uint32_t *)ws_alloc(ctx, 4 * ((size_t)chunk_cap + 1));
    }
    if (exp_rows_dev) { // (a whole [n, F] device matrix takes the host matrix's place: what the kernels' writes over the link cost)
        rows_direct = (float *)ws_alloc(ctx, 4 * (size_t)n * F + 8);
        if (!rows_direct) return api_fail(ctx, NFX_ERR_OUT_OF_MEMORY, "workspace");
        for (int b = 0; b < NB; ++b) d_rows[b] = rows_direct;
    }
    if (!d_off_all || !d_xy || !d_cen || !d_slots || !d_patches[NB - 1] || !d_rows[NB - 1] || !d_sel[NB - 1] || !d_off[NB - 1])
        return api_fail(ctx, NFX_ERR_OUT_OF_MEMORY, "workspace");
    // a chunk's extraction bump-allocates its scratch from one of two marks: two extractions are in flight, on two streams
    const size_t ws_mark[2] = {ctx->ws->off, ctx->ws->off + al(scratch)};
    if (ctx->ws->cap - ws_mark[0] < 2 * al(scratch)) return api_fail(ctx, NFX_ERR_OUT_OF_MEMORY, "workspace");
    if ((st = prepare_sets(ctx, sets, n_sets)) != NFX_OK) return st;
    if (!ctx->ev_front) NFX_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_front, hipEventDisableTiming));
    if (!ctx->ev_front_up) NFX_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_front_up, hipEventDisableTiming));
    if (!ctx->ev_front_cen) NFX_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_front_cen, hipEventDisableTiming));
    // The centroids address the windows (k_preprocess: the loader's own arithmetic), so the host needs them before it can plan a chunk.
    // They are made a FRONTIER at a time, a chunk or two ahead of the plan, not all up front: the rings of nuclei [front, upto) travel on the
    // copy stream (ahead of the bands that will be planned from them), are centred on the rows stream and their centroids come home on
    // the d2h stream -- so the first band leaves a millisecond into the call instead of behind 310 MB of vertices (1.25 M nuclei), and
    // the rest of the rings travel in the link's spare time (it outruns the extraction).
    float *const cen = (float *)Pinned::get(ctx, 0, 8 * (size_t)n + 64); // pinned: the copy home is asynchronous
    if (!cen) return api_fail(ctx, NFX_ERR_OUT_OF_MEMORY, "pinned staging (centroids)");
    NFX_HIP(ctx, hipMemcpyAsync(d_off_all, ring_offsets, 4 * ((size_t)n + 1), hipMemcpyHostToDevice, copy));
    int64_t front = 0, other_h2d = 4 * (n + 1) + 8 * nv;
    auto advance_front = [&](int64_t upto) -> hipError_t { // enqueue only; ev_front fires when the centroids of [0, upto) are on the host
        upto = std::min<int64_t>(upto, n);
        if (upto <= front) return hipSuccess;
        const size_t v0 = ring_offsets[front], v1 = ring_offsets[upto];
        hipError_t e = hipSuccess;
        if (v1 > v0) e = hipMemcpyAsync(d_xy + 2 * v0, ring_xy + 2 * v0, 8 * (v1 - v0), hipMemcpyHostToDevice, copy);
        if (e == hipSuccess) e = hipEventRecord(ctx->ev_front_up, copy);
        if (e == hipSuccess) e = hipStreamWaitEvent(rows_s, ctx->ev_front_up, 0);
        if (e != hipSuccess) return e;
        ctx->stream = rows_s;
        e = launch_preprocess(ctx, upto - front, d_off_all + front, d_xy, d_cen + 2 * front, nullptr); // (offsets are absolute: d_xy is the base)
        ctx->stream = compute;
        if (e == hipSuccess) e = hipEventRecord(ctx->ev_front_cen, rows_s);
        if (e == hipSuccess) e = hipStreamWaitEvent(d2h, ctx->ev_front_cen, 0);
        if (e == hipSuccess) e = hipMemcpyAsync(cen + 2 * front, d_cen + 2 * front, 8 * (size_t)(upto - front), hipMemcpyDeviceToHost, d2h);
        if (e == hipSuccess) e = hipEventRecord(ctx->ev_front, d2h);
        front = upto;
        return e;
    };

    // ---- the plan, made chunk by chunk while the device works on the chunks before: the band of every nucleus's window, chunks of
    // whole batches whose bands fit the cache, the chunk's nuclei grouped by band (a counting sort: bands are few) -----------------
    const float half = (float)ps / 2.0f;
    auto band_of = [&](int64_t i) {
        int64_t t0 = window_origin(cen[2 * i + 1], half, slide);
        t0 = std::min<int64_t>(t0, H - 1); // a window that starts past the image reads nothing: any band will do
        return t0 / B;
    };
    std::vector<int64_t> stamp((size_t)nblocks, -1);   // chunk a band was last counted for
    std::vector<uint32_t> band_count((size_t)nblocks, 0);
    std::vector<int64_t> blk_buf((size_t)chunk_cap);
    int64_t plan_next = 0, plan_index = 0;             // first nucleus / index of the chunk the next plan() call makes
    std::vector<int64_t> block_slot((size_t)nblocks, -1); // band -> cache slot (-1: not resident); kept by enqueue_copy
    // RAMP-UP.  The call's first chunks end where their nuclei would need a SECOND band that is not on the device yet: the first chunk
    // then waits for one band (302 MB at 47 703 pixels across: 5 ms) instead of two, and while it is extracted the link delivers exactly
    // the one band the next chunk needs.  After `ramp` chunks the link is ahead for good (it outruns the extraction) and chunks take
    // their full size -- fewer launches, fewer tails.
    int64_t ramp = 2;
    auto cap_of = [&](int64_t) -> int64_t { return chunk_cap; };
#ifdef NFX_EXPERIMENTS // NFX_STREAM_RAMP (0: none)
    if (const char *e = getenv("NFX_STREAM_RAMP")) ramp = atoll(e);
#endif
    const int64_t ramp_min = std::min<int64_t>(chunk_cap, std::max<int64_t>(batch_size, (chunk_cap / 8 / batch_size) * batch_size)); // (never slivers)
    auto plan = [&](Chunk &c) {                        // a single batch that needs more bands than the cache holds is a chunk of its own
        c.i0 = plan_next; c.blocks.clear(); c.sel.clear(); c.sel_off.clear();
        const int64_t ck = plan_index++;
        int64_t i = c.i0, distinct = 0, fresh = 0;    // bands of the chunk | those among them that still have to travel
        const int64_t cap = cap_of(ck);
        while (i < n && i - c.i0 < cap) {
            const int64_t e = std::min<int64_t>(n, i + batch_size);
            int64_t add = 0, add_fresh = 0;
            for (int64_t j = i; j < e; ++j) {
                const int64_t bnd = band_of(j);
                blk_buf[(size_t)(j - c.i0)] = bnd;
                if (stamp[bnd] != ck && stamp[bnd] != -2 - ck) { stamp[bnd] = -2 - ck; ++add; add_fresh += block_slot[bnd] < 0; }
            }
            bool fits = distinct + add <= n_slots;
            if (fits && ck < ramp && fresh + add_fresh > 1 && i - c.i0 >= ramp_min) fits = false; // ramp-up: one travelling band per chunk
            for (int64_t j = i; j < e; ++j) { // settle the tentative marks: counted (the batch joins the chunk) or forgotten
                int64_t &m = stamp[blk_buf[(size_t)(j - c.i0)]];
                if (m == -2 - ck) m = (fits || i == c.i0) ? ck : -1;
            }
            if (!fits && i > c.i0) break;      // this batch opens the next chunk
            distinct += add;
            fresh += add_fresh;
            i = e;
            if (!fits) break;                  // ... one batch wider than the cache: gathered in several passes (enqueue_copy)
        }
        c.i1 = i;
        plan_next = i;
        const int64_t m = c.i1 - c.i0;
        for (int64_t j = 0; j < m; ++j) ++band_count[blk_buf[(size_t)j]];
        for (int64_t bnd = 0; bnd < nblocks; ++bnd)  // (a few dozen bands: a scan is cheaper than collecting and sorting the touched ones)
            if (band_count[bnd]) { c.blocks.push_back(bnd); c.sel_off.push_back((uint32_t)c.sel.size()); c.sel.resize(c.sel.size() + band_count[bnd]); }
        c.sel_off.push_back((uint32_t)c.sel.size());
        {   // scatter: nucleus j goes to the next free place of its band's run (runs in ascending band order, nuclei in index order)
            std::vector<uint32_t> &at = band_count; // reuse as write cursors: at[band] = start of the band's run
            for (size_t q = 0; q < c.blocks.size(); ++q) at[c.blocks[q]] = c.sel_off[q];
            for (int64_t j = 0; j < m; ++j) c.sel[at[blk_buf[(size_t)j]]++] = (uint32_t)j;
            for (size_t q = 0; q < c.blocks.size(); ++q) at[c.blocks[q]] = 0;
        }
    };
    Chunk plans[4]; // chunk k lives in plans[k & 3] from its planning to the enqueue of its extraction (up to `lead` + 1 chunks are alive)

    // ---- pinned staging: selection lists and rebased offsets of two chunks -------------------------------------------------------
    const bool img_pinned = is_pinned(image_u8); // (a pageable image is copied by the runtime through its own staging: correct, not overlapped)
    // (every chunk's lists have a place of their own: the host enqueues far ahead of the device and must never overwrite a list the device has
    // not read yet.  sel: one entry per nucleus; off: one per nucleus + one per chunk, and there are no more chunks than batches)
    const size_t max_chunks = (size_t)((n + batch_size - 1) / batch_size) + 1;
    const size_t stage_bytes = 4 * (size_t)n + 64 + 4 * ((size_t)n + max_chunks) + 64;
    uint8_t *pin = (uint8_t *)Pinned::get(ctx, 1, stage_bytes);
    if (!pin) return api_fail(ctx, NFX_ERR_OUT_OF_MEMORY, "pinned staging");
    uint32_t *const h_sel_all = (uint32_t *)pin, *const h_off_all = (uint32_t *)(pin + ((4 * (size_t)n + 63) & ~(size_t)63));
    auto stage_lists = [&](int64_t k, const Chunk &c, uint32_t *&hs, uint32_t *&ho) { // fills and returns chunk k's staged lists
        const int64_t m = c.i1 - c.i0;
        hs = h_sel_all + c.i0;
        ho = h_off_all + c.i0 + k;
        memcpy(hs, c.sel.data(), 4 * (size_t)m);
        const uint32_t v0 = ring_offsets[c.i0];
        for (int64_t j = 0; j <= m; ++j) ho[j] = ring_offsets[c.i0 + j] - v0;
    };

#ifdef NFX_EXPERIMENTS // MEASUREMENT ONLY (wrong rows unless the call before left the same bands in the same arena): NFX_EXP_STREAM_NOCOPY=1 enqueues no
    // band copy -- what the chunked extraction takes with the link silent (profiles/r05_stream_experiments.txt)
    const bool exp_nocopy = [] { const char *e = getenv("NFX_EXP_STREAM_NOCOPY"); return e && atoi(e) == 1; }();
#else
    constexpr bool exp_nocopy = false;
#endif
    // ---- block cache ---------------------------------------------------------------------------------------------------------
    std::vector<int64_t> slot_block((size_t)n_slots, -1), slot_use((size_t)n_slots, -1);
    int64_t img_h2d = 0, blocks_copied = 0, rows_d2h = 0, use_tick = 0, gather_passes = 0;

    auto fail_sync = [&](nfx_status s) { // an error mid-way: nothing may still be in flight on the private streams when the call returns
        (void)hipStreamSynchronize(copy); (void)hipStreamSynchronize(cs[0]); (void)hipStreamSynchronize(cs[1]); (void)hipStreamSynchronize(rows_s); (void)hipStreamSynchronize(d2h);
        ctx->stream = compute;
        return s;
    };
#define NFX_HIP_S(expr)                                                                                                                  \
    do {                                                                                                                                 \
        hipError_t e_ = (expr);                                                                                                          \
        if (e_ != hipSuccess) {                                                                                                          \
            (void)hipGetLastError(); /* reported here: the next call must not find it again */                                           \
            return fail_sync(api_fail(ctx, e_ == hipErrorOutOfMemory ? NFX_ERR_OUT_OF_MEMORY : NFX_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_))); \
        }                                                                                                                                \
    } while (0)

    // H2D of chunk k's missing bands on the copy stream -- nothing but copies there, so the link never waits for a kernel -- and the
    // gathers of its windows behind them on the rows stream (which also carries the rows home)
    auto enqueue_copy = [&](int64_t k) -> nfx_status {
        const Chunk &c = plans[k & 3];
        const int b = (int)(k % NB);
        const int64_t m = c.i1 - c.i0;
        if (k >= NB) NFX_HIP_S(hipEventSynchronize(ctx->ev_done[b])); // extract(k-3) has finished: patches[b], sel[b], off[b] and their staging are free
        uint32_t *hs, *ho;
        stage_lists(k, c, hs, ho);
        NFX_HIP_S(hipMemcpyAsync(d_sel[b], hs, 4 * (size_t)m, hipMemcpyHostToDevice, copy));
        NFX_HIP_S(hipMemcpyAsync(d_off[b], ho, 4 * ((size_t)m + 1), hipMemcpyHostToDevice, copy));
        other_h2d += 8 * m + 4;
        // The chunk's bands in PASSES of at most n_slots (one pass unless a single batch is wider than the cache): the bands of a pass
        // that are not resident take the slots that have gone unused longest, their copies go out back to back, the windows that start
        // in them are gathered behind the copies; a later pass overwrites slots the pass before it has gathered from and waits for those gathers.
        std::vector<int64_t> slot_of_q(c.blocks.size());
        bool evicted_waited = false;
        for (size_t q0 = 0; q0 < c.blocks.size();) {
            const size_t q1 = std::min(c.blocks.size(), q0 + (size_t)n_slots);
            ++use_tick;
            for (size_t q = q0; q < q1; ++q) if (block_slot[c.blocks[q]] >= 0) slot_use[block_slot[c.blocks[q]]] = use_tick;
            if (q0 > 0) { // the gathers of the pass before read slots this pass may overwrite
                NFX_HIP_S(hipEventRecord(ctx->ev_pass, rows_s));
                NFX_HIP_S(hipStreamWaitEvent(copy, ctx->ev_pass, 0));
                evicted_waited = true; // (the rows stream is in order: that covers the earlier chunks' gathers as well)
            }
            for (size_t q = q0; q < q1; ++q) {
                const int64_t bid = c.blocks[q];
                const int64_t wy0 = bid * B, wh = std::min<int64_t>(H - wy0, B + ps), ww = W;
                int64_t s = block_slot[bid];
                if (s < 0) { // not resident: take the slot that has gone unused longest (never one this pass needs)
                    s = 0;
                    for (int64_t t = 1; t < n_slots; ++t) if (slot_use[t] < slot_use[s]) s = t;
                    if (slot_use[s] == use_tick) return fail_sync(api_fail(ctx, NFX_ERR_DEVICE, "band cache: no free slot within a pass (internal)"));
                    if (slot_block[s] >= 0) {
                        block_slot[slot_block[s]] = -1;
                        // an evicted band may still be read by the gathers of an earlier chunk (they run on the other stream, in order):
                        // the overwriting copy waits for the latest of them
                        if (!evicted_waited && k >= 1) { NFX_HIP_S(hipStreamWaitEvent(copy, ctx->ev_ready[(k - 1) % NB], 0)); evicted_waited = true; }
                    }
                    slot_block[s] = bid; block_slot[bid] = s; slot_use[s] = use_tick;
                    uint8_t *dst = d_slots + (size_t)s * slot_bytes;
                    for (int ch = 0; ch < 3 && !exp_nocopy; ++ch) // rows [wy0, wy0 + wh) of a channel are contiguous in the planar image
                        NFX_HIP_S(hipMemcpyAsync(dst + (size_t)ch * wh * ww, image_u8 + ((size_t)ch * H + wy0) * W, (size_t)wh * ww, hipMemcpyHostToDevice, copy));
                    img_h2d += 3 * wh * ww;
                    ++blocks_copied;
                }
                slot_of_q[q] = s;
            }
            NFX_HIP_S(hipEventRecord(ctx->ev_bands[b], copy));
            NFX_HIP_S(hipStreamWaitEvent(rows_s, ctx->ev_bands[b], 0));
            for (size_t q = q0; q < q1; ++q) {
                const int64_t bid = c.blocks[q];
                const int64_t wy0 = bid * B, wh = std::min<int64_t>(H - wy0, B + ps);
                ctx->stream = rows_s;
                const hipError_t ge = launch_gather_patches_window(ctx, c.sel_off[q + 1] - c.sel_off[q], d_sel[b] + c.sel_off[q], d_cen + 2 * c.i0,
                                                                   d_slots + (size_t)slot_of_q[q] * slot_bytes, H, W, wy0, 0, wh, W, ps, d_patches[b]);
                ctx->stream = compute;
                NFX_HIP_S(ge);
            }
            if (q1 - q0 < c.blocks.size()) ++gather_passes;
            q0 = q1;
        }
        NFX_HIP_S(hipEventRecord(ctx->ev_ready[b], rows_s));
        return NFX_OK;
    };

    // BANDS AHEAD (round 5).  When every band keeps a cache slot of its own (the whole slide fits: the default on 288 GB) nothing is ever evicted,
    // so a chunk's bands depend on nothing but the plan -- they are sent `lead` chunks ahead of the extraction, back to back, and the link
    // never waits for the compute side (before: chunk k+1's bands left only once extract(k-2) had finished, because the same host step
    // also reused that chunk's patch buffer; the link then ran one chunk ahead at best and the next extraction often started behind its
    // bands).  The gathers still wait for their patch buffer, but on the DEVICE (an event of extract(k-3)), not on the host, and the
    // chunk's lists travel on the rows stream right in front of them.  A smaller cache keeps the fused form below (lead 1, enqueue_copy).
    const bool all_resident = n_slots >= nblocks;
    int64_t lead = all_resident ? 2 : 1;
#ifdef NFX_EXPERIMENTS // NFX_STREAM_LEAD = 1..3
    if (const char *e = getenv("NFX_STREAM_LEAD")) lead = std::max<int64_t>(1, std::min<int64_t>(3, atoll(e)));
    if (!all_resident) lead = 1;
#endif
    const bool split = lead > 1;
    auto send_bands = [&](int64_t k) -> nfx_status { // chunk k's bands that are not on the device yet: slot = band (nothing is evicted)
        const Chunk &c = plans[k & 3];
        for (size_t q = 0; q < c.blocks.size(); ++q) {
            const int64_t bid = c.blocks[q];
            if (block_slot[bid] >= 0) continue;
            const int64_t wy0 = bid * B, wh = std::min<int64_t>(H - wy0, B + ps), ww = W;
            block_slot[bid] = bid; slot_block[bid] = bid;
            uint8_t *dst = d_slots + (size_t)bid * slot_bytes;
            for (int ch = 0; ch < 3 && !exp_nocopy; ++ch)
                NFX_HIP_S(hipMemcpyAsync(dst + (size_t)ch * wh * ww, image_u8 + ((size_t)ch * H + wy0) * W, (size_t)wh * ww, hipMemcpyHostToDevice, copy));
            img_h2d += 3 * wh * ww;
            ++blocks_copied;
        }
        NFX_HIP_S(hipEventRecord(ctx->ev_bands[k & 3], copy));
        return NFX_OK;
    };
    auto gather_chunk = [&](int64_t k) -> nfx_status { // (split form) lists up, windows gathered: all on the rows stream, behind the chunk's bands and its buffer's last reader
        const Chunk &c = plans[k & 3];
        const int b = (int)(k % NB);
        const int64_t m = c.i1 - c.i0;
        uint32_t *hs, *ho;
        stage_lists(k, c, hs, ho);
        NFX_HIP_S(hipStreamWaitEvent(rows_s, ctx->ev_bands[k & 3], 0));
        if (k >= NB) NFX_HIP_S(hipStreamWaitEvent(rows_s, ctx->ev_done[b], 0)); // extract(k-3) read patches[b], sel[b], off[b]
        NFX_HIP_S(hipMemcpyAsync(d_sel[b], hs, 4 * (size_t)m, hipMemcpyHostToDevice, rows_s));
        NFX_HIP_S(hipMemcpyAsync(d_off[b], ho, 4 * ((size_t)m + 1), hipMemcpyHostToDevice, rows_s));
        other_h2d += 8 * m + 4;
        for (size_t q = 0; q < c.blocks.size(); ++q) {
            const int64_t bid = c.blocks[q];
            const int64_t wy0 = bid * B, wh = std::min<int64_t>(H - wy0, B + ps);
            ctx->stream = rows_s;
            const hipError_t ge = launch_gather_patches_window(ctx, c.sel_off[q + 1] - c.sel_off[q], d_sel[b] + c.sel_off[q], d_cen + 2 * c.i0,
                                                               d_slots + (size_t)block_slot[bid] * slot_bytes, H, W, wy0, 0, wh, W, ps, d_patches[b]);
            ctx->stream = compute;
            NFX_HIP_S(ge);
        }
        NFX_HIP_S(hipEventRecord(ctx->ev_ready[b], rows_s));
        return NFX_OK;
    };

    // Order of the host's work per chunk k (the loop below): enqueue extract(k) -- its windows were gathered while chunk k-1 ran --, then plan
    // chunk k + lead and send its bands on their way (split form; in the fused form enqueue_copy does that for chunk k+1 after a HOST wait for
    // extract(k-2), whose buffers it takes over), then the lists and gathers of chunk k+1 on the rows stream, behind a DEVICE-side wait for
    // their patch buffer's last reader.  Two extractions are always queued, so the device never runs dry while the host plans.
    // The centroid frontier runs two chunks ahead of the plan: plan(j) reads the centroids of [plan_next, plan_next + cap(j)); right
    // after it the frontier is pushed to plan_next + cap(j+1) + cap(j+2) -- enqueued AHEAD of chunk j's bands on the copy stream (25 MB of rings
    // before 300-600 MB of bands), so that by the time the host comes to plan(j + 1) it waits for nothing that lies behind those bands.
    int64_t nchunks = 0, planned = 0;  // chunks whose plan exists (and, in the split form, whose bands are on their way)
    auto plan_next_chunk = [&]() -> nfx_status {
        if (planned == 0) NFX_HIP_S(advance_front(cap_of(0)));
        NFX_HIP_S(hipEventSynchronize(ctx->ev_front)); // centroids of [0, plan_next + this chunk's cap) and beyond are home
        plan(plans[planned & 3]);
        NFX_HIP_S(advance_front(plan_next + cap_of(planned + 1) + cap_of(planned + 2)));
        if (split) { const nfx_status s2 = send_bands(planned); if (s2 != NFX_OK) return s2; }
        ++planned;
        return NFX_OK;
    };
    while (planned < lead && plan_next < n)
        if ((st = plan_next_chunk()) != NFX_OK) return st;
    if (planned > 0 && (st = split ? gather_chunk(0) : enqueue_copy(0)) != NFX_OK) return st;
    for (int64_t k = 0; k < planned; ++k) {
        const int b = (int)(k % NB);
        hipStream_t const cstream = cs[k & 1];
        const int64_t i0 = plans[k & 3].i0, i1 = plans[k & 3].i1, m = i1 - i0;
        ++nchunks;
        NFX_HIP_S(hipStreamWaitEvent(cstream, ctx->ev_ready[b], 0));
        if (k >= NB && !rows_direct) NFX_HIP_S(hipStreamWaitEvent(cstream, ctx->ev_rows[b], 0)); // rows[b] of chunk k-3 have left for the host
        ctx->ws->off = ws_mark[k & 1];
        ctx->stream = cstream;
        const uint32_t v0 = ring_offsets[i0], v1 = ring_offsets[i1];
        st = extract_device(ctx, sets, n_sets, m, (int64_t)(v1 - v0), d_off[b], d_xy + 2 * (size_t)v0, d_patches[b], nullptr, 0, 0, ps, batch_size,
                            rows_direct ? rows_direct + (size_t)i0 * F : d_rows[b], nullptr);
        ctx->stream = compute;
        if (st != NFX_OK) return fail_sync(st);
        NFX_HIP_S(hipEventRecord(ctx->ev_done[b], cstream));
        if (!rows_direct) {
            NFX_HIP_S(hipStreamWaitEvent(d2h, ctx->ev_done[b], 0));
            NFX_HIP_S(hipMemcpyAsync(out_rows + (size_t)i0 * F, d_rows[b], 4 * (size_t)m * F, hipMemcpyDeviceToHost, d2h));
            NFX_HIP_S(hipEventRecord(ctx->ev_rows[b], d2h));
        }
        rows_d2h += 4 * m * F;
        if (plan_next < n && (st = plan_next_chunk()) != NFX_OK) return st;                                    // chunk k + lead: plan, bands on their way
        if (k + 1 < planned && (st = split ? gather_chunk(k + 1) : enqueue_copy(k + 1)) != NFX_OK) return st;  // chunk k + 1: windows into its patch buffer
    }
    NFX_HIP_S(advance_front(n)); // (nuclei behind the last chunk's end: none -- every nucleus belongs to a chunk; kept for the invariant front == n)
#undef NFX_HIP_S
    NFX_HIP(ctx, hipStreamSynchronize(copy));
    NFX_HIP(ctx, hipStreamSynchronize(cs[0]));
    NFX_HIP(ctx, hipStreamSynchronize(cs[1]));
    NFX_HIP(ctx, hipStreamSynchronize(rows_s));
    NFX_HIP(ctx, hipStreamSynchronize(d2h));
    if (out_centroids_xy) memcpy(out_centroids_xy, cen, 8 * (size_t)n);
    {
        nfx_stream_stats full{};
        full.image_bytes_h2d = img_h2d; full.other_bytes_h2d = other_h2d; full.rows_bytes_d2h = rows_d2h + 8 * n;
        full.blocks_copied = blocks_copied; full.blocks_total = nblocks; full.chunks = nchunks; full.block_slots = n_slots;
        full.image_pinned = img_pinned ? 1 : 0; full.split_gather_passes = gather_passes; full.rows_direct = rows_direct ? 1 : 0;
        put_stats(full);
    }
    return NFX_OK;
}

} // extern "C"
