// nfx_api.cpp -- the C ABI declared in include/nfx.h: registry, context, staging and the
// orchestration of the HIP kernels.  Compiled with hipcc (host code only in this file).
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <strings.h>

#include "nfx_internal.h"

using namespace nfx;

namespace nfx {
const char *const kKernelSlotNames[KS_COUNT] = {"k_preprocess", "k_raster_shape", "k_polygon",  "k_pack_masks", "k_gather_patches",
                                                "k_hue_batch",  "k_color",        "k_glcm",     "k_glrlm",      "k_gabor"};
}

// ------------------------------------------------------------------------------------------------
// registry: names exactly as the reference emits them
// ------------------------------------------------------------------------------------------------
static const char *const kSetNames[NFX_SET_COUNT] = {"geometry", "color", "GLCM", "GLRLM", "gabor filter"};
static const char *const kSetCliNames[NFX_SET_COUNT] = {"geometry", "color", "glcm", "glrlm", "gabor"};

static std::vector<std::string> g_cols[NFX_SET_COUNT];
static std::once_flag g_cols_once;

static std::string rust_f32_display(float v)
{
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
    char buf[128];
    auto r = std::to_chars(buf, buf + sizeof(buf), v, std::chars_format::fixed);
    return std::string(buf, r.ptr);
}

static void build_columns()
{
    // shape.rs:113-128
    g_cols[0] = {"area", "major_axis", "minor_axis", "eccentricity", "orientation", "perimeter", "equivalent_perimeter",
                 "compacity", "eliptic_deviation", "convex_hull_area", "convex_deffect", "convex_perimeter"};
    // color.rs:80-99
    g_cols[1] = {"mean_r", "mean_g", "mean_b", "std_r", "std_g", "std_b", "mean_h", "mean_s", "mean_v",
                 "std_h", "std_s", "std_v", "mean_haematoxylin", "mean_eosin", "mean_dab", "std_haematoxylin",
                 "std_eosin", "std_dab"};
    // texture.rs:19-20,81-157
    static const int levels[4] = {32, 64, 128, 254};
    static const int offs[4][2] = {{0, 1}, {1, 1}, {1, 0}, {1, -1}};
    static const char *const gf[14] = {"correlation", "contrast", "dissimilarity", "entropy", "angular_second_moment",
                                       "sum_average", "sum_variance", "sum_entropy", "sum_of_squares",
                                       "inverse_difference_moment", "difference_average", "difference_variance",
                                       "information_measure_correlation1", "information_measure_correlation2"};
    for (int l = 0; l < 4; ++l)
        for (int o = 0; o < 4; ++o)
            for (int f = 0; f < 14; ++f) {
                char b[128];
                snprintf(b, sizeof b, "%s_%d_%d_%d", gf[f], offs[o][0], offs[o][1], levels[l]);
                g_cols[2].push_back(b);
            }
    // texture.rs:176,244-301
    static const int dirs[4][2] = {{1, 0}, {1, 1}, {0, 1}, {-1, 1}};
    static const char *const rf[17] = {"short_run_emphasis", "long_run_emphasis", "gray_level_nonuniformity",
                                       "run_length_nonuniformity", "low_gray_level_run_emphasis",
                                       "high_gray_level_run_emphasis", "short_run_low_gray_level_emphasis",
                                       "short_run_high_gray_level_emphasis", "long_run_low_gray_level_emphasis",
                                       "long_run_high_gray_level_emphasis", "short_run_mid_gray_level_emphasis",
                                       "long_run_mid_gray_level_emphasis", "short_run_extreme_gray_level_emphasis",
                                       "long_run_extreme_gray_level_emphasis", "run_percentage", "run_length_mean",
                                       "run_length_variance"};
    for (int d = 0; d < 4; ++d)
        for (int f = 0; f < 17; ++f) {
            char b[128];
            snprintf(b, sizeof b, "%s_%d_%d", rf[f], dirs[d][0], dirs[d][1]);
            g_cols[3].push_back(b);
        }
    // texture.rs:319-320,346-361: angle = (j / 6) as f32 * 45.0, frequency = FREQUENCIES[j % 6] as f32
    static const float freqs[6] = {0.5f, 1.0f, 2.0f, 4.0f, 6.0f, 8.0f};
    for (int j = 0; j < 48; ++j) {
        const std::string a = rust_f32_display((float)(j / 6) * 45.0f), f = rust_f32_display(freqs[j % 6]);
        g_cols[4].push_back("gabor_angle_" + a + "_frequency_" + f + "_mean");
        g_cols[4].push_back("gabor_angle_" + a + "_frequency_" + f + "_variance");
    }
}

extern "C" {

int nfx_version(void) { return NFX_VERSION; }
int nfx_feature_set_count(void) { return NFX_SET_COUNT; }
const char *nfx_feature_set_name(int set) { return (set >= 0 && set < NFX_SET_COUNT) ? kSetNames[set] : nullptr; }
const char *nfx_feature_set_cli_name(int set) { return (set >= 0 && set < NFX_SET_COUNT) ? kSetCliNames[set] : nullptr; }
int nfx_feature_set_num_columns(int set) { return (set >= 0 && set < NFX_SET_COUNT) ? kNumCols[set] : -1; }

int nfx_feature_set_column_name(int set, int j, char *buf, size_t buflen)
{
    if (set < 0 || set >= NFX_SET_COUNT || !buf) return -1;
    std::call_once(g_cols_once, build_columns);
    if (j < 0 || j >= (int)g_cols[set].size()) return -1;
    const std::string &s = g_cols[set][j];
    if (s.size() + 1 > buflen) return -1;
    memcpy(buf, s.c_str(), s.size() + 1);
    return (int)s.size();
}

int nfx_feature_set_column_dtype(int set, int j)
{
    if (set < 0 || set >= NFX_SET_COUNT || j < 0 || j >= kNumCols[set]) return -1;
    // shape.rs:116-127: area, perimeter, equivalent_perimeter, compacity, convex_hull_area, convex_deffect, convex_perimeter
    // are geometric_features' return values; everything else in every set was read out of an f32 tensor
    static const bool poly[12] = {true, false, false, false, false, true, true, true, false, true, true, true};
    return (set == NFX_SET_GEOMETRY && poly[j]) ? NFX_DTYPE_F64 : NFX_DTYPE_F32;
}

int nfx_parse_feature_sets(const char *const *names, int n_names, int *out_sets, int cap)
{
    int m = 0;
    auto push = [&](int s) {
        if (m < cap) out_sets[m] = s;
        ++m;
    };
    for (int i = 0; i < n_names; ++i) {
        const char *s = names[i];
        if (!s) return -1;
        if (!strcasecmp(s, "geometry")) push(NFX_SET_GEOMETRY);
        else if (!strcasecmp(s, "color")) push(NFX_SET_COLOR);
        else if (!strcasecmp(s, "glcm")) push(NFX_SET_GLCM);
        else if (!strcasecmp(s, "glrlm")) push(NFX_SET_GLRLM);
        else if (!strcasecmp(s, "gabor")) push(NFX_SET_GABOR);
        else if (!strcasecmp(s, "texture")) { push(NFX_SET_GLCM); push(NFX_SET_GLRLM); push(NFX_SET_GABOR); }
        else if (!strcasecmp(s, "all")) { for (int k = 0; k < NFX_SET_COUNT; ++k) push(k); }
        else return -1;
    }
    return m <= cap ? m : -2;
}

int nfx_centroid_key(float x, float y, char *buf, size_t buflen)
{
    const std::string s = rust_f32_display(x) + "," + rust_f32_display(y);
    if (!buf || s.size() + 1 > buflen) return -1;
    memcpy(buf, s.c_str(), s.size() + 1);
    return (int)s.size();
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// context, workspace, timing
// ------------------------------------------------------------------------------------------------
namespace nfx {
nfx_status api_fail(nfx_ctx *ctx, nfx_status st, const std::string &msg)
{
    if (ctx) ctx->err = msg;
    return st;
}
}  // namespace nfx
static nfx_status fail(nfx_ctx *ctx, nfx_status st, const std::string &msg) { return nfx::api_fail(ctx, st, msg); }

// (NFX_HIP is defined in nfx_internal.h)

namespace nfx {

bool stream_capturing(nfx_ctx *ctx)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(ctx->stream, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
    return st != hipStreamCaptureStatusNone;
}

bool tables_locked(nfx_ctx *ctx) { return (ctx->reserved && ctx->call_mem == NFX_MEM_DEVICE) || stream_capturing(ctx); }

// Reallocates an arena (synchronise + free + malloc).  For dev_ws this moves addresses a captured graph may hold: the callers
// below decide when that is allowed, and every such move is counted (nfx_ctx_workspace_generation).
static nfx_status arena_grow(nfx_ctx *ctx, Arena &a, size_t bytes)
{
    if (a.base) {
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
        (void)hipFree(a.base);
        a.base = nullptr;
        a.cap = 0;
        if (&a == &ctx->dev_ws) ++ctx->ws_generation;
    }
    const size_t cap = bytes + (bytes >> 3) + (1u << 20);
    if (hipMalloc(reinterpret_cast<void **>(&a.base), cap) != hipSuccess) {
        (void)hipGetLastError();
        ctx->err = "device workspace: out of memory";
        return NFX_ERR_OUT_OF_MEMORY;
    }
    a.cap = cap;
    return NFX_OK;
}

nfx_status ws_reserve(nfx_ctx *ctx, size_t bytes, nfx_mem mem)
{
    Arena &a = *ctx->ws; // host_ws for a NFX_MEM_HOST call (StreamScope), dev_ws otherwise
    a.off = 0;
    if (mem == NFX_MEM_DEVICE) {
        const bool capturing = stream_capturing(ctx);
        if (capturing) ctx->capture_seen = true;
        if (bytes <= a.cap) return NFX_OK;
        // stream-ordered calls promise not to synchronise or allocate (nfx.h): honoured strictly once the caller has made a
        // reservation, and always during a capture (a hipMalloc there would invalidate the capture anyway)
        if (ctx->reserved) {
            ctx->err = "device-memory call exceeds the context's reservation (" + std::to_string(bytes) + " > " + std::to_string(a.cap) +
                       " bytes): call nfx_ctx_reserve for this shape first";
            return NFX_ERR_OUT_OF_MEMORY;
        }
        if (capturing) {
            ctx->err = "device-memory call during stream capture needs a workspace that is not there yet: call nfx_ctx_reserve before capturing";
            return NFX_ERR_INVALID_ARGUMENT;
        }
        if (ctx->capture_seen) {
            // a graph captured earlier on this context holds addresses inside the workspace: growing it here, silently, would
            // leave that graph replaying into freed memory.  nfx_ctx_reserve is the explicit way (it counts the move).
            ctx->err = "device-memory call needs a larger workspace (" + std::to_string(bytes) + " > " + std::to_string(a.cap) +
                       " bytes) on a context a graph was captured from: call nfx_ctx_reserve for this shape and capture again";
            return NFX_ERR_OUT_OF_MEMORY;
        }
    } else if (bytes <= a.cap) {
        return NFX_OK;
    }
    return arena_grow(ctx, a, bytes);
}


void *ws_alloc(nfx_ctx *ctx, size_t bytes)
{
    Arena &w = *ctx->ws;
    const size_t a = al(bytes);
    if (w.off + a > w.cap) return nullptr;
    void *p = w.base + w.off;
    w.off += a;
    return p;
}

static hipEvent_t get_event(nfx_ctx *ctx)
{
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

void timing_begin(nfx_ctx *ctx, int slot)
{
    if (!ctx->timing) return;
    Timing t{get_event(ctx), get_event(ctx), slot};
    (void)hipEventRecord(t.start, ctx->stream);
    ctx->pending.push_back(t);
}

void timing_end(nfx_ctx *ctx)
{
    if (!ctx->timing || ctx->pending.empty()) return;
    (void)hipEventRecord(ctx->pending.back().stop, ctx->stream);
}

static void timing_collect(nfx_ctx *ctx)
{
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &t : ctx->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.start, t.stop) == hipSuccess) {
            ctx->total_ms[t.slot] += ms;
            ctx->launches[t.slot] += 1;
        }
        ctx->event_pool.push_back(t.start);
        ctx->event_pool.push_back(t.stop);
    }
    ctx->pending.clear();
}

}  // namespace nfx

extern "C" {

nfx_status nfx_ctx_create(int device, nfx_ctx **out)
{
    if (!out) return NFX_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return NFX_ERR_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return NFX_ERR_DEVICE;
    nfx_ctx *c = new nfx_ctx();
    c->device = device;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
        nfx_ctx_destroy(c);
        return NFX_ERR_DEVICE;
    }
    *out = c;
    return NFX_OK;
}

void nfx_ctx_destroy(nfx_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->own_stream) { (void)hipStreamSynchronize(ctx->own_stream); (void)hipStreamDestroy(ctx->own_stream); }
    if (ctx->side_stream) { (void)hipStreamSynchronize(ctx->side_stream); (void)hipStreamDestroy(ctx->side_stream); }
    if (ctx->copy_stream) { (void)hipStreamSynchronize(ctx->copy_stream); (void)hipStreamDestroy(ctx->copy_stream); }
    if (ctx->alt_stream) { (void)hipStreamSynchronize(ctx->alt_stream); (void)hipStreamDestroy(ctx->alt_stream); }
    if (ctx->d2h_stream) { (void)hipStreamSynchronize(ctx->d2h_stream); (void)hipStreamDestroy(ctx->d2h_stream); }
    if (ctx->rows_stream) { (void)hipStreamSynchronize(ctx->rows_stream); (void)hipStreamDestroy(ctx->rows_stream); }
    for (int b = 0; b < 3; ++b)
        for (hipEvent_t e : {ctx->ev_ready[b], ctx->ev_done[b], ctx->ev_rows[b]})
            if (e) (void)hipEventDestroy(e);
    for (int b = 0; b < 4; ++b)
        if (ctx->ev_bands[b]) (void)hipEventDestroy(ctx->ev_bands[b]);
    if (ctx->ev_pass) (void)hipEventDestroy(ctx->ev_pass);
    for (hipEvent_t e : {ctx->ev_front, ctx->ev_front_up, ctx->ev_front_cen})
        if (e) (void)hipEventDestroy(e);
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    if (ctx->pin_cen) (void)hipHostFree(ctx->pin_cen);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    for (auto &t : ctx->pending) { (void)hipEventDestroy(t.start); (void)hipEventDestroy(t.stop); }
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->dev_ws.base) (void)hipFree(ctx->dev_ws.base);
    if (ctx->host_ws.base) (void)hipFree(ctx->host_ws.base);
    if (ctx->gabor_tab) (void)hipFree(ctx->gabor_tab);
    if (ctx->od_lut) (void)hipFree(ctx->od_lut);
    if (ctx->glcm_lut) (void)hipFree(ctx->glcm_lut);
    delete ctx;
}

const char *nfx_last_error(const nfx_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

nfx_status nfx_ctx_set_stream(nfx_ctx *ctx, void *hip_stream)
{
    if (!ctx) return NFX_ERR_INVALID_ARGUMENT;
    ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
    return NFX_OK;
}

nfx_status nfx_ctx_set_patch_addressing(nfx_ctx *ctx, nfx_addressing mode)
{
    if (!ctx) return NFX_ERR_INVALID_ARGUMENT;
    if (mode != NFX_ADDR_IMAGE && mode != NFX_ADDR_SLIDE) return fail(ctx, NFX_ERR_INVALID_ARGUMENT, "unknown patch addressing mode");
    ctx->addressing = mode;
    return NFX_OK;
}

uint64_t nfx_ctx_workspace_generation(const nfx_ctx *ctx) { return ctx ? ctx->ws_generation : 0; }

nfx_status nfx_ctx_synchronize(nfx_ctx *ctx)
{
    if (!ctx) return NFX_ERR_INVALID_ARGUMENT;
    NFX_HIP(ctx, hipSetDevice(ctx->device));
    NFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NFX_OK;
}

nfx_status nfx_ctx_enable_timing(nfx_ctx *ctx, int on)
{
    if (!ctx) return NFX_ERR_INVALID_ARGUMENT;
    if (!on && ctx->timing) timing_collect(ctx);
    ctx->timing = on != 0;
    return NFX_OK;
}

int nfx_ctx_kernel_timing(nfx_ctx *ctx, int k, const char **name, int64_t *launches, double *total_ms)
{
    if (!ctx) return -1;
    timing_collect(ctx);
    if (k >= 0 && k < KS_COUNT) {
        if (name) *name = kKernelSlotNames[k];
        if (launches) *launches = ctx->launches[k];
        if (total_ms) *total_ms = ctx->total_ms[k];
    }
    return KS_COUNT;
}

nfx_status nfx_ctx_set_option(nfx_ctx *ctx, int option, int64_t value)
{
    if (!ctx) return NFX_ERR_INVALID_ARGUMENT;
    switch (option) {
    case NFX_OPT_STREAM_ROWS_STAGED:
        if (value != 0 && value != 1) return fail(ctx, NFX_ERR_INVALID_ARGUMENT, "NFX_OPT_STREAM_ROWS_STAGED takes 0 or 1");
        ctx->opt_stream_rows_staged = value == 1;
        return NFX_OK;
    default:
        return fail(ctx, NFX_ERR_INVALID_ARGUMENT, "unknown option " + std::to_string(option));
    }
}

nfx_status nfx_ctx_reset_timing(nfx_ctx *ctx)
{
    if (!ctx) return NFX_ERR_INVALID_ARGUMENT;
    timing_collect(ctx);
    for (int k = 0; k < KS_COUNT; ++k) { ctx->launches[k] = 0; ctx->total_ms[k] = 0.0; }
    return NFX_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// device-resident orchestration
// ------------------------------------------------------------------------------------------------
namespace nfx {
nfx_status check_common(nfx_ctx *ctx, int64_t n, int64_t nv, int ps)
{
    if (!ctx) return NFX_ERR_INVALID_ARGUMENT;
    if (n < 0 || nv < 0) return fail(ctx, NFX_ERR_INVALID_ARGUMENT, "negative size");
    if (ps < 1) return fail(ctx, NFX_ERR_INVALID_ARGUMENT, "patch size must be >= 1");
    if (ps > NFX_MAX_PATCH_SIZE) return fail(ctx, NFX_ERR_UNSUPPORTED, "patch size > " + std::to_string(NFX_MAX_PATCH_SIZE) + " is not supported");
    if (nv > 0xFFFFFFFFll) return fail(ctx, NFX_ERR_UNSUPPORTED, "more than 2^32-1 vertices in one call");
    return NFX_OK;
}

// host-memory offsets are validated before anything is launched: the kernels index the vertex array with them
nfx_status check_offsets(nfx_ctx *ctx, const uint32_t *off, int64_t n, int64_t nv, const char *what)
{
    if (off[0] != 0u) return fail(ctx, NFX_ERR_INVALID_ARGUMENT, std::string(what) + "[0] != 0");
    for (int64_t i = 0; i < n; ++i)
        if (off[i + 1] < off[i]) return fail(ctx, NFX_ERR_INVALID_ARGUMENT, std::string(what) + " is not non-decreasing at " + std::to_string(i + 1));
    if (off[n] != (uint32_t)nv) return fail(ctx, NFX_ERR_INVALID_ARGUMENT, std::string(what) + "[n] != n_vertices");
    return NFX_OK;
}

int sets_total_columns(const int *sets, int n_sets)
{
    int F = 0;
    for (int s = 0; s < n_sets; ++s) {
        if (sets[s] < 0 || sets[s] >= NFX_SET_COUNT) return -1;
        F += kNumCols[sets[s]];
    }
    return F;
}

bool any_pixel_set(const int *sets, int n_sets)
{
    for (int s = 0; s < n_sets; ++s)
        if (sets[s] != NFX_SET_GEOMETRY) return true;
    return false;
}

int count_set(const int *sets, int n_sets, int which)
{
    int c = 0;
    for (int s = 0; s < n_sets; ++s) c += sets[s] == which;
    return c;
}

// workspace bytes needed by extract_device (excluding host staging): mirrors its ws_alloc calls one for one
size_t extract_ws_bytes(const int *sets, int n_sets, int64_t n, int64_t nv, int ps, int batch_size, bool need_centroids,
                               bool gather)
{
    size_t b = 0;
    b += al(sizeof(float) * 2 * (size_t)nv);                     // centred polygons
    if (need_centroids) b += al(sizeof(float) * 2 * (size_t)n);
    if (any_pixel_set(sets, n_sets)) b += al(sizeof(uint64_t) * (size_t)n * ps * mask_words_per_row(ps));
    if (count_set(sets, n_sets, NFX_SET_GLCM)) b += al(sizeof(int) * (size_t)n); // mask areas
    if (count_set(sets, n_sets, NFX_SET_GEOMETRY)) b += al(polygon_scratch_bytes(n, nv));
    if (gather) b += al((size_t)n * 3 * ps * ps); // packed patches gathered from the resident image
    Job j; j.n = n; j.ps = ps; j.batch_size = batch_size;
    b += (size_t)count_set(sets, n_sets, NFX_SET_COLOR) * al(color_scratch_bytes(j)); // a repeated "color" gets scratch of its own
    if (count_set(sets, n_sets, NFX_SET_GLCM)) b += al(glcm_scratch_bytes(j));
    if (count_set(sets, n_sets, NFX_SET_GABOR)) b += al(gabor_scratch_bytes(j));
    if (count_set(sets, n_sets, NFX_SET_GLRLM)) b += al(glrlm_scratch_bytes(j));
    return b + 4096;
}

nfx_status prepare_sets(nfx_ctx *ctx, const int *sets, int n_sets)
{
    if (count_set(sets, n_sets, NFX_SET_COLOR)) NFX_HIP(ctx, prepare_color(ctx));
    if (count_set(sets, n_sets, NFX_SET_GLCM)) NFX_HIP(ctx, prepare_glcm(ctx));
    if (count_set(sets, n_sets, NFX_SET_GABOR)) NFX_HIP(ctx, prepare_gabor(ctx));
    return NFX_OK;
}

// everything on the device; ws already reserved.  `forked` tells the caller that the side stream holds work to be joined.
static nfx_status extract_device_body(nfx_ctx *ctx, const int *sets, int n_sets, int64_t n, int64_t nv, const uint32_t *ring_off,
                                      const float *ring_xy, const uint8_t *patches_u8, const uint8_t *image, int64_t H, int64_t W,
                                      int ps, int batch_size, float *out_rows, float *out_centroids, bool &forked)
{
    const int F = sets_total_columns(sets, n_sets);
    float *centred = (float *)ws_alloc(ctx, sizeof(float) * 2 * (size_t)nv);
    float *centroids = out_centroids;
    if (!centroids && image) centroids = (float *)ws_alloc(ctx, sizeof(float) * 2 * (size_t)n);
    uint64_t *rows = nullptr;
    const bool pix = any_pixel_set(sets, n_sets);
    if (pix) rows = (uint64_t *)ws_alloc(ctx, sizeof(uint64_t) * (size_t)n * ps * mask_words_per_row(ps));
    int *areas = count_set(sets, n_sets, NFX_SET_GLCM) ? (int *)ws_alloc(ctx, sizeof(int) * (size_t)n) : nullptr;
    if (!centred || (pix && !rows) || (image && !centroids) || (count_set(sets, n_sets, NFX_SET_GLCM) && !areas))
        return fail(ctx, NFX_ERR_OUT_OF_MEMORY, "workspace exhausted");
    // a missing table is built here, before anything is enqueued (refused on a reserved context and during capture)
    const nfx_status pst = prepare_sets(ctx, sets, n_sets);
    if (pst != NFX_OK) return fail(ctx, pst, "device tables of the selected sets are not built: call nfx_ctx_reserve first (" + ctx->err + ")");

    NFX_HIP(ctx, launch_preprocess(ctx, n, ring_off, ring_xy, centroids, centred));
    // fork: the polygon columns only need the centred rings; their kernel runs on the side stream next to everything below
    // and is joined at the end of the call
    hipStream_t const main_stream = ctx->stream;
    const bool want_fork = count_set(sets, n_sets, NFX_SET_GEOMETRY) && pix; // next to the rasteriser alone (geometry only) the two kernels just share the issue slots: no gain (measured again in round 2: 0.492 vs 0.479 ms)
    if (want_fork) {
        NFX_HIP(ctx, hipEventRecord(ctx->ev_fork, main_stream));
        NFX_HIP(ctx, hipStreamWaitEvent(ctx->side_stream, ctx->ev_fork, 0));
        forked = true;
    }
    if (image && pix) {
        uint8_t *gp = (uint8_t *)ws_alloc(ctx, (size_t)n * 3 * ps * ps);
        if (!gp) return fail(ctx, NFX_ERR_OUT_OF_MEMORY, "workspace exhausted");
        // the windows are MATERIALISED as aligned packed patches: reading them in place was built and measured slower (the consumers' loads at
        // any alignment cost more than the one copy saves: profiles/EXPERIMENTS.md, "windows in place")
        NFX_HIP(ctx, launch_gather_patches_window(ctx, n, nullptr, centroids, image, H, W, 0, 0, H, W, ps, gp));
        patches_u8 = gp;
    }
    // rasterise once; the mask-derived geometry columns are produced in the same kernel
    int geo_col = -1;
    {
        int col = 0;
        for (int s = 0; s < n_sets; ++s) {
            if (sets[s] == NFX_SET_GEOMETRY && geo_col < 0) geo_col = col;
            col += kNumCols[sets[s]];
        }
    }
    NFX_HIP(ctx, launch_raster_shape(ctx, n, ring_off, centred, ps, rows, geo_col >= 0 ? out_rows : nullptr, F, geo_col >= 0 ? geo_col : 0, areas));
    Job job;
    job.n = n; job.ps = ps; job.batch_size = batch_size; job.poly_off = ring_off; job.poly_xy = centred;
    job.patches = patches_u8; job.patches_f32 = false; job.mask_rows = rows; job.areas = areas; job.out = out_rows; job.F = F;

    int col = 0;
    bool geo_done = false;
    void *poly_scr = nullptr, *glcm_scr = nullptr, *gabor_scr = nullptr, *glrlm_scr = nullptr;
    for (int s = 0; s < n_sets; ++s) {
        switch (sets[s]) {
        case NFX_SET_GEOMETRY: {
            if (!poly_scr) poly_scr = ws_alloc(ctx, polygon_scratch_bytes(n, nv));
            void *scr = poly_scr;
            if (!scr) return fail(ctx, NFX_ERR_OUT_OF_MEMORY, "workspace exhausted");
            if (geo_done) // a repeated "geometry" (args.rs does not de-duplicate): recompute into its own columns
                NFX_HIP(ctx, launch_raster_shape(ctx, n, ring_off, centred, ps, nullptr, out_rows, F, col));
            if (forked) ctx->stream = ctx->side_stream; // launch (and time) on the side stream
            const hipError_t pe = launch_polygon(ctx, n, ring_off, centred, scr, out_rows, F, col, nullptr, forked);
            ctx->stream = main_stream;
            NFX_HIP(ctx, pe);
            geo_done = true;
            break;
        }
        case NFX_SET_COLOR: {
            void *scr = ws_alloc(ctx, color_scratch_bytes(job));
            if (!scr) return fail(ctx, NFX_ERR_OUT_OF_MEMORY, "workspace exhausted");
            NFX_HIP(ctx, launch_color(ctx, job, col, scr));
            break;
        }
        case NFX_SET_GLCM: {
            if (!glcm_scr) glcm_scr = ws_alloc(ctx, glcm_scratch_bytes(job));
            if (!glcm_scr) return fail(ctx, NFX_ERR_OUT_OF_MEMORY, "workspace exhausted");
            NFX_HIP(ctx, launch_glcm(ctx, job, col, glcm_scr));
            break;
        }
        case NFX_SET_GLRLM: {
            if (!glrlm_scr && glrlm_scratch_bytes(job)) glrlm_scr = ws_alloc(ctx, glrlm_scratch_bytes(job)); // (level maps of patches above 128 x 128 only)
            if (!glrlm_scr && glrlm_scratch_bytes(job)) return fail(ctx, NFX_ERR_OUT_OF_MEMORY, "workspace exhausted");
            NFX_HIP(ctx, launch_glrlm(ctx, job, col, glrlm_scr));
            break;
        }
        case NFX_SET_GABOR: {
            if (!gabor_scr) gabor_scr = ws_alloc(ctx, gabor_scratch_bytes(job)); // (a repeated "gabor" runs after the first on the same stream: shared)
            if (!gabor_scr && gabor_scratch_bytes(job)) return fail(ctx, NFX_ERR_OUT_OF_MEMORY, "workspace exhausted");
            NFX_HIP(ctx, launch_gabor(ctx, job, col, gabor_scr));
            break;
        }
        }
        col += kNumCols[sets[s]];
    }
    return NFX_OK;
}

nfx_status extract_device(nfx_ctx *ctx, const int *sets, int n_sets, int64_t n, int64_t nv, const uint32_t *ring_off,
                          const float *ring_xy, const uint8_t *patches_u8, const uint8_t *image, int64_t H, int64_t W,
                                 int ps, int batch_size, float *out_rows, float *out_centroids)
{
    bool forked = false;
    hipStream_t const main_stream = ctx->stream;
    nfx_status st = extract_device_body(ctx, sets, n_sets, n, nv, ring_off, ring_xy, patches_u8, image, H, W, ps, batch_size,
                                        out_rows, out_centroids, forked);
    ctx->stream = main_stream;
    if (forked) { // join on every path: an error return must not leave the side stream dangling off the caller's stream (or capture)
        const hipError_t e1 = hipEventRecord(ctx->ev_join, ctx->side_stream);
        const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(main_stream, ctx->ev_join, 0) : e1;
        if (st == NFX_OK && e2 != hipSuccess) return fail(ctx, NFX_ERR_DEVICE, std::string("joining the side stream: ") + hipGetErrorString(e2));
    }
    return st;
}

}  // namespace nfx

// host staging helper: copies src (host) into the workspace
template <typename T>
static T *stage_in(nfx_ctx *ctx, const T *src, size_t count)
{
    T *d = (T *)ws_alloc(ctx, sizeof(T) * (count ? count : 1));
    if (d && count && hipMemcpyAsync(d, src, sizeof(T) * count, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return nullptr;
    return d;
}

// workspace of one trait-entry call in device memory
static size_t trait_ws_bytes(int set, int64_t n, int64_t nv, int ps)
{
    Job job;
    job.n = n; job.ps = ps; job.batch_size = (int)(n > 0 ? n : 1);
    size_t need = al(8 * (size_t)n * ps * mask_words_per_row(ps) + 8) + 4096;
    if (set == NFX_SET_GEOMETRY) need += al(polygon_scratch_bytes(n, nv));
    if (set == NFX_SET_COLOR) need += al(color_scratch_bytes(job));
    if (set == NFX_SET_GLCM) need += al(glcm_scratch_bytes(job));
    if (set == NFX_SET_GABOR) need += al(gabor_scratch_bytes(job));
    if (set == NFX_SET_GLRLM) need += al(glrlm_scratch_bytes(job));
    return need;
}

extern "C" {

nfx_status nfx_ctx_reserve(nfx_ctx *ctx, const int *sets, int n_sets, int64_t n, int64_t nv, int ps, int batch_size)
{
    nfx_status st = check_common(ctx, n, nv, ps);
    if (st != NFX_OK) return st;
    if (!sets || n_sets <= 0 || sets_total_columns(sets, n_sets) < 0) return fail(ctx, NFX_ERR_INVALID_ARGUMENT, "unknown feature set");
    if (batch_size < 1) return fail(ctx, NFX_ERR_INVALID_ARGUMENT, "batch_size must be >= 1");
    NFX_HIP(ctx, hipSetDevice(ctx->device));
    if (stream_capturing(ctx)) return fail(ctx, NFX_ERR_INVALID_ARGUMENT, "nfx_ctx_reserve during stream capture");
    // the largest of: the fused entries (tile variant: gathered patches + centroids), one trait-entry call per listed set,
    // the rasteriser's private row words
    size_t need = extract_ws_bytes(sets, n_sets, n, nv, ps, batch_size, true, any_pixel_set(sets, n_sets));
    for (int s = 0; s < n_sets; ++s) {
        const size_t t = trait_ws_bytes(sets[s], n, nv, ps);
        need = t > need ? t : need;
    }
    const bool was = ctx->reserved;
    ctx->reserved = false; // the builders below are the one place allowed to allocate
    ctx->ws = &ctx->dev_ws;
    st = prepare_sets(ctx, sets, n_sets);
    // growing an existing workspace moves it: graphs captured earlier hold the old addresses (nfx.h, nfx_ctx_workspace_generation)
    if (st == NFX_OK && need > ctx->dev_ws.cap) st = arena_grow(ctx, ctx->dev_ws, need);
    ctx->reserved = was || st == NFX_OK;
    return st;
}

nfx_status nfx_preprocess_polygons(nfx_ctx *ctx, int64_t n, int64_t nv, const uint32_t *ring_offsets, const float *ring_xy,
                                   float *centroids_xy, float *centred_xy, nfx_mem mem)
{
    nfx_status st = check_common(ctx, n, nv, 1);
    if (st != NFX_OK) return st;
    if (!ring_offsets || (nv && !ring_xy)) return fail(ctx, NFX_ERR_INVALID_ARGUMENT, "null input");
    NFX_HIP(ctx, hipSetDevice(ctx->device));
    StreamScope stream_scope(ctx, mem);
    if (mem == NFX_MEM_DEVICE) {
        NFX_HIP(ctx, launch_preprocess(ctx, n, ring_offsets, ring_xy, centroids_xy, centred_xy));
        return NFX_OK;
    }
    if ((st = check_offsets(ctx, ring_offsets, n, nv, "ring_offsets")) != NFX_OK) return st;
    if ((st = ws_reserve(ctx, al(4 * (n + 1)) + 2 * al(8 * nv + 8) + al(8 * n + 8) + 4096, mem)) != NFX_OK) return st;
    uint32_t *d_off = stage_in(ctx, ring_offsets, (size_t)n + 1);
    float *d_xy = stage_in(ctx, ring_xy, 2 * (size_t)nv);
    float *d_cen = (float *)ws_alloc(ctx, 8 * (size_t)n + 8);
    float *d_ctr = (float *)ws_alloc(ctx, 8 * (size_t)nv + 8);
    if (!d_off || !d_xy || !d_cen || !d_ctr) return fail(ctx, NFX_ERR_OUT_OF_MEMORY, "workspace");
    NFX_HIP(ctx, launch_preprocess(ctx, n, d_off, d_xy, d_cen, d_ctr));
    if (centroids_xy && n) NFX_HIP(ctx, hipMemcpyAsync(centroids_xy, d_cen, 8 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    if (centred_xy && nv) NFX_HIP(ctx, hipMemcpyAsync(centred_xy, d_ctr, 8 * (size_t)nv, hipMemcpyDeviceToHost, ctx->stream));
    NFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NFX_OK;
}

nfx_status nfx_rasterize_polygons(nfx_ctx *ctx, int64_t n, int64_t nv, const uint32_t *poly_offsets, const float *poly_xy, int ps,
                                  float *masks_f32, uint64_t *mask_rows, nfx_mem mem)
{
    nfx_status st = check_common(ctx, n, nv, ps);
    if (st != NFX_OK) return st;
    if (!poly_offsets || (nv && !poly_xy)) return fail(ctx, NFX_ERR_INVALID_ARGUMENT, "null input");
    NFX_HIP(ctx, hipSetDevice(ctx->device));
    StreamScope stream_scope(ctx, mem);
    const size_t rows_b = sizeof(uint64_t) * (size_t)n * ps * mask_words_per_row(ps), mask_b = sizeof(float) * (size_t)n * ps * ps;
    if (mem == NFX_MEM_DEVICE) {
        uint64_t *rows = mask_rows;
        if (!rows) {
            if ((st = ws_reserve(ctx, al(rows_b + 8) + 4096, mem)) != NFX_OK) return st;
            rows = (uint64_t *)ws_alloc(ctx, rows_b + 8);
            if (!rows) return fail(ctx, NFX_ERR_OUT_OF_MEMORY, "workspace");
        }
        NFX_HIP(ctx, launch_raster_shape(ctx, n, poly_offsets, poly_xy, ps, rows, nullptr, 0, 0));
        if (masks_f32) NFX_HIP(ctx, launch_unpack_masks(ctx, n, rows, ps, masks_f32));
        return NFX_OK;
    }
    if ((st = check_offsets(ctx, poly_offsets, n, nv, "poly_offsets")) != NFX_OK) return st;
    if ((st = ws_reserve(ctx, al(4 * (n + 1)) + al(8 * nv + 8) + al(rows_b + 8) + al(mask_b + 8) + 4096, mem)) != NFX_OK) return st;
    uint32_t *d_off = stage_in(ctx, poly_offsets, (size_t)n + 1);
    float *d_xy = stage_in(ctx, poly_xy, 2 * (size_t)nv);
    uint64_t *d_rows = (uint64_t *)ws_alloc(ctx, rows_b + 8);
    float *d_m = masks_f32 ? (float *)ws_alloc(ctx, mask_b + 8) : nullptr;
    if (!d_off || !d_xy || !d_rows || (masks_f32 && !d_m)) return fail(ctx, NFX_ERR_OUT_OF_MEMORY, "workspace");
    NFX_HIP(ctx, launch_raster_shape(ctx, n, d_off, d_xy, ps, d_rows, nullptr, 0, 0));
    if (masks_f32) {
        NFX_HIP(ctx, launch_unpack_masks(ctx, n, d_rows, ps, d_m));
        if (n) NFX_HIP(ctx, hipMemcpyAsync(masks_f32, d_m, mask_b, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (mask_rows && n) NFX_HIP(ctx, hipMemcpyAsync(mask_rows, d_rows, rows_b, hipMemcpyDeviceToHost, ctx->stream));
    NFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NFX_OK;
}

nfx_status nfx_polygon_features_f64(nfx_ctx *ctx, int64_t n, int64_t nv, const uint32_t *poly_offsets, const float *poly_xy,
                                    double *out_f64, nfx_mem mem)
{
    nfx_status st = check_common(ctx, n, nv, 1);
    if (st != NFX_OK) return st;
    if (!poly_offsets || (nv && !poly_xy) || !out_f64) return fail(ctx, NFX_ERR_INVALID_ARGUMENT, "null input");
    NFX_HIP(ctx, hipSetDevice(ctx->device));
    StreamScope stream_scope(ctx, mem);
    size_t need = al(polygon_scratch_bytes(n, nv)) + 4096;
    if (mem == NFX_MEM_HOST) {
        if ((st = check_offsets(ctx, poly_offsets, n, nv, "poly_offsets")) != NFX_OK) return st;
        need += al(4 * (n + 1)) + al(8 * nv + 8) + al(56 * (size_t)n + 8);
    }
    if ((st = ws_reserve(ctx, need, mem)) != NFX_OK) return st;
    double *d_out = out_f64;
    if (mem == NFX_MEM_HOST) {
        poly_offsets = stage_in(ctx, poly_offsets, (size_t)n + 1);
        poly_xy = stage_in(ctx, poly_xy, 2 * (size_t)nv);
        d_out = (double *)ws_alloc(ctx, 56 * (size_t)n + 8);
        if (!poly_offsets || !poly_xy || !d_out) return fail(ctx, NFX_ERR_OUT_OF_MEMORY, "workspace");
    }
    void *scr = ws_alloc(ctx, polygon_scratch_bytes(n, nv));
    if (!scr) return fail(ctx, NFX_ERR_OUT_OF_MEMORY, "workspace");
    NFX_HIP(ctx, launch_polygon(ctx, n, poly_offsets, poly_xy, scr, nullptr, 0, 0, d_out));
    if (mem == NFX_MEM_HOST) {
        if (n) NFX_HIP(ctx, hipMemcpyAsync(out_f64, d_out, 56 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        NFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return NFX_OK;
}

nfx_status nfx_compute_features_batched(nfx_ctx *ctx, int set, int64_t n, int64_t nv, const float *centroids_xy,
                                        const uint32_t *poly_offsets, const float *poly_xy, const float *patchs, const float *masks,
                                        int ps, float *out, nfx_mem mem)
{
    (void)centroids_xy;
    nfx_status st = check_common(ctx, n, nv, ps);
    if (st != NFX_OK) return st;
    if (set < 0 || set >= NFX_SET_COUNT) return fail(ctx, NFX_ERR_INVALID_ARGUMENT, "unknown feature set");
    if (!poly_offsets || (nv && !poly_xy) || !masks || !out || (set != NFX_SET_GEOMETRY && !patchs))
        return fail(ctx, NFX_ERR_INVALID_ARGUMENT, "null input");
    if (n > 0x7FFFFFFF) return fail(ctx, NFX_ERR_UNSUPPORTED, "batch too large");
    NFX_HIP(ctx, hipSetDevice(ctx->device));
    StreamScope stream_scope(ctx, mem);
    const int F = kNumCols[set];
    const size_t P = (size_t)ps * ps;
    Job job;
    job.n = n; job.ps = ps; job.batch_size = (int)(n > 0 ? n : 1); job.patches_f32 = true; job.F = F;
    size_t need = trait_ws_bytes(set, n, nv, ps);
    if (mem == NFX_MEM_HOST) {
        if ((st = check_offsets(ctx, poly_offsets, n, nv, "poly_offsets")) != NFX_OK) return st;
        need += al(4 * (n + 1)) + al(8 * nv + 8) + al(4 * n * P + 8) + al(12 * n * P + 8) + al(4 * (size_t)n * F + 8);
    }
    if ((st = ws_reserve(ctx, need, mem)) != NFX_OK) return st;
    if ((st = prepare_sets(ctx, &set, 1)) != NFX_OK)
        return fail(ctx, st, "device tables of this set are not built: call nfx_ctx_reserve first (" + ctx->err + ")");
    float *d_out = out;
    if (mem == NFX_MEM_HOST) {
        poly_offsets = stage_in(ctx, poly_offsets, (size_t)n + 1);
        poly_xy = stage_in(ctx, poly_xy, 2 * (size_t)nv);
        masks = stage_in(ctx, masks, (size_t)n * P);
        if (set != NFX_SET_GEOMETRY) patchs = stage_in(ctx, patchs, 3 * (size_t)n * P);
        d_out = (float *)ws_alloc(ctx, 4 * (size_t)n * F + 4);
        if (!poly_offsets || !poly_xy || !masks || (set != NFX_SET_GEOMETRY && !patchs) || !d_out)
            return fail(ctx, NFX_ERR_OUT_OF_MEMORY, "workspace");
    }
    uint64_t *rows = (uint64_t *)ws_alloc(ctx, 8 * (size_t)n * ps * mask_words_per_row(ps) + 8);
    if (!rows) return fail(ctx, NFX_ERR_OUT_OF_MEMORY, "workspace");
    NFX_HIP(ctx, launch_pack_masks(ctx, n, masks, ps, rows));
    job.poly_off = poly_offsets; job.poly_xy = poly_xy; job.patches = patchs; job.mask_rows = rows; job.out = d_out;
    switch (set) {
    case NFX_SET_GEOMETRY: {
        void *scr = ws_alloc(ctx, polygon_scratch_bytes(n, nv));
        if (!scr) return fail(ctx, NFX_ERR_OUT_OF_MEMORY, "workspace");
        NFX_HIP(ctx, launch_shape_from_rows(ctx, n, rows, ps, d_out, F, 0));
        NFX_HIP(ctx, launch_polygon(ctx, n, poly_offsets, poly_xy, scr, d_out, F, 0));
        break;
    }
    case NFX_SET_COLOR: {
        void *scr = ws_alloc(ctx, color_scratch_bytes(job));
        if (!scr) return fail(ctx, NFX_ERR_OUT_OF_MEMORY, "workspace");
        NFX_HIP(ctx, launch_color(ctx, job, 0, scr));
        break;
    }
    case NFX_SET_GLCM: {
        void *scr = ws_alloc(ctx, glcm_scratch_bytes(job));
        if (!scr) return fail(ctx, NFX_ERR_OUT_OF_MEMORY, "workspace");
        NFX_HIP(ctx, launch_glcm(ctx, job, 0, scr));
        break;
    }
    case NFX_SET_GLRLM: {
        void *scr = glrlm_scratch_bytes(job) ? ws_alloc(ctx, glrlm_scratch_bytes(job)) : nullptr;
        if (!scr && glrlm_scratch_bytes(job)) return fail(ctx, NFX_ERR_OUT_OF_MEMORY, "workspace");
        NFX_HIP(ctx, launch_glrlm(ctx, job, 0, scr));
        break;
    }
    case NFX_SET_GABOR: {
        void *scr = ws_alloc(ctx, gabor_scratch_bytes(job));
        if (!scr && gabor_scratch_bytes(job)) return fail(ctx, NFX_ERR_OUT_OF_MEMORY, "workspace");
        NFX_HIP(ctx, launch_gabor(ctx, job, 0, scr));
        break;
    }
    }
    if (mem == NFX_MEM_HOST) {
        if (n) NFX_HIP(ctx, hipMemcpyAsync(out, d_out, 4 * (size_t)n * F, hipMemcpyDeviceToHost, ctx->stream));
        NFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return NFX_OK;
}

static nfx_status extract_common(nfx_ctx *ctx, const int *sets, int n_sets, int64_t n, int64_t nv, const uint32_t *ring_offsets,
                                 const float *ring_xy, const uint8_t *patches_u8, const uint8_t *image, int64_t H, int64_t W, int ps,
                                 int batch_size, float *out_rows, float *out_centroids, nfx_mem mem)
{
    nfx_status st = check_common(ctx, n, nv, ps);
    if (st != NFX_OK) return st;
    if (!sets || n_sets <= 0) return fail(ctx, NFX_ERR_INVALID_ARGUMENT, "no feature set selected");
    const int F = sets_total_columns(sets, n_sets);
    if (F < 0) return fail(ctx, NFX_ERR_INVALID_ARGUMENT, "unknown feature set");
    if (batch_size < 1) return fail(ctx, NFX_ERR_INVALID_ARGUMENT, "batch_size must be >= 1");
    const bool pix = any_pixel_set(sets, n_sets);
    if (!ring_offsets || (nv && !ring_xy) || !out_rows || (pix && !patches_u8 && !image))
        return fail(ctx, NFX_ERR_INVALID_ARGUMENT, "null input");
    if (image && (H < 1 || W < 1)) return fail(ctx, NFX_ERR_INVALID_ARGUMENT, "empty image");
    NFX_HIP(ctx, hipSetDevice(ctx->device));
    StreamScope stream_scope(ctx, mem);
    const size_t P = (size_t)ps * ps;
    size_t need = extract_ws_bytes(sets, n_sets, n, nv, ps, batch_size, image && !out_centroids, image && pix);
    if (mem == NFX_MEM_HOST) {
        if ((st = check_offsets(ctx, ring_offsets, n, nv, "ring_offsets")) != NFX_OK) return st;
        need += al(4 * (n + 1)) + al(8 * nv + 8) + al(4 * (size_t)n * F + 8) + al(8 * n + 8);
        if (pix) need += image ? al(3 * (size_t)H * W) : al(3 * (size_t)n * P + 8);
    }
    if ((st = ws_reserve(ctx, need, mem)) != NFX_OK) return st;
    if (mem == NFX_MEM_DEVICE)
        return extract_device(ctx, sets, n_sets, n, nv, ring_offsets, ring_xy, patches_u8, image, H, W, ps, batch_size, out_rows, out_centroids);

    uint32_t *d_off = stage_in(ctx, ring_offsets, (size_t)n + 1);
    float *d_xy = stage_in(ctx, ring_xy, 2 * (size_t)nv);
    const uint8_t *d_p = nullptr, *d_img = nullptr;
    if (pix && image) d_img = stage_in(ctx, image, 3 * (size_t)H * W);
    else if (pix) d_p = stage_in(ctx, patches_u8, 3 * (size_t)n * P);
    float *d_out = (float *)ws_alloc(ctx, 4 * (size_t)n * F + 4);
    float *d_cen = (float *)ws_alloc(ctx, 8 * (size_t)n + 8);
    if (!d_off || !d_xy || (pix && !d_p && !d_img) || !d_out || !d_cen) return fail(ctx, NFX_ERR_OUT_OF_MEMORY, "workspace");
    st = extract_device(ctx, sets, n_sets, n, nv, d_off, d_xy, d_p, d_img, H, W, ps, batch_size, d_out, d_cen);
    if (st != NFX_OK) { (void)hipStreamSynchronize(ctx->stream); return st; }
    if (n) NFX_HIP(ctx, hipMemcpyAsync(out_rows, d_out, 4 * (size_t)n * F, hipMemcpyDeviceToHost, ctx->stream));
    if (out_centroids && n) NFX_HIP(ctx, hipMemcpyAsync(out_centroids, d_cen, 8 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    NFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NFX_OK;
}

nfx_status nfx_extract(nfx_ctx *ctx, const int *sets, int n_sets, int64_t n, int64_t nv, const uint32_t *ring_offsets,
                       const float *ring_xy, const uint8_t *patches_u8, int ps, int batch_size, float *out_rows,
                       float *out_centroids_xy, nfx_mem mem)
{
    return extract_common(ctx, sets, n_sets, n, nv, ring_offsets, ring_xy, patches_u8, nullptr, 0, 0, ps, batch_size, out_rows,
                          out_centroids_xy, mem);
}

nfx_status nfx_extract_tile(nfx_ctx *ctx, const int *sets, int n_sets, int64_t n, int64_t nv, const uint32_t *ring_offsets,
                            const float *ring_xy, const uint8_t *image_u8, int64_t H, int64_t W, int ps, int batch_size,
                            float *out_rows, float *out_centroids_xy, nfx_mem mem)
{
    if (!image_u8) return fail(ctx, NFX_ERR_INVALID_ARGUMENT, "null image");
    return extract_common(ctx, sets, n_sets, n, nv, ring_offsets, ring_xy, nullptr, image_u8, H, W, ps, batch_size, out_rows,
                          out_centroids_xy, mem);
}

nfx_status nfx_gather_patches(nfx_ctx *ctx, int64_t n, const float *centroids_xy, const uint8_t *image_u8, int64_t H, int64_t W, int ps,
                              uint8_t *patches_u8, nfx_mem mem)
{
    nfx_status st = check_common(ctx, n, 0, ps);
    if (st != NFX_OK) return st;
    if (!centroids_xy || !image_u8 || !patches_u8 || H < 1 || W < 1) return fail(ctx, NFX_ERR_INVALID_ARGUMENT, "null input");
    NFX_HIP(ctx, hipSetDevice(ctx->device));
    StreamScope stream_scope(ctx, mem);
    if (mem == NFX_MEM_DEVICE) {
        NFX_HIP(ctx, launch_gather_patches(ctx, n, centroids_xy, image_u8, H, W, ps, patches_u8));
        return NFX_OK;
    }
    const size_t pb = (size_t)n * 3 * ps * ps;
    if ((st = ws_reserve(ctx, al(8 * n + 8) + al(3 * (size_t)H * W) + al(pb + 8) + 4096, mem)) != NFX_OK) return st;
    float *d_c = stage_in(ctx, centroids_xy, 2 * (size_t)n);
    uint8_t *d_i = stage_in(ctx, image_u8, 3 * (size_t)H * W);
    uint8_t *d_p = (uint8_t *)ws_alloc(ctx, pb + 8);
    if (!d_c || !d_i || !d_p) return fail(ctx, NFX_ERR_OUT_OF_MEMORY, "workspace");
    NFX_HIP(ctx, launch_gather_patches(ctx, n, d_c, d_i, H, W, ps, d_p));
    if (n) NFX_HIP(ctx, hipMemcpyAsync(patches_u8, d_p, pb, hipMemcpyDeviceToHost, ctx->stream));
    NFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NFX_OK;
}

// test hook: the integers behind the GLCM / GLRLM columns (nfx.h)
nfx_status nfx_debug_texture_counts(nfx_ctx *ctx, int64_t n, int64_t nv, const uint32_t *ring_offsets, const float *ring_xy,
                                    const uint8_t *patches_u8, int ps, int64_t *glcm_sums, uint32_t *glrlm_bins, nfx_mem mem)
{
    nfx_status st = check_common(ctx, n, nv, ps);
    if (st != NFX_OK) return st;
    if (!ring_offsets || (nv && !ring_xy) || !patches_u8) return fail(ctx, NFX_ERR_INVALID_ARGUMENT, "null input");
    NFX_HIP(ctx, hipSetDevice(ctx->device));
    StreamScope stream_scope(ctx, mem);
    const size_t P = (size_t)ps * ps;
    const size_t gb = sizeof(int64_t) * 96 * (size_t)n, rb = sizeof(uint32_t) * 4 * 384 * (size_t)n, fb = sizeof(float) * 224 * (size_t)n;
    Job job;
    job.n = n; job.ps = ps; job.batch_size = 100;
    size_t need = al(8 * nv + 8) + al(8 * (size_t)n * ps * mask_words_per_row(ps) + 8) + al(glcm_scratch_bytes(job)) + al(glrlm_scratch_bytes(job)) + al(fb + 8) + 4096;
    if (mem == NFX_MEM_HOST) {
        if ((st = check_offsets(ctx, ring_offsets, n, nv, "ring_offsets")) != NFX_OK) return st;
        need += al(4 * (n + 1)) + al(8 * nv + 8) + al(3 * n * P + 8) + al(gb + 8) + al(rb + 8);
    }
    if ((st = ws_reserve(ctx, need, mem)) != NFX_OK) return st;
    int64_t *d_g = glcm_sums;
    uint32_t *d_r = glrlm_bins;
    if (mem == NFX_MEM_HOST) {
        ring_offsets = stage_in(ctx, ring_offsets, (size_t)n + 1);
        ring_xy = stage_in(ctx, ring_xy, 2 * (size_t)nv);
        patches_u8 = stage_in(ctx, patches_u8, 3 * (size_t)n * P);
        if (glcm_sums) d_g = (int64_t *)ws_alloc(ctx, gb + 8);
        if (glrlm_bins) d_r = (uint32_t *)ws_alloc(ctx, rb + 8);
        if (!ring_offsets || !ring_xy || !patches_u8 || (glcm_sums && !d_g) || (glrlm_bins && !d_r)) return fail(ctx, NFX_ERR_OUT_OF_MEMORY, "workspace");
    }
    float *centred = (float *)ws_alloc(ctx, 8 * (size_t)nv + 8);
    uint64_t *rows = (uint64_t *)ws_alloc(ctx, 8 * (size_t)n * ps * mask_words_per_row(ps) + 8);
    float *feat = (float *)ws_alloc(ctx, fb + 8); // the feature columns themselves are computed and dropped
    void *scr = ws_alloc(ctx, glcm_scratch_bytes(job));
    void *rl_scr = glrlm_scratch_bytes(job) ? ws_alloc(ctx, glrlm_scratch_bytes(job)) : nullptr;
    if (!centred || !rows || !feat || !scr || (!rl_scr && glrlm_scratch_bytes(job))) return fail(ctx, NFX_ERR_OUT_OF_MEMORY, "workspace");
    const int glcm_set = NFX_SET_GLCM;
    if ((st = prepare_sets(ctx, &glcm_set, 1)) != NFX_OK) return st;
    NFX_HIP(ctx, launch_preprocess(ctx, n, ring_offsets, ring_xy, nullptr, centred));
    NFX_HIP(ctx, launch_raster_shape(ctx, n, ring_offsets, centred, ps, rows, nullptr, 0, 0));
    job.poly_off = ring_offsets; job.poly_xy = centred; job.patches = patches_u8; job.patches_f32 = false; job.mask_rows = rows;
    job.out = feat; job.F = 224; job.dbg_glcm = d_g; job.dbg_glrlm = d_r;
    if (d_g) NFX_HIP(ctx, launch_glcm(ctx, job, 0, scr));
    if (d_r) NFX_HIP(ctx, launch_glrlm(ctx, job, 0, rl_scr));
    if (mem == NFX_MEM_HOST) {
        if (glcm_sums && n) NFX_HIP(ctx, hipMemcpyAsync(glcm_sums, d_g, gb, hipMemcpyDeviceToHost, ctx->stream));
        if (glrlm_bins && n) NFX_HIP(ctx, hipMemcpyAsync(glrlm_bins, d_r, rb, hipMemcpyDeviceToHost, ctx->stream));
        NFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return NFX_OK;
}

}  // extern "C"
