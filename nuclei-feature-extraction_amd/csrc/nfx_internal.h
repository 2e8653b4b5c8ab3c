// nfx_internal.h -- shared declarations of the engine (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/nfx.h"

namespace nfx {

// kernel timing slots (nfx_ctx_kernel_timing)
enum KernelSlot {
    KS_PREPROCESS = 0,
    KS_RASTER_SHAPE,
    KS_POLYGON,
    KS_PACK_MASKS,
    KS_GATHER,
    KS_HUE_BATCH,
    KS_COLOR,
    KS_GLCM,
    KS_GLRLM,
    KS_GABOR,
    KS_COUNT
};
extern const char *const kKernelSlotNames[KS_COUNT];

static constexpr int kNumCols[NFX_SET_COUNT] = {12, 18, 224, 68, 96};
// The fast kernels hold one mask row in one lane's 64-bit word; patches above this size take the "wide" family
// (one 256-thread workgroup per nucleus, mask rows of two words, lists and level maps in LDS) up to NFX_WIDE_PS, and above
// that the "huge" family (the same workgroup-per-nucleus shape, rows of up to NFX_MAX_PATCH_SIZE / 64 words, maps and count
// tables in global scratch memory) up to NFX_MAX_PATCH_SIZE
#define NFX_ROW_WORD_PS 64
#define NFX_WIDE_PS 128
#define NFX_MAX_ROW_WORDS (NFX_MAX_PATCH_SIZE / 64)
static inline int mask_words_per_row(int ps) { return (ps + 63) / 64; }

struct Timing {
    hipEvent_t start, stop;
    int slot;
};

struct Arena {
    char *base = nullptr;
    size_t cap = 0, off = 0;
};

}  // namespace nfx

struct nfx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;     // where NFX_MEM_DEVICE calls are enqueued (nfx_ctx_set_stream; NULL = the default stream)
    hipStream_t own_stream = nullptr; // non-blocking stream of the synchronous NFX_MEM_HOST calls: contexts driven from
                                      // different host threads overlap on the GPU instead of queueing on the default stream
    hipStream_t side_stream = nullptr; // the polygon kernel (64 serial programs per wave: latency-bound, hardly any issue slots) runs
    hipEvent_t ev_fork = nullptr, ev_join = nullptr; // here, beside the rasteriser and the pixel kernels of the same call
    // nfx_extract_tile_streamed_v2 (nfx_stream.cpp): block copies + window gathers | rows back to the host, created on first use
    hipStream_t copy_stream = nullptr, rows_stream = nullptr, d2h_stream = nullptr, alt_stream = nullptr;
    hipEvent_t ev_ready[3] = {}, ev_done[3] = {}, ev_rows[3] = {}, ev_bands[4] = {}, ev_pass = nullptr;
    hipEvent_t ev_front = nullptr, ev_front_up = nullptr, ev_front_cen = nullptr; // the centroid frontier: centroids home | rings up | centroids made
    void *pin = nullptr;      // grow-only pinned host staging (selection lists, rebased offsets of the chunks in flight)
    size_t pin_cap = 0;
    void *pin_cen = nullptr;  // grow-only pinned host staging: the centroids of a streamed call
    size_t pin_cen_cap = 0;
    nfx_addressing addressing = NFX_ADDR_IMAGE;
    bool opt_stream_rows_staged = false; // nfx_ctx_set_option(NFX_OPT_STREAM_ROWS_STAGED)
    std::string err;
    // Two grow-only device workspaces with a bump pointer each, rewound at the start of every API call:
    //   dev_ws   everything a NFX_MEM_DEVICE call needs.  A HIP graph captured from such a call holds addresses inside it, so it is
    //            reallocated only where nfx.h says so: by nfx_ctx_reserve (always) and by an unreserved call that outgrows it as long
    //            as no capture has been seen on the context.  `ws_generation` counts its reallocations.
    //   host_ws  staging and scratch of the synchronous NFX_MEM_HOST calls; grows freely, never seen by a captured graph.
    nfx::Arena dev_ws, host_ws;
    nfx::Arena *ws = &dev_ws;       // the arena of the API call in progress (StreamScope)
    nfx_mem call_mem = NFX_MEM_DEVICE; // memory kind of the API call in progress
    bool reserved = false;     // nfx_ctx_reserve was called: device-memory calls never grow the workspace or build tables
    bool capture_seen = false; // a device-memory call was enqueued into a stream capture: dev_ws must stay where it is
    uint64_t ws_generation = 0;
    // timing
    bool timing = false;
    std::vector<nfx::Timing> pending;
    std::vector<hipEvent_t> event_pool;
    int64_t launches[nfx::KS_COUNT] = {};
    double total_ms[nfx::KS_COUNT] = {};
    // Gabor filter tables (device), built lazily
    float *gabor_tab = nullptr;
    // u8 -> optical-density lookup (device), built lazily
    float *od_lut = nullptr;
    // GLCM rank-difference / x ln x tables (device), built lazily
    double *glcm_lut = nullptr;
};

namespace nfx {

// One unit of device work: nuclei [0,n) with everything resident on the device.
struct Job {
    int64_t n = 0;
    int ps = 0;
    int batch_size = 0;                 // reference batch (par_chunks) size; hue statistics couple within it
    const uint32_t *poly_off = nullptr; // [n+1] centred polygons
    const float *poly_xy = nullptr;
    const void *patches = nullptr;      // [n,3,ps,ps] u8 or f32
    bool patches_f32 = false;
    const uint64_t *mask_rows = nullptr; // [n,ps]
    const int *areas = nullptr;          // [n] mask pixel counts when the rasteriser produced them (else counted from the rows)
    float *out = nullptr;               // [n,F]
    int F = 0;
    // test hooks (nfx_debug_texture_counts); nullptr on every product path
    int64_t *dbg_glcm = nullptr;        // [n][4 levels][4 offsets][6]
    uint32_t *dbg_glrlm = nullptr;      // [n][4 directions][24][16]
};

// ---- shared by the API translation units (nfx_api.cpp, nfx_stream.cpp) -----------------------------------------------------
nfx_status api_fail(nfx_ctx *ctx, nfx_status st, const std::string &msg);
#define NFX_HIP(ctx, expr)                                                                                 \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) {                                                                            \
            (void)hipGetLastError(); /* reported here: the next call must not find it again */             \
            return nfx::api_fail(ctx, e_ == hipErrorOutOfMemory ? NFX_ERR_OUT_OF_MEMORY : NFX_ERR_DEVICE,  \
                                 std::string(#expr) + ": " + hipGetErrorString(e_));                       \
        }                                                                                                  \
    } while (0)
static inline size_t al(size_t b) { return (b + 255) & ~size_t(255); } // workspace allocations are 256-byte aligned
nfx_status check_common(nfx_ctx *ctx, int64_t n, int64_t nv, int ps);
nfx_status check_offsets(nfx_ctx *ctx, const uint32_t *off, int64_t n, int64_t nv, const char *what);
int sets_total_columns(const int *sets, int n_sets);
bool any_pixel_set(const int *sets, int n_sets);
int count_set(const int *sets, int n_sets, int which);
// workspace bytes of one extract_device call (excluding staging): mirrors its ws_alloc calls one for one
size_t extract_ws_bytes(const int *sets, int n_sets, int64_t n, int64_t nv, int ps, int batch_size, bool need_centroids, bool gather);
nfx_status prepare_sets(nfx_ctx *ctx, const int *sets, int n_sets);
// everything on the device, workspace reserved by the caller (bump-allocates from ctx->ws; joins its own side-stream fork)
nfx_status extract_device(nfx_ctx *ctx, const int *sets, int n_sets, int64_t n, int64_t nv, const uint32_t *ring_off, const float *ring_xy,
                          const uint8_t *patches_u8, const uint8_t *image, int64_t H, int64_t W, int ps, int batch_size, float *out_rows,
                          float *out_centroids);

// A synchronous (host-memory) call runs on the context's own stream and in the context's host arena (its staging buffers and
// scratch never touch the workspace of the device-memory calls, whose addresses a captured graph may hold).  It first waits
// for whatever the caller still has in flight on the device-call stream: both kinds of call share the device tables.
struct StreamScope {
    nfx_ctx *ctx;
    hipStream_t saved;
    StreamScope(nfx_ctx *c, nfx_mem mem) : ctx(c), saved(c->stream)
    {
        c->call_mem = mem;
        c->ws = mem == NFX_MEM_HOST ? &c->host_ws : &c->dev_ws;
        if (mem == NFX_MEM_HOST && c->own_stream) {
            (void)hipStreamSynchronize(saved);
            c->stream = c->own_stream;
        }
    }
    ~StreamScope()
    {
        ctx->stream = saved;
        ctx->ws = &ctx->dev_ws;
        ctx->call_mem = NFX_MEM_DEVICE;
    }
};

// RAII-less helpers ---------------------------------------------------------------------------
// make capacity >= bytes and rewind the bump pointer (only before ws_alloc calls).  Growing reallocates (synchronise +
// free + malloc): refused for device-memory calls on a reserved context or during stream capture
nfx_status ws_reserve(nfx_ctx *ctx, size_t bytes, nfx_mem mem);
bool stream_capturing(nfx_ctx *ctx);
void *ws_alloc(nfx_ctx *ctx, size_t bytes);        // 256-B aligned bump allocation; nullptr when exhausted
void timing_begin(nfx_ctx *ctx, int slot);
void timing_end(nfx_ctx *ctx);

struct ScopedTiming {
    nfx_ctx *c;
    ScopedTiming(nfx_ctx *ctx, int slot) : c(ctx) { timing_begin(c, slot); }
    ~ScopedTiming() { timing_end(c); }
};

// kernels (host launchers).  Every launcher enqueues on ctx->stream and returns hipGetLastError().
hipError_t launch_preprocess(nfx_ctx *ctx, int64_t n, const uint32_t *ring_off, const float *ring_xy, float *centroids,
                             float *centred);
// rasterise (+ the mask-derived geometry columns when out != nullptr).  rows_out may be nullptr.
hipError_t launch_raster_shape(nfx_ctx *ctx, int64_t n, const uint32_t *poly_off, const float *poly_xy, int ps,
                               uint64_t *rows_out, float *out, int F, int col0, int *areas_out = nullptr);
// geometry columns from existing mask rows (trait entry: masks are given, not rasterised)
hipError_t launch_shape_from_rows(nfx_ctx *ctx, int64_t n, const uint64_t *rows, int ps, float *out, int F, int col0);
// polygon-derived geometry columns; scratch: 3*total_vertices+2n+... doubles pairs (see k_geometry.hip)
// out (f32 rows, columns 0,5,6,7,9,10,11 of the geometry block) and out64 ([n][7] f64) may each be nullptr.
// beside_other_work: the launch sits on a side stream next to other kernels of the call (the issue-frugal serial kernel) rather
// than on the call's critical path (the group-parallel kernel)
hipError_t launch_polygon(nfx_ctx *ctx, int64_t n, const uint32_t *poly_off, const float *poly_xy, void *scratch,
                          float *out, int F, int col0, double *out64 = nullptr, bool beside_other_work = false);
size_t polygon_scratch_bytes(int64_t n, int64_t total_vertices);
hipError_t launch_pack_masks(nfx_ctx *ctx, int64_t n, const float *masks_f32, int ps, uint64_t *rows);
hipError_t launch_unpack_masks(nfx_ctx *ctx, int64_t n, const uint64_t *rows, int ps, float *masks_f32);
hipError_t launch_gather_patches(nfx_ctx *ctx, int64_t n, const float *centroids, const uint8_t *image, int64_t H,
                                 int64_t W, int ps, uint8_t *patches);
// the same out of a WINDOW of the image (rows [wy0, wy0+wh) x columns [wx0, wx0+ww), stored [3][wh][ww]) for the nuclei sel[0..n_sel)
// of the call (nullptr: nuclei 0..n_sel): patch of nucleus i lands at patches[i]
hipError_t launch_gather_patches_window(nfx_ctx *ctx, int64_t n_sel, const uint32_t *sel, const float *centroids, const uint8_t *window, int64_t H,
                                        int64_t W, int64_t wy0, int64_t wx0, int64_t wh, int64_t ww, int ps, uint8_t *patches);

hipError_t launch_color(nfx_ctx *ctx, const Job &job, int col0, void *scratch);
size_t color_scratch_bytes(const Job &job);
hipError_t launch_glcm(nfx_ctx *ctx, const Job &job, int col0, void *scratch);
size_t glcm_scratch_bytes(const Job &job);
hipError_t launch_glrlm(nfx_ctx *ctx, const Job &job, int col0, void *scratch);
size_t glrlm_scratch_bytes(const Job &job); // 0 up to NFX_WIDE_PS
hipError_t launch_gabor(nfx_ctx *ctx, const Job &job, int col0, void *scratch);
size_t gabor_scratch_bytes(const Job &job);
// device tables of a feature set (no-ops once built).  Building allocates and copies synchronously: nfx_ctx_reserve does it
// up front; a launcher that finds its table missing builds it on the spot unless tables_locked(ctx)
bool tables_locked(nfx_ctx *ctx); // a device-memory call on a reserved context, or a stream capture in progress: no allocation allowed
hipError_t prepare_color(nfx_ctx *ctx);
hipError_t prepare_glcm(nfx_ctx *ctx);
hipError_t prepare_gabor(nfx_ctx *ctx);

}  // namespace nfx
