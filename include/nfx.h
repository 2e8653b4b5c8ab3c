/*
 * nfx.h -- C ABI of the MI355X-native per-nucleus feature-extraction engine.
 *
 * Drop-in boundary for the hot path of oxabz/nuclei-feature-extraction (reference paths below
 * are file:line in that repository).  Plain pointers and sizes only: no tensor types, no
 * allocation crosses the ABI, the caller owns every buffer.
 *
 * What each entry point replaces
 *   nfx_feature_set_* / nfx_parse_feature_sets   args.rs:18-73 (names, `texture`/`all` expansion,
 *                                                 registry order) and the `name()` strings of
 *                                                 shape.rs:132-134, color.rs:104-106,
 *                                                 texture.rs:169-171,312-314,371-373; column names
 *                                                 of shape.rs:113-128, color.rs:80-99,
 *                                                 texture.rs:81-157,244-301,346-361
 *   nfx_compute_features_batched                  trait FeatureSet::compute_features_batched,
 *                                                 features/mod.rs:12-28 (one set, one batch, f32
 *                                                 patchs [N,3,ps,ps] + f32 masks [N,1,ps,ps])
 *   nfx_preprocess_polygons                       preprocess_polygon, utils.rs:54-74
 *   nfx_rasterize_polygons                        tch_utils::shapes::polygon call sites,
 *                                                 utils.rs:90-95,152-157
 *   nfx_extract                                   the body of the rayon pipeline main.rs:146-151 on
 *                                                 packed u8 patches: preprocess + rasterise +
 *                                                 u8/255 (utils.rs:172) + every selected set in
 *                                                 CLI order + column concat (main.rs:76-89)
 *   nfx_extract_tile                              same, patches gathered from a resident [3,H,W]
 *                                                 u8 image by load_image_dataset's addressing rules
 *                                                 (utils.rs:141-206)
 *   nfx_extract_tile_streamed_v2                  the same pipeline stage on a slide that is NOT resident: the image
 *                                                 travels band by band over the host link while earlier bands'
 *                                                 nuclei are extracted (BASELINE configs[4]); replaces the crop
 *                                                 utils.rs:141-206 + move_tensors_to_device utils.rs:211-224 per batch
 *   nfx_centroid_key                              centroid_to_key_string, utils.rs:226-228
 *   nfx_comm_* / nfx_gather_rows                  the host-side row merge append_features (vstack_mut under a
 *                                                 mutex), main.rs:96-108, and the per-worker device choice
 *                                                 utils.rs:211-224: one process per GPU, one gather of the
 *                                                 rows to the root over RCCL / xGMI
 *
 * Errors: every function that can fail returns an nfx_status; nfx_last_error(ctx) gives the
 * text.  Degenerate nuclei do not fail: they yield NaN exactly where the reference's arithmetic
 * does (empty mask -> 0/0, shape.rs:142-184).
 *
 * Threading: one context per device (or per host thread); calls on distinct contexts are
 * concurrent and re-entrant (the reference invokes its feature sets from every rayon worker,
 * main.rs:146-151).  A context is not internally locked.
 *
 * Memory spaces: with NFX_MEM_HOST all data pointers are host memory and the call is
 * synchronous.  With NFX_MEM_DEVICE all data pointers are device memory of the context's GPU,
 * the work is enqueued on the context's stream (nfx_ctx_set_stream) and the call returns
 * without synchronising (stream-ordered: safe to chain, overlap H2D on another stream).
 */
#ifndef NFX_H
#define NFX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 0.5.0.  Against 0.4.0: (i) the streamed entry is nfx_extract_tile_streamed_v2 -- the un-suffixed symbol of 0.3.0 / 0.4.0 is still
   exported and returns NFX_ERR_UNSUPPORTED, so a binary built against an older header (whose nfx_stream_stats had no leading
   `struct_size`, or a shorter one) gets an error instead of memory written past its struct; (ii) nfx_ctx_set_option replaces the
   environment variables that used to steer the library (the default build reads NO environment variable except NFX_RCCL_PATH, see
   nfx_comm_*); (iii) nfx_build_stamp ties a binary to the sources it was built from.  Everything else is unchanged.
   0.4.0: nfx_stream_stats starts with `struct_size` (set by the caller) and the library never writes past it. */
#define NFX_VERSION 500

typedef struct nfx_ctx nfx_ctx;

typedef enum {
    NFX_OK = 0,
    NFX_ERR_INVALID_ARGUMENT = 1,
    NFX_ERR_UNSUPPORTED = 2,   /* e.g. patch size > NFX_MAX_PATCH_SIZE */
    NFX_ERR_DEVICE = 3,        /* HIP runtime error, text in nfx_last_error */
    NFX_ERR_OUT_OF_MEMORY = 4
} nfx_status;

/* registry order == args.rs `all` expansion order (args.rs:38-44) */
typedef enum {
    NFX_SET_GEOMETRY = 0, /* ShapeFeatureSet        12 columns */
    NFX_SET_COLOR = 1,    /* ColorFeatureSet        18 columns */
    NFX_SET_GLCM = 2,     /* GlcmFeatureSet        224 columns */
    NFX_SET_GLRLM = 3,    /* GLRLMFeatureSet        68 columns */
    NFX_SET_GABOR = 4,    /* GaborFilterFeatureSet  96 columns */
    NFX_SET_COUNT = 5
} nfx_feature_set;

typedef enum { NFX_MEM_HOST = 0, NFX_MEM_DEVICE = 1 } nfx_mem;

/* Patch sizes (args.rs:93-94, default 64).  Up to 64 a mask row is one 64-bit word in one lane's registers and the
   wave-per-nucleus kernels run; 65..128 take a second kernel family (one workgroup per nucleus, two words per mask row; Gabor on
   64 x 64 tiles through the fast kernel): same results, lower throughput (all five sets: 0.5 M nuclei/s at 128 x 128 against 7 M/s
   at 64 x 64); 129..512 take a third, plainer family (the same workgroup-per-nucleus kernels with mask rows of up to eight words,
   level maps and dense count tables in device scratch memory instead of LDS, Gabor on up to 15 x 15 tiles): same results, a
   correctness path with no claim to speed.  Larger patches are refused with NFX_ERR_UNSUPPORTED (the reference's `-p` takes any
   u32, args.rs:93-94; its default is 64). */
#define NFX_MAX_PATCH_SIZE 512

/* ---- registry (no context, no GPU needed) ------------------------------------------------ */
int nfx_version(void);
/* sha-256 (first 16 hex digits) of the library's sources -- csrc's *.hip, *.cpp, *.h and this header, in name order -- as
   csrc/Makefile computed it when the library was built (scripts/csrc_stamp.py is that computation).  bench.py prints it as
   config.csrc_stamp and refuses to run when it differs from the hash of the sources on disk: a stale prebuilt library cannot carry a
   fresh stamp.  "unstamped" when the library was built without the Makefile.  The binary also holds the string
   "NFX_BUILD_STAMP=<stamp>", so that a tool can read the stamp out of the file without loading it (not an environment variable). */
const char *nfx_build_stamp(void);
int nfx_feature_set_count(void);
/* FeatureSet::name(): "geometry","color","GLCM","GLRLM","gabor filter"; NULL if out of range */
const char *nfx_feature_set_name(int set);
/* CLI spelling accepted by args.rs:21-34: geometry color glcm glrlm gabor */
const char *nfx_feature_set_cli_name(int set);
int nfx_feature_set_num_columns(int set); /* -1 if out of range */
/* writes the NUL-terminated name of column j (excluding the leading "centroid" key column);
   returns its length, or -1 on bad index / short buffer */
int nfx_feature_set_column_name(int set, int j, char *buf, size_t buflen);
/* args.rs FeatureSet::from_str + flat(): case-insensitive; "texture" -> glcm,glrlm,gabor;
   "all" -> geometry,color,glcm,glrlm,gabor; no de-duplication.  Returns the number of sets
   written to out_sets (<= cap), -1 if a name is not a feature set, -2 if cap is too small. */
int nfx_parse_feature_sets(const char *const *names, int n_names, int *out_sets, int cap);
/* "{x},{y}" with Rust's f32 Display (shortest round-trip, positional).  Returns length or -1. */
int nfx_centroid_key(float x, float y, char *buf, size_t buflen);
/* dtype of column j in the reference's frame: every tensor-derived column is f32 (`f32::from` / `Vec::<f32>::from`,
   shape.rs:138-224, color.rs:55-77, texture.rs:64-79); the seven polygon-derived geometry columns are pushed as
   geometric_features' own return values (shape.rs:89-97,116-127), f64 by inference from its (f64,f64) input points
   (utils.rs:39-42).  nfx_extract writes f32 rows; nfx_polygon_features_f64 gives those seven columns unrounded. */
typedef enum { NFX_DTYPE_F32 = 0, NFX_DTYPE_F64 = 1 } nfx_dtype;
int nfx_feature_set_column_dtype(int set, int j); /* -1 on bad index */

/* ---- context ----------------------------------------------------------------------------- */
nfx_status nfx_ctx_create(int device, nfx_ctx **out);
void nfx_ctx_destroy(nfx_ctx *ctx);
const char *nfx_last_error(const nfx_ctx *ctx);
/* hipStream_t the NFX_MEM_DEVICE calls are enqueued on (NULL = the default stream).  NFX_MEM_HOST calls are synchronous
   and run on a private non-blocking stream of the context, so contexts driven from different host threads (the reference's
   rayon workers, main.rs:146-151) overlap on the GPU; such a call first waits for the device calls still in flight on
   this context (the device tables are shared; staging and scratch live in an arena of their own). */
nfx_status nfx_ctx_set_stream(nfx_ctx *ctx, void *hip_stream);
/* How nfx_extract_tile / nfx_gather_patches place the ps x ps window around a centroid.  The reference has two loaders:
     NFX_ADDR_IMAGE  load_image_dataset (png/jpg), utils.rs:159-191: box [trunc(c-ps/2), trunc(c+ps/2)) as i64, clipped
                     to the image, pasted at offset (-min(top,0), -min(left,0)) of a zero patch.  Default.
     NFX_ADDR_SLIDE  load_slide_dataset (svs through OpenSlide), utils.rs:97-122: region address `(c-ps/2) as u32`
                     (negatives saturate to 0, so border nuclei are shifted, not padded), ps x ps from there, anything the
                     slide does not cover stays 0 at the bottom/right.  (A failed region read drops the nucleus in the
                     reference, utils.rs:125; a resident image cannot fail, so no nucleus is ever dropped here.) */
typedef enum { NFX_ADDR_IMAGE = 0, NFX_ADDR_SLIDE = 1 } nfx_addressing;
nfx_status nfx_ctx_set_patch_addressing(nfx_ctx *ctx, nfx_addressing mode);
/* blocks until everything enqueued by this context has finished */
nfx_status nfx_ctx_synchronize(nfx_ctx *ctx);
/* Behaviour switches of a context (set between calls; each takes effect from the next call on).  Unknown option or value:
   NFX_ERR_INVALID_ARGUMENT.
     NFX_OPT_STREAM_ROWS_STAGED  0 (default): when nfx_extract_tile_streamed_v2's out_rows is pinned the kernels write the rows
                                 straight into it over the link; 1: rows are always staged in a device buffer per chunk and copied home
                                 behind it (what a pageable out_rows gets anyway).  Same rows either way. */
typedef enum { NFX_OPT_STREAM_ROWS_STAGED = 0, NFX_OPT_COUNT = 1 } nfx_option;
nfx_status nfx_ctx_set_option(nfx_ctx *ctx, int option, int64_t value);
/* Sizes the context's device workspace and builds every device table the listed sets need, once, up front, for
   device-memory calls (nfx_extract, nfx_extract_tile, nfx_compute_features_batched, nfx_rasterize_polygons) of up to n nuclei,
   n_vertices ring vertices, patch size ps and reference batch size batch_size (repeat the call for the largest of several
   shapes: the reservation only grows).  Synchronises; call it when nothing is in flight on the context.
   After a reservation NFX_MEM_DEVICE calls never allocate, free or synchronise: a call that does not fit fails with
   NFX_ERR_OUT_OF_MEMORY instead of growing the workspace.  WITHOUT a reservation the first device-memory call of a larger
   shape grows the workspace itself (hipStreamSynchronize + hipFree + hipMalloc) and builds missing tables with a synchronous
   copy -- legal only outside stream capture; during a capture such a call fails with NFX_ERR_INVALID_ARGUMENT.

   Captured graphs and the workspace.  A HIP graph captured from a device-memory call holds addresses inside the workspace of
   the device-memory calls.  That workspace moves in exactly one situation once a capture has been seen on the context: a
   later nfx_ctx_reserve for a LARGER shape (it frees and reallocates; every graph captured before it must be captured again --
   nfx_ctx_workspace_generation tells).  Nothing else moves it: NFX_MEM_HOST calls stage and compute in an arena of their own,
   whatever their size, and build missing device tables (separate allocations) also on a reserved context; an unreserved
   device-memory call that would outgrow the workspace after a capture fails with NFX_ERR_OUT_OF_MEMORY instead of growing. */
nfx_status nfx_ctx_reserve(nfx_ctx *ctx, const int *sets, int n_sets, int64_t n, int64_t n_vertices, int ps, int batch_size);
/* Number of times the workspace of the device-memory calls has been reallocated.  A graph captured while this read g is valid
   for as long as it still reads g. */
uint64_t nfx_ctx_workspace_generation(const nfx_ctx *ctx);

/* Per-kernel device timing with HIP events on the context's stream (off by default).
   nfx_ctx_kernel_timing synchronises, then reports for kernel slot k: name, number of launches
   and total milliseconds since the last reset.  Returns the number of slots. */
nfx_status nfx_ctx_enable_timing(nfx_ctx *ctx, int on);
int nfx_ctx_kernel_timing(nfx_ctx *ctx, int k, const char **name, int64_t *launches, double *total_ms);
nfx_status nfx_ctx_reset_timing(nfx_ctx *ctx);

/* ---- R1: utils.rs:54-74 -------------------------------------------------------------------
   ring_offsets [n+1] (vertex index of each nucleus' outer ring, ring_offsets[0] == 0),
   ring_xy [n_vertices][2] slide-pixel coordinates, closing duplicate included as listed;
   n_vertices == ring_offsets[n] (passed by value so device-resident calls need no read-back).
   Outputs: centroids_xy [n][2], centred_xy same shape as ring_xy. */
nfx_status nfx_preprocess_polygons(nfx_ctx *ctx, int64_t n, int64_t n_vertices, const uint32_t *ring_offsets,
                                   const float *ring_xy, float *centroids_xy, float *centred_xy, nfx_mem mem);

/* ---- R2: tch_utils::shapes::polygon call sites utils.rs:90-95,152-157 ---------------------
   centred polygons in, masks out.  Either output may be NULL:
     masks_f32  [n][1][ps][ps] f32 in {0,1}  (the reference's tensor)
     mask_rows  [n][ps][W] u64, W = ceil(ps / 64): 1 for ps <= 64 (bit c of word r == pixel (r,c)), 2 for ps 65..128, up to 8 (bit c & 63 of word
                c >> 6)  (the engine's native form) */
nfx_status nfx_rasterize_polygons(nfx_ctx *ctx, int64_t n, int64_t n_vertices, const uint32_t *poly_offsets,
                                  const float *poly_xy, int ps, float *masks_f32, uint64_t *mask_rows, nfx_mem mem);

/* ---- trait FeatureSet::compute_features_batched, features/mod.rs:21-27 --------------------
   ONE feature set on ONE reference batch (hue statistics couple the nuclei of a batch,
   color.rs:50-51,144-155, so the caller's batch partition is part of the result).
     centroids_xy [n][2]           unused by the arithmetic (key column only); may be NULL
     poly_offsets [n+1], poly_xy   centred polygons (utils.rs:66-73)
     patchs [n][3][ps][ps] f32 in [0,1], masks [n][1][ps][ps] f32 in {0,1}
     out [n][nfx_feature_set_num_columns(set)] row-major f32 */
nfx_status nfx_compute_features_batched(nfx_ctx *ctx, int set, int64_t n, int64_t n_vertices,
                                        const float *centroids_xy, const uint32_t *poly_offsets, const float *poly_xy,
                                        const float *patchs, const float *masks, int ps, float *out, nfx_mem mem);

/* ---- pipeline body main.rs:146-151 on packed patches --------------------------------------
     sets[n_sets]                  feature sets in output order (as nfx_parse_feature_sets gives)
     ring_offsets/ring_xy          raw outer rings (as nfx_preprocess_polygons)
     patches_u8 [n][3][ps][ps]     planar RGB u8; the engine applies u8/255 (utils.rs:172)
     batch_size                    the reference's par_chunks size (args.rs:107-108, default 100)
     out_rows [n][F]               F = sum of the sets' column counts; rows in input order
     out_centroids_xy [n][2]       for the key column; may be NULL */
nfx_status nfx_extract(nfx_ctx *ctx, const int *sets, int n_sets, int64_t n, int64_t n_vertices,
                       const uint32_t *ring_offsets, const float *ring_xy, const uint8_t *patches_u8, int ps,
                       int batch_size, float *out_rows, float *out_centroids_xy, nfx_mem mem);

/* same, patches gathered from image_u8 [3][H][W] (planar, tch::vision::image::load layout,
   main.rs:26) by the rules of load_image_dataset, utils.rs:159-191 */
nfx_status nfx_extract_tile(nfx_ctx *ctx, const int *sets, int n_sets, int64_t n, int64_t n_vertices,
                            const uint32_t *ring_offsets, const float *ring_xy, const uint8_t *image_u8, int64_t H,
                            int64_t W, int ps, int batch_size, float *out_rows, float *out_centroids_xy, nfx_mem mem);

/* ---- the slide streamed band by band, host -> device copies overlapped with compute (BASELINE configs[4]) -------------------
   All pointers are HOST memory; the call is synchronous (rows and centroids are on the host when it returns) and runs on streams
   of the context's own.  image_u8 [3][H][W] is cut into horizontal bands of `block` rows (0 = 2048; at least 64) over the full
   width, each with a patch-size halo below -- a band's rows are contiguous in the planar image, so it travels as three large
   contiguous copies; nuclei are processed in index order in chunks of whole reference batches (`chunk_nuclei`, 0 = 100 000,
   rounded down to a multiple of batch_size; the workspace holds three chunks' patches and rows): while chunks k-1 and k are
   extracted (two at a time, on two streams: one's first kernels fill the other's tails), the bands chunk k+1 touches travel to the device (only
   those not resident: the device keeps `cache_bytes` of bands, 0 = the whole slide if 60 % of the free device memory (this
   context's own reusable staging arena counted as free) holds it,
   least recently used out first) and its nuclei's windows are gathered from them by load_image_dataset's rules (utils.rs:159-191;
   NFX_ADDR_SLIDE: utils.rs:97-122), and the rows of chunk k-2 travel back.  Rows are bit-identical to nfx_extract_tile on the
   resident image, whatever the order of the nuclei (slide order -- what segmentation tools emit -- makes a chunk touch one or
   two bands; any other order only costs copies).  `cache_bytes` is honoured down to ONE band (a smaller value keeps one): a batch
   whose windows start in more bands than the cache holds is gathered in several passes over its bands -- the gather needs one band
   at a time, only the extraction needs all of a batch's patches -- so no cache size makes a call fail, a small one only re-copies.  For the link's full rate and for copies that overlap compute, image_u8 and out_rows should be pinned
   for the HIP runtime this library is bound to (nfx_host_alloc, or nfx_host_register on memory of the caller's own).  A process holds ONE
   HIP runtime: /opt/rocm's when nothing else brought one, PyTorch's bundled one when torch was imported before this library was loaded --
   its pinned tensors are then pinned for this library too; the other load order maps two runtimes and leaves torch without a GPU
   (the Python binding refuses it).
   When out_rows is pinned the kernels write the rows straight into it over the link (nfx_ctx_set_option(NFX_OPT_STREAM_ROWS_STAGED, 1)
   keeps the staged form: a device buffer per chunk and a copy behind it).
   stats (may be NULL; set stats->struct_size = sizeof(nfx_stream_stats) before the call): what travelled ("blocks" = bands). */
typedef struct {
    int64_t struct_size;      /* IN: sizeof(nfx_stream_stats) as the caller was compiled; the library fills at most that many bytes
                                 (a value below 16, above 4096 or not a multiple of 8 is NFX_ERR_INVALID_ARGUMENT) and leaves it as it was */
    int64_t image_bytes_h2d;  /* block windows copied to the device (3 bytes per pixel, halos and re-copies included) */
    int64_t other_bytes_h2d;  /* ring offsets and vertices, per-chunk selection lists and rebased offsets */
    int64_t rows_bytes_d2h;   /* feature rows + centroids */
    int64_t blocks_copied, blocks_total, block_slots, chunks;
    int64_t image_pinned;     /* 1: the image is pinned for this runtime (asynchronous copies); 0: pageable */
    int64_t split_gather_passes; /* band passes made by chunks whose bands did not fit the cache at once (0: one pass per chunk) */
    int64_t rows_direct;      /* 1: out_rows is pinned and the feature kernels wrote the rows into it themselves (no staging buffer, no
                                 device -> host copy); 0: rows staged on the device and copied home chunk by chunk */
} nfx_stream_stats;
nfx_status nfx_extract_tile_streamed_v2(nfx_ctx *ctx, const int *sets, int n_sets, int64_t n, int64_t n_vertices,
                                        const uint32_t *ring_offsets, const float *ring_xy, const uint8_t *image_u8, int64_t H, int64_t W,
                                        int ps, int batch_size, int block, int64_t chunk_nuclei, int64_t cache_bytes, float *out_rows,
                                        float *out_centroids_xy, nfx_stream_stats *stats);
/* pinned host memory of this library's HIP runtime (hipHostMalloc / hipHostRegister): what makes host <-> device copies asynchronous
   and full rate.  nfx_host_alloc returns NULL when out of memory; nfx_host_register accepts memory that is registered already. */
void *nfx_host_alloc(size_t bytes);
void nfx_host_free(void *p);
nfx_status nfx_host_register(void *p, size_t bytes);
void nfx_host_unregister(void *p);

/* ---- geometric_features::{area, perimeter, equivalent_perimeter, compacity, convex_hull::convex_hull_features},
   call sites shape.rs:89-97 -- the polygon-derived geometry columns at the reference's own width:
     poly_offsets/poly_xy   centred polygons (utils.rs:66-73), as for nfx_compute_features_batched
     out_f64 [n][7]         area, perimeter, equivalent_perimeter, compacity, convex_hull_area, convex_deffect, convex_perimeter
   (the f32 rows of nfx_extract / nfx_compute_features_batched hold these same values rounded once to f32) */
nfx_status nfx_polygon_features_f64(nfx_ctx *ctx, int64_t n, int64_t n_vertices, const uint32_t *poly_offsets,
                                    const float *poly_xy, double *out_f64, nfx_mem mem);

/* utils.rs:159-191 alone: crops [n][3][ps][ps] u8 out of image_u8 at the given centroids */
nfx_status nfx_gather_patches(nfx_ctx *ctx, int64_t n, const float *centroids_xy, const uint8_t *image_u8, int64_t H,
                              int64_t W, int ps, uint8_t *patches_u8, nfx_mem mem);

/* ---- multi-GPU: the row merge main.rs:96-108 as ONE gather over RCCL / xGMI --------------------------------------------
   The reference merges rows on the host (vstack_mut under a mutex) from rayon workers that each picked `gpus[thread % n]`
   (utils.rs:211-224).  Here every process drives one GPU through its own context and extracts a contiguous index shard whose
   boundaries are multiples of the reference batch size (the hue mean couples the nuclei of a batch, color.rs:144-155); nothing is
   exchanged during extraction.  nfx_gather_rows then moves every shard's [count, F] f32 rows once, device to device, to the
   root: grouped ncclSend / ncclRecv (ragged shards need no padding), enqueued on the context's stream -- stream-ordered like
   every NFX_MEM_DEVICE call, no host synchronisation.  No PyTorch / libtorch is involved: librccl is opened (dlopen) at the first
   nfx_comm_* call; a process that never calls them never loads it.  Returns NFX_ERR_UNSUPPORTED if librccl is not there (the
   environment variable NFX_RCCL_PATH, when set, names the ONE library file that is tried instead of the usual sonames).

     nfx_comm_unique_id   one rank (by convention rank 0) makes the 128-byte id; the host distributes it out of band (file, socket,
                          environment -- ncclGetUniqueId semantics) and every rank passes the same bytes to nfx_comm_create
     nfx_comm_create      collective: returns once all `world` ranks have called it (ncclCommInitRank); the communicator lives on
                          ctx's device
     nfx_gather_rows      rows [counts[rank]][F] (device), counts [world] (HOST array, the same on every rank), dst
                          [sum counts][F] (device, root only; may be NULL elsewhere): rank r's rows land at row sum(counts[0..r)),
                          i.e. in nucleus-index order.  Collective: every rank of the communicator calls it with the same F,
                          counts and root. */
#define NFX_UNIQUE_ID_BYTES 128
typedef struct nfx_comm nfx_comm;
nfx_status nfx_comm_unique_id(void *id_out /* NFX_UNIQUE_ID_BYTES */);
nfx_status nfx_comm_create(nfx_ctx *ctx, int rank, int world, const void *unique_id, nfx_comm **out);
void nfx_comm_destroy(nfx_comm *comm);
int nfx_comm_rank(const nfx_comm *comm);
int nfx_comm_world(const nfx_comm *comm);
nfx_status nfx_gather_rows(nfx_ctx *ctx, nfx_comm *comm, const float *rows, int F, const int64_t *counts, float *dst, int root);

/* ---- TEST HOOK (not part of the reference's surface): the exact integers behind the texture columns, for count-level
   parity checks against tch_utils::glcm::glcm / glrlm::glrlm as called at texture.rs:40-46,193-194.  Inputs as nfx_extract.
   Either output may be NULL.
     glcm_sums  [n][4 levels: 32,64,128,254][4 offsets: texture.rs:20][6] int64 of the SYMMETRIC count matrix G (G[a][b] +=1 and
                G[b][a] += 1 per in-mask pair): pairs N = sum G / 2, sum (a+b) G / 2, sum (a+b)^2 G / 2, sum |a-b| G / 2,
                sum (a-b)^2 G / 2, sum G^2
     glrlm_bins [n][4 directions: texture.rs:176][24 levels][16 lengths] uint32 run counts */
nfx_status nfx_debug_texture_counts(nfx_ctx *ctx, int64_t n, int64_t n_vertices, const uint32_t *ring_offsets,
                                    const float *ring_xy, const uint8_t *patches_u8, int ps, int64_t *glcm_sums,
                                    uint32_t *glrlm_bins, nfx_mem mem);

#ifdef __cplusplus
}
#endif
#endif /* NFX_H */
