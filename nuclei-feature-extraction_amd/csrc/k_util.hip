// k_util.hip -- polygon preprocessing, mask format conversion, patch gathering.
//
// Replaces (reference file:line):
//   preprocess_polygon                 utils.rs:54-74
//   load_image_dataset patch addressing utils.rs:159-191
#include "nfx_internal.h"

namespace nfx {

// One lane per nucleus for the centroid: it is a SEQUENTIAL f32 sum over the listed ring vertices (utils.rs:56-64), so
// the summation order is part of the result and is kept.  The centred vertices are then written by the whole wave: the
// 64 rings of a wave are one contiguous run of vertices, lane l takes vertices l, l + 64, ... of the run and finds the
// ring of each by a binary search over the wave's offsets (LDS) -- coalesced 8-byte loads and stores instead of 64
// separate streams.
__global__ __launch_bounds__(256) void k_preprocess(int64_t n, const uint32_t *__restrict__ ring_off,
                                                    const float *__restrict__ ring_xy, float *__restrict__ centroids,
                                                    float *__restrict__ centred)
{
    __shared__ uint32_t s_off[4][65];
    __shared__ float2 s_cen[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t i0 = ((int64_t)blockIdx.x * 4 + wv) * 64; // first nucleus of this wave
    if (i0 >= n) return;                                    // wave-uniform
    const int64_t i = i0 + lane;
    const int live = (int)((n - i0) < 64 ? (n - i0) : 64);  // nuclei of this wave
    const float2 *p = reinterpret_cast<const float2 *>(ring_xy);
    const uint32_t v0 = ring_off[i < n ? i : n], v1 = ring_off[i < n ? i + 1 : n];
    float ax = 0.0f, ay = 0.0f;
    for (uint32_t k = v0; k < v1; ++k) {
        const float2 v = p[k];
        ax += v.x;
        ay += v.y;
    }
    const float cnt = (float)(v1 - v0);
    const float cx = ax / cnt, cy = ay / cnt;
    if (centroids && i < n) {
        centroids[2 * i] = cx;
        centroids[2 * i + 1] = cy;
    }
    if (!centred) return;
    s_off[wv][lane] = v0;
    if (lane == 63) s_off[wv][64] = v1;
    s_cen[wv][lane] = make_float2(cx, cy);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t va = s_off[wv][0], vb = s_off[wv][live];
    float2 *q = reinterpret_cast<float2 *>(centred);
    for (uint32_t v = va + (uint32_t)lane; v < vb; v += 64u) {
        int lo = 0, hi = live - 1; // last ring whose first vertex is <= v (empty rings share their start with the next one
        while (lo < hi) {          // and are never chosen: the search keeps the LAST such ring)
            const int mid = (lo + hi + 1) >> 1;
            if (s_off[wv][mid] <= v) lo = mid; else hi = mid - 1;
        }
        const float2 c = s_cen[wv][lo], x = p[v];
        q[v] = make_float2(x.x - c.x, x.y - c.y);
    }
}

// f32 mask [n,1,ps,ps] -> row words.  One wave per (nucleus,row): lane c tests pixel c, the row
// word is the wave ballot -- a coalesced ps*4-byte read and one 8-byte store.
__global__ __launch_bounds__(256) void k_pack_masks(int64_t nrows, const float *__restrict__ masks, int ps,
                                                    uint64_t *__restrict__ rows)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nrows) return;
    const int W = (ps + 63) >> 6; // one word per row up to ps = 64, two words up to 128
    for (int k = 0; k < W; ++k) {
        const int c = 64 * k + lane;
        const bool on = c < ps && masks[r * ps + c] != 0.0f;
        const uint64_t w = __ballot(on);
        if (lane == 0) rows[r * W + k] = w;
    }
}

__global__ __launch_bounds__(256) void k_unpack_masks(int64_t nrows, const uint64_t *__restrict__ rows, int ps,
                                                      float *__restrict__ masks)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nrows) return;
    const int W = (ps + 63) >> 6;
    for (int k = 0; k < W; ++k) {
        const uint64_t w = rows[r * W + k];
        const int c = 64 * k + lane;
        if (c < ps) masks[r * ps + c] = ((w >> lane) & 1ull) ? 1.0f : 0.0f;
    }
}

// Rust's `f32 as i64` (utils.rs:159-162): truncates, saturates, NaN -> 0 (a C cast of such values is undefined).  The result
// is further clamped to +-2^40: no image is that large, so the window arithmetic below behaves the same and cannot overflow.
static __device__ __forceinline__ int64_t rust_f32_as_i64(float v)
{
    if (v != v) return 0;
    const float lim = 1099511627776.0f; // 2^40
    return v >= lim ? (int64_t)1 << 40 : (v <= -lim ? -((int64_t)1 << 40) : (int64_t)v);
}

// utils.rs:159-191: top/left/bottom/right = trunc(c -/+ ps/2) as i64, clipped to the image; the
// crop is pasted at (-min(top,0), -min(left,0)) into a zeroed [3,ps,ps] patch.
// The source is a WINDOW of the image: rows [wy0, wy0 + wh) x columns [wx0, wx0 + ww) of the H x W image, stored contiguously
// as [3][wh][ww] (the whole image: window = image).  With `sel`, block b serves nucleus sel[b] of the call (the nuclei whose
// boxes lie in this window; nfx_stream.cpp streams a slide block by block and gathers per block into one packed patch buffer).
// A box pixel outside the window reads as 0 (cannot happen when the planner's windows cover their nuclei's boxes; the test
// only keeps a wrong plan from becoming an out-of-bounds read).
__global__ __launch_bounds__(1024) void k_gather_patches(int64_t n, const uint32_t *__restrict__ sel, const float *__restrict__ centroids,
                                                        const uint8_t *__restrict__ image, int64_t H, int64_t W, int64_t wy0, int64_t wx0,
                                                        int64_t wh, int64_t ww, int ps, int slide_addressing, uint8_t *__restrict__ patches)
{
    const int64_t i = sel ? (int64_t)sel[blockIdx.x] : (int64_t)blockIdx.x;
    const float cx = centroids[2 * i], cy = centroids[2 * i + 1];
    const float half = (float)ps / 2.0f;
    int64_t t0, l0, b0, r0, oy, ox;
    if (slide_addressing) {
        // load_slide_dataset, utils.rs:97-122: the region's address is `(c - ps/2) as u32` (truncates, negatives and NaN
        // give 0, large values saturate), the region is ps x ps from there and whatever the slide cannot supply stays 0
        auto as_u32 = [](float v) -> int64_t { return v >= 4294967296.0f ? 4294967295ll : (v > 0.0f ? (int64_t)v : 0ll); };
        t0 = as_u32(cy - half);
        l0 = as_u32(cx - half);
        b0 = t0 + ps < H ? t0 + ps : H;
        r0 = l0 + ps < W ? l0 + ps : W;
        oy = 0;
        ox = 0;
    } else {
        // load_image_dataset, utils.rs:159-191
        const int64_t top = rust_f32_as_i64(cy - half), left = rust_f32_as_i64(cx - half);
        const int64_t bottom = rust_f32_as_i64(cy + half), right = rust_f32_as_i64(cx + half);
        t0 = top > 0 ? top : 0, l0 = left > 0 ? left : 0;
        b0 = bottom < H ? bottom : H, r0 = right < W ? right : W;
        oy = -(top < 0 ? top : 0), ox = -(left < 0 ? left : 0);
    }
    const bool in_window = t0 >= wy0 && l0 >= wx0 && b0 <= wy0 + wh && r0 <= wx0 + ww;
    image -= (size_t)wy0 * ww + (size_t)wx0; // window-relative addressing below: ((ch * wh + (row - wy0)) * ww + (col - wx0))
    H = wh;                                  // (the channel stride of the stored window)
    W = ww;
    const int P = ps * ps;
    uint8_t *dst = patches + (size_t)i * 3 * P;
    // a window that lies wholly inside the image -- all but the border nuclei -- is 3 ps rows of ps consecutive bytes each
    // way: 16 bytes per thread and trip (one unaligned load, one aligned store) instead of one
    if (in_window && oy == 0 && ox == 0 && b0 - t0 == ps && r0 - l0 == ps && (ps & 15) == 0 && (reinterpret_cast<uintptr_t>(patches) & 15) == 0) {
        const int cpr = ps >> 4; // 16-byte chunks per row
        const int chunks = 3 * ps * cpr;
        // four chunks per thread, all loads before the first store.  (With nuclei in random order over a large slide every
        // 48-byte row of a window is its own DRAM burst: the kernel then runs at the memory's pace for ~4x the useful
        // bytes, whatever the shape of the copy loop; slide-ordered nuclei hit L2 for the 2.25x overlap of their windows.)
        for (int k0 = threadIdx.x; k0 < chunks; k0 += 4 * blockDim.x) {
            uint4 v[4];
            int at[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + u * (int)blockDim.x;
                const int kk = k < chunks ? k : k0;
                const int rowi = kk / cpr, chunk = kk - rowi * cpr; // rowi = channel * ps + patch row
                const int ch = rowi / ps, pr = rowi - ch * ps;
                const uint8_t *src = image + ((size_t)ch * H + (size_t)(t0 + pr)) * W + (size_t)l0 + 16 * chunk;
                __builtin_memcpy(&v[u], src, 16); // the target allows unaligned vector loads: one global_load_dwordx4
                at[u] = k < chunks ? rowi * ps + 16 * chunk : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (at[u] >= 0) *reinterpret_cast<uint4 *>(dst + at[u]) = v[u];
        }
        return;
    }
    for (int k = threadIdx.x; k < 3 * P; k += blockDim.x) {
        const int ch = k / P, rem = k - ch * P;
        const int pr = rem / ps, pc = rem - pr * ps;
        const int64_t sr = t0 + (pr - oy), sc = l0 + (pc - ox);
        uint8_t v = 0;
        if (in_window && pr >= oy && pc >= ox && sr < b0 && sc < r0) v = image[((size_t)ch * H + sr) * W + sc];
        dst[k] = v;
    }
}

hipError_t launch_preprocess(nfx_ctx *ctx, int64_t n, const uint32_t *ring_off, const float *ring_xy, float *centroids,
                             float *centred)
{
    if (n == 0) return hipSuccess;
    ScopedTiming t(ctx, KS_PREPROCESS);
    hipLaunchKernelGGL(k_preprocess, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n, ring_off, ring_xy, centroids, centred);
    return hipGetLastError();
}

hipError_t launch_pack_masks(nfx_ctx *ctx, int64_t n, const float *masks_f32, int ps, uint64_t *rows)
{
    if (n == 0) return hipSuccess;
    ScopedTiming t(ctx, KS_PACK_MASKS);
    const int64_t nrows = n * ps;
    hipLaunchKernelGGL(k_pack_masks, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, ctx->stream, nrows, masks_f32, ps, rows);
    return hipGetLastError();
}

hipError_t launch_unpack_masks(nfx_ctx *ctx, int64_t n, const uint64_t *rows, int ps, float *masks_f32)
{
    if (n == 0) return hipSuccess;
    ScopedTiming t(ctx, KS_PACK_MASKS);
    const int64_t nrows = n * ps;
    hipLaunchKernelGGL(k_unpack_masks, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, ctx->stream, nrows, rows, ps, masks_f32);
    return hipGetLastError();
}

hipError_t launch_gather_patches_window(nfx_ctx *ctx, int64_t n_sel, const uint32_t *sel, const float *centroids, const uint8_t *window, int64_t H,
                                        int64_t W, int64_t wy0, int64_t wx0, int64_t wh, int64_t ww, int ps, uint8_t *patches)
{
    if (n_sel == 0) return hipSuccess;
    ScopedTiming t(ctx, KS_GATHER);
    // one block per nucleus; when the 16-byte path applies a thread moves four chunks (128 threads for 48x48, 192 for 64x64)
    int threads = 256;
    if ((ps & 15) == 0) {
        const int chunks = 3 * ps * (ps >> 4);
        threads = ((chunks + 3) / 4 + 63) / 64 * 64; // four chunks per thread
        threads = threads > 1024 ? 1024 : threads;
    }
    hipLaunchKernelGGL(k_gather_patches, dim3((unsigned)n_sel), dim3((unsigned)threads), 0, ctx->stream, n_sel, sel, centroids, window, H, W, wy0, wx0,
                       wh, ww, ps, ctx->addressing == NFX_ADDR_SLIDE ? 1 : 0, patches);
    return hipGetLastError();
}

hipError_t launch_gather_patches(nfx_ctx *ctx, int64_t n, const float *centroids, const uint8_t *image, int64_t H,
                                 int64_t W, int ps, uint8_t *patches)
{
    return launch_gather_patches_window(ctx, n, nullptr, centroids, image, H, W, 0, 0, H, W, ps, patches);
}

}  // namespace nfx
