// k_util.hip -- polygon preprocessing, mask format conversion, patch gathering.
//
// Replaces (reference file:line):
//   preprocess_polygon                 utils.rs:54-74
//   load_image_dataset patch addressing utils.rs:159-191
#include "nfx_internal.h"

namespace nfx {

// One lane per nucleus: the centroid is a SEQUENTIAL f32 sum over the listed ring vertices
// (utils.rs:56-64), so the summation order is part of the result and is kept.
__global__ __launch_bounds__(256) void k_preprocess(int64_t n, const uint32_t *__restrict__ ring_off,
                                                    const float *__restrict__ ring_xy, float *__restrict__ centroids,
                                                    float *__restrict__ centred)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t v0 = ring_off[i], v1 = ring_off[i + 1];
    const float2 *p = reinterpret_cast<const float2 *>(ring_xy);
    float ax = 0.0f, ay = 0.0f;
    for (uint32_t k = v0; k < v1; ++k) {
        const float2 v = p[k];
        ax += v.x;
        ay += v.y;
    }
    const float cnt = (float)(v1 - v0);
    const float cx = ax / cnt, cy = ay / cnt;
    if (centroids) {
        centroids[2 * i] = cx;
        centroids[2 * i + 1] = cy;
    }
    if (centred) {
        float2 *q = reinterpret_cast<float2 *>(centred);
        for (uint32_t k = v0; k < v1; ++k) {
            const float2 v = p[k];
            q[k] = make_float2(v.x - cx, v.y - cy);
        }
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
    const bool on = lane < ps && masks[r * ps + lane] != 0.0f;
    const uint64_t w = __ballot(on);
    if (lane == 0) rows[r] = w;
}

__global__ __launch_bounds__(256) void k_unpack_masks(int64_t nrows, const uint64_t *__restrict__ rows, int ps,
                                                      float *__restrict__ masks)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nrows) return;
    const uint64_t w = rows[r];
    if (lane < ps) masks[r * ps + lane] = ((w >> lane) & 1ull) ? 1.0f : 0.0f;
}

// utils.rs:159-191: top/left/bottom/right = trunc(c -/+ ps/2) as i64, clipped to the image; the
// crop is pasted at (-min(top,0), -min(left,0)) into a zeroed [3,ps,ps] patch.
__global__ __launch_bounds__(256) void k_gather_patches(int64_t n, const float *__restrict__ centroids,
                                                        const uint8_t *__restrict__ image, int64_t H, int64_t W, int ps,
                                                        uint8_t *__restrict__ patches)
{
    const int64_t i = blockIdx.x;
    const float cx = centroids[2 * i], cy = centroids[2 * i + 1];
    const float half = (float)ps / 2.0f;
    const int64_t top = (int64_t)(cy - half), left = (int64_t)(cx - half);
    const int64_t bottom = (int64_t)(cy + half), right = (int64_t)(cx + half);
    const int64_t t0 = top > 0 ? top : 0, l0 = left > 0 ? left : 0;
    const int64_t b0 = bottom < H ? bottom : H, r0 = right < W ? right : W;
    const int64_t oy = -(top < 0 ? top : 0), ox = -(left < 0 ? left : 0);
    const int P = ps * ps;
    uint8_t *dst = patches + (size_t)i * 3 * P;
    for (int k = threadIdx.x; k < 3 * P; k += blockDim.x) {
        const int ch = k / P, rem = k - ch * P;
        const int pr = rem / ps, pc = rem - pr * ps;
        const int64_t sr = t0 + (pr - oy), sc = l0 + (pc - ox);
        uint8_t v = 0;
        if (pr >= oy && pc >= ox && sr < b0 && sc < r0) v = image[((size_t)ch * H + sr) * W + sc];
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

hipError_t launch_gather_patches(nfx_ctx *ctx, int64_t n, const float *centroids, const uint8_t *image, int64_t H,
                                 int64_t W, int ps, uint8_t *patches)
{
    if (n == 0) return hipSuccess;
    ScopedTiming t(ctx, KS_GATHER);
    hipLaunchKernelGGL(k_gather_patches, dim3((unsigned)n), dim3(256), 0, ctx->stream, n, centroids, image, H, W, ps, patches);
    return hipGetLastError();
}

}  // namespace nfx
