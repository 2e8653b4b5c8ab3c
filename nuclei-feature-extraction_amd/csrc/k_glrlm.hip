// k_glrlm.hip -- the "glrlm" feature set: 4 directions x 17 run-length features.
//
// Replaces (reference file:line):
//   GLRLMFeatureSet::compute_features_batched    texture.rs:181-310 (GLRLM_LEVELS=24,
//                                                GLRLM_MAX_LENGTH=16, DIRECTIONS, 174-176)
//   tch_utils::glrlm::{glrlm, glrlm_features}    call site texture.rs:193-215
//
// ONE WAVEFRONT PER NUCLEUS (persistent, grid-stride; four independent waves per workgroup, each with a
// private LDS slice and no s_barrier -- a wave's LDS ops execute in order).  The 24-level map of the patch
// is staged in LDS as one byte per pixel (0xFF outside the mask); lanes walk the mask rows
// (lane = column), a lane whose pixel starts a run follows it for at most 16 steps and bumps one
// of the 384 run-length bins with an LDS atomic.  The 17 features are weighted sums over the bins:
// with bin = lane + 64k a lane always sees the same run length l = lane % 16 and six fixed gray
// levels, so every weight is a register computed once per kernel; row/column sums for the
// non-uniformity features are 16-lane / 4-lane butterfly reductions.  No table, no LDS traffic
// beyond the bins themselves.
#include "nfx_internal.h"
#include "wave_ops.h"

namespace nfx {

static constexpr int RL_LEVELS = 24, RL_MAXLEN = 16, RL_BINS = RL_LEVELS * RL_MAXLEN; // 384 = 6 * 64
__constant__ int kRlDirs[4][2] = {{1, 0}, {1, 1}, {0, 1}, {-1, 1}};                  // (dx,dy) texture.rs:176

template <typename T> struct RPx;
template <> struct RPx<uint8_t> {
    static __device__ __forceinline__ float ld(const uint8_t *p, size_t i) { return (float)p[i] / 255.0f; }
};
template <> struct RPx<float> {
    static __device__ __forceinline__ float ld(const float *p, size_t i) { return p[i]; }
};

static __device__ __forceinline__ double rsum(double v) { return wave_total_f64(v); }

template <typename T>
__global__ __launch_bounds__(256) void k_glrlm(int64_t n, int ps, const T *__restrict__ patches,
                                              const uint64_t *__restrict__ mask_rows, float *__restrict__ out, int F,
                                              int col0)
{
    __shared__ uint8_t s_q_all[4][NFX_MAX_PATCH_SIZE * NFX_MAX_PATCH_SIZE];
    __shared__ unsigned s_R_all[4][RL_BINS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint8_t *s_q = s_q_all[wv];
    unsigned *s_R = s_R_all[wv];
    const int P = ps * ps;
    const int rpi = ps <= 32 ? 2 : 1, cw = ps <= 32 ? 32 : 64;
    const int c = lane & (cw - 1), rsub = rpi == 2 ? (lane >> 5) : 0;

    // per-lane constant weights: l = lane%16 + 1, g_k = lane/16 + 4k + 1
    const double l = (double)((lane & 15) + 1);
    const double l2 = l * l, il2 = 1.0 / l2;
    double g2[6], ig2[6], wmid[6], wext[6];
    const double cmid = ((double)RL_LEVELS + 1.0) / 2.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double g = (double)((lane >> 4) + 4 * k + 1);
        g2[k] = g * g;
        ig2[k] = 1.0 / g2[k];
        const double e = (g - cmid) * (g - cmid);
        wext[k] = e;
        wmid[k] = 1.0 / (1.0 + e);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) s_R[lane + 64 * k] = 0u;
    wave_lds_fence();

    for (int64_t i = (int64_t)blockIdx.x * 4 + wv; i < n; i += (int64_t)gridDim.x * 4) {
        const uint64_t myrow = lane < ps ? mask_rows[i * ps + lane] : 0ull;
        const uint64_t anyrow = __ballot(myrow != 0ull);
        const double Np = (double)wave_total_i32(__popcll(myrow)); // pixel_count, texture.rs:196
        const T *pp = patches + (size_t)i * 3 * P;

        // stage the 24-level map
        for (int r0 = 0; r0 < ps; r0 += rpi) {
            const int r = r0 + rsub;
            const uint64_t w = __shfl(myrow, r & 63, 64);
            if (r < ps && c < ps) {
                const int p = r * ps + c;
                uint8_t qv = 0xFF;
                if ((w >> c) & 1ull) {
                    float gs = RPx<T>::ld(pp, p) + RPx<T>::ld(pp, P + p);
                    gs = gs + RPx<T>::ld(pp, 2 * P + p);
                    gs = gs / 3.0f;
                    int q = (int)floorf(gs * (float)RL_LEVELS);
                    q = q < 0 ? 0 : (q > RL_LEVELS - 1 ? RL_LEVELS - 1 : q);
                    qv = (uint8_t)q;
                }
                s_q[p] = qv;
            }
        }
        wave_lds_fence();

        for (int d = 0; d < 4; ++d) {
            const int dx = kRlDirs[d][0], dy = kRlDirs[d][1];
            for (int r0 = 0; r0 < ps; r0 += rpi) {
                if (((anyrow >> r0) & (rpi == 2 ? 3ull : 1ull)) == 0ull) continue;
                const int r = r0 + rsub;
                if (r < ps && c < ps) {
                    const unsigned q = s_q[r * ps + c];
                    if (q != 0xFFu) {
                        const int rp = r - dy, cp = c - dx;
                        const bool cont = rp >= 0 && rp < ps && cp >= 0 && cp < ps && s_q[rp * ps + cp] == q;
                        if (!cont) { // run start: follow it, lengths beyond 16 share the last bin
                            int len = 1, rn = r + dy, cn = c + dx;
                            while (len < RL_MAXLEN && rn >= 0 && rn < ps && cn >= 0 && cn < ps && s_q[rn * ps + cn] == q) {
                                ++len; rn += dy; cn += dx;
                            }
                            atomicAdd(&s_R[q * RL_MAXLEN + (len - 1)], 1u);
                        }
                    }
                }
            }
            wave_lds_fence();
            // weighted sums over the bins this lane owns
            double nr = 0, sre = 0, lre = 0, lgre = 0, hgre = 0, srlge = 0, srhge = 0, lrlge = 0, lrhge = 0;
            double srmge = 0, lrmge = 0, srege = 0, lrege = 0, rlm = 0, gln = 0, colsum = 0;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const double v = (double)s_R[lane + 64 * k];
                s_R[lane + 64 * k] = 0u;
                nr += v;
                const double vs = v * il2, vl = v * l2;
                sre += vs; lre += vl;
                lgre += v * ig2[k]; hgre += v * g2[k];
                srlge += vs * ig2[k]; srhge += vs * g2[k];
                lrlge += vl * ig2[k]; lrhge += vl * g2[k];
                srmge += vs * wmid[k]; lrmge += vl * wmid[k];
                srege += vs * wext[k]; lrege += vl * wext[k];
                rlm += v * l;
                colsum += v;
                // row sum of gray level g_k over the 16 lanes that share lane/16
                const double rs = row16_total_f64(v);
                if ((lane & 15) == 0) gln += rs * rs;
            }
            // column sum of run length l over the 4 lanes that share lane%16
            colsum += __shfl_xor(colsum, 16, 64);
            colsum += __shfl_xor(colsum, 32, 64);
            double rln = lane < 16 ? colsum * colsum : 0.0;
            nr = rsum(nr); sre = rsum(sre); lre = rsum(lre); lgre = rsum(lgre); hgre = rsum(hgre);
            srlge = rsum(srlge); srhge = rsum(srhge); lrlge = rsum(lrlge); lrhge = rsum(lrhge);
            srmge = rsum(srmge); lrmge = rsum(lrmge); srege = rsum(srege); lrege = rsum(lrege);
            rlm = rsum(rlm); gln = rsum(gln); rln = rsum(rln);
            if (lane == 0) {
                float *o = out + i * (int64_t)F + col0 + 17 * d; // column order texture.rs:244-301
                const double mean = rlm / nr;
                o[0] = (float)(sre / nr);    o[1] = (float)(lre / nr);
                o[2] = (float)(gln / nr);    o[3] = (float)(rln / nr);
                o[4] = (float)(lgre / nr);   o[5] = (float)(hgre / nr);
                o[6] = (float)(srlge / nr);  o[7] = (float)(srhge / nr);
                o[8] = (float)(lrlge / nr);  o[9] = (float)(lrhge / nr);
                o[10] = (float)(srmge / nr); o[11] = (float)(lrmge / nr);
                o[12] = (float)(srege / nr); o[13] = (float)(lrege / nr);
                o[14] = (float)(nr / Np);
                o[15] = (float)mean;
                o[16] = (float)(lre / nr - mean * mean); // sum R (l-mean)^2 / Nr
            }
            wave_lds_fence();
        }
    }
}

hipError_t launch_glrlm(nfx_ctx *ctx, const Job &job, int col0)
{
    if (job.n == 0) return hipSuccess;
    ScopedTiming t(ctx, KS_GLRLM);
    const int64_t maxb = 256 * 7 * 2, need = (job.n + 3) / 4; // 7 workgroups (28 waves) fit a CU's LDS
    const unsigned grid = (unsigned)(need < maxb ? need : maxb);
    if (job.patches_f32)
        hipLaunchKernelGGL(k_glrlm<float>, dim3(grid), dim3(256), 0, ctx->stream, job.n, job.ps,
                           reinterpret_cast<const float *>(job.patches), job.mask_rows, job.out, job.F, col0);
    else
        hipLaunchKernelGGL(k_glrlm<uint8_t>, dim3(grid), dim3(256), 0, ctx->stream, job.n, job.ps,
                           reinterpret_cast<const uint8_t *>(job.patches), job.mask_rows, job.out, job.F, col0);
    return hipGetLastError();
}

}  // namespace nfx
