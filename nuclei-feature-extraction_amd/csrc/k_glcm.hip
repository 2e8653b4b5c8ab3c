// k_glcm.hip -- the "glcm" feature set: 4 quantisation levels x 4 offsets x 14 Haralick features.
//
// Replaces (reference file:line):
//   GlcmFeatureSet::compute_features_batched     texture.rs:25-167 (GLCM_LEVELS, OFFSETS 19-20)
//   tch_utils::glcm::glcm + glcm_features         call site texture.rs:40-63
//
// The reference materialises a dense [N,L,L] f32 matrix per (level, offset) -- 258 KB per nucleus at
// L=254 -- and reduces it fourteen times.  A nucleus has at most ps*(ps-1) co-occurring pairs (a few
// thousand), so the matrix is >90 % zeros.  Here nothing dense is ever reduced:
//   * a persistent 256-thread workgroup owns one nucleus at a time; the patch is read from HBM once
//     per nucleus (gray values stay in registers across the four levels);
//   * the symmetric matrix lives in LDS as a TRIANGULAR table of packed 16-bit counters (64.8 KB at
//     L=254, two workgroups per CU).  Pass 1 ds_add's one counter per pair, pass 2 reads the
//     multiplicity m of every pair's cell (sum G^2 and sum G ln G are sums over pairs of 2m|4m and
//     2 ln m | 2 ln 2m), pass 3 stores zero to exactly the touched counters -- the table is never
//     swept;
//   * every other feature is a closed form of four exact integer power sums over the pairs
//     (sum s, s^2, d, d^2 with s=a+b, d=|a-b|) plus two 1-D histogram entropies.
// All sums are integers (exact) or f64; the 14 features of the 16 (level,offset) combinations are
// evaluated by 16 lanes at the end.
#include "nfx_internal.h"

namespace nfx {

static constexpr int GL_THREADS = 256;
static constexpr int GL_MAXK = (NFX_MAX_PATCH_SIZE * NFX_MAX_PATCH_SIZE) / GL_THREADS; // 16 pixels per thread
static constexpr int GL_TAB_WORDS = (254 * 255 / 2 + 1) / 2 + 1;                       // packed u16 triangular, L=254
static constexpr int GL_NACC = 10;

__constant__ int kGlcmLevels[4] = {32, 64, 128, 254};                     // texture.rs:19
__constant__ int kGlcmOffsets[4][2] = {{0, 1}, {1, 1}, {1, 0}, {1, -1}};  // texture.rs:20 (dy,dx)

template <typename T> struct GPx;
template <> struct GPx<uint8_t> {
    static __device__ __forceinline__ float ld(const uint8_t *p, size_t i) { return (float)p[i] / 255.0f; }
};
template <> struct GPx<float> {
    static __device__ __forceinline__ float ld(const float *p, size_t i) { return p[i]; }
};

static __device__ __forceinline__ double wsum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

static __device__ __forceinline__ double ln_int(unsigned m, const double *lnk)
{
    return m < 256u ? lnk[m] : log((double)m);
}

template <typename T>
__global__ __launch_bounds__(GL_THREADS, 2) void k_glcm(int64_t n, int ps, const T *__restrict__ patches,
                                                        const uint64_t *__restrict__ mask_rows, float *__restrict__ out,
                                                        int F, int col0)
{
    __shared__ unsigned s_tab[GL_TAB_WORDS];
    __shared__ unsigned s_hx[256];
    __shared__ unsigned s_hs[512];
    __shared__ uint64_t s_rows[64];
    __shared__ uint8_t s_q[NFX_MAX_PATCH_SIZE * NFX_MAX_PATCH_SIZE];
    __shared__ double s_lnk[256];
    __shared__ double s_idm[256];
    __shared__ double s_part[4][GL_NACC];
    __shared__ double s_comb[16][GL_NACC];

    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int P = ps * ps;
    for (int w = t; w < GL_TAB_WORDS; w += GL_THREADS) s_tab[w] = 0u;
    s_hx[t] = 0u; s_hs[t] = 0u; s_hs[t + 256] = 0u;
    s_lnk[t] = t > 0 ? log((double)t) : 0.0;
    s_idm[t] = 1.0 / (1.0 + (double)t * (double)t);
    __syncthreads();

    for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
        if (t < 64) s_rows[t] = t < ps ? mask_rows[i * ps + t] : 0ull;
        // gray = ((r+g)+b)/3 in f32 (texture.rs:36), kept in registers for the four levels
        float gray[GL_MAXK];
        const T *pp = patches + (size_t)i * 3 * P;
#pragma unroll
        for (int k = 0; k < GL_MAXK; ++k) {
            const int p = t + k * GL_THREADS;
            float gsum = 0.0f;
            if (p < P) {
                gsum = GPx<T>::ld(pp, p) + GPx<T>::ld(pp, P + p);
                gsum = gsum + GPx<T>::ld(pp, 2 * P + p);
                gsum = gsum / 3.0f;
            }
            gray[k] = gsum;
        }
        __syncthreads();

        for (int li = 0; li < 4; ++li) {
            const int L = kGlcmLevels[li];
            // quantise: q = min(floor(gray*L), L-1); 0xFF marks pixels outside the mask
#pragma unroll
            for (int k = 0; k < GL_MAXK; ++k) {
                const int p = t + k * GL_THREADS;
                if (p < P) {
                    const int r = p / ps, c = p - r * ps;
                    int q = (int)floorf(gray[k] * (float)L);
                    q = q < 0 ? 0 : (q > L - 1 ? L - 1 : q);
                    s_q[p] = ((s_rows[r] >> c) & 1ull) ? (uint8_t)q : (uint8_t)0xFF;
                }
            }
            __syncthreads();

            for (int oi = 0; oi < 4; ++oi) {
                const int dy = kGlcmOffsets[oi][0], dx = kGlcmOffsets[oi][1];
                // ---- pass 1: count ---------------------------------------------------------------
                int np = 0, S1 = 0, D1 = 0;
                long long S2 = 0, D2 = 0;
                double idm = 0.0;
                unsigned keys[GL_MAXK];
#pragma unroll
                for (int k = 0; k < GL_MAXK; ++k) {
                    const int p = t + k * GL_THREADS;
                    unsigned key = 0xFFFFFFFFu;
                    if (p < P) {
                        const int r = p / ps, c = p - r * ps;
                        const int r2 = r + dy, c2 = c + dx;
                        const unsigned a = s_q[p];
                        if (a != 0xFFu && r2 < ps && c2 >= 0 && c2 < ps) {
                            const unsigned b = s_q[r2 * ps + c2];
                            if (b != 0xFFu) {
                                const unsigned lo = a < b ? a : b, hi = a < b ? b : a;
                                const unsigned idx = hi * (hi + 1u) / 2u + lo;
                                key = idx | (lo == hi ? 0x80000000u : 0u);
                                atomicAdd(&s_tab[idx >> 1], 1u << ((idx & 1u) * 16u));
                                atomicAdd(&s_hx[a], 1u);
                                atomicAdd(&s_hx[b], 1u);
                                atomicAdd(&s_hs[a + b], 1u);
                                const int s = (int)(a + b), d = (int)(hi - lo);
                                np += 1; S1 += s; D1 += d; S2 += s * s; D2 += d * d;
                                idm += s_idm[d];
                            }
                        }
                    }
                    keys[k] = key;
                }
                __syncthreads();
                // ---- pass 2: multiplicities and histogram entropies ----------------------------------
                long long asmn = 0;
                double ents = 0.0;
#pragma unroll
                for (int k = 0; k < GL_MAXK; ++k) {
                    const unsigned key = keys[k];
                    if (key != 0xFFFFFFFFu) {
                        const unsigned idx = key & 0x7FFFFFFFu;
                        const unsigned m = (s_tab[idx >> 1] >> ((idx & 1u) * 16u)) & 0xFFFFu;
                        if (key & 0x80000000u) { asmn += 4ll * m; ents += 2.0 * ln_int(2u * m, s_lnk); }
                        else                   { asmn += 2ll * m; ents += 2.0 * ln_int(m, s_lnk); }
                    }
                }
                double hxs = 0.0, hss = 0.0;
                {
                    const unsigned h = s_hx[t];
                    if (h > 1u) hxs = (double)h * ln_int(h, s_lnk);
                    const unsigned a0 = s_hs[t], a1 = s_hs[t + 256];
                    if (a0 > 1u) hss = (double)a0 * ln_int(a0, s_lnk);
                    if (a1 > 1u) hss += (double)a1 * ln_int(a1, s_lnk);
                }
                double acc[GL_NACC] = {(double)np, (double)S1, (double)S2, (double)D1, (double)D2,
                                       idm, (double)asmn, ents, hxs, hss};
#pragma unroll
                for (int a = 0; a < GL_NACC; ++a) {
                    const double v = wsum(acc[a]);
                    if (lane == 0) s_part[wv][a] = v;
                }
                __syncthreads();
                // ---- pass 3: clear exactly what was touched; publish the combination's sums ------------
#pragma unroll
                for (int k = 0; k < GL_MAXK; ++k) {
                    const unsigned key = keys[k];
                    if (key != 0xFFFFFFFFu) {
                        const unsigned idx = key & 0x7FFFFFFFu;
                        reinterpret_cast<unsigned short *>(s_tab)[idx] = 0;
                    }
                }
                s_hx[t] = 0u; s_hs[t] = 0u; s_hs[t + 256] = 0u;
                if (t < GL_NACC) s_comb[li * 4 + oi][t] = s_part[0][t] + s_part[1][t] + s_part[2][t] + s_part[3][t];
                __syncthreads();
            }
        }

        // ---- 16 lanes evaluate the 14 features of their (level, offset) combination -------------------
        if (t < 16) {
            const double *a = s_comb[t];
            const double N = a[0], S1 = a[1], S2 = a[2], D1 = a[3], D2 = a[4], IDM = a[5], ASMN = a[6], ENTS = a[7],
                         HXS = a[8], HSS = a[9];
            float *o = out + i * (int64_t)F + col0 + t * 14;
            if (N == 0.0) {
                const float qnan = __builtin_nanf("");
#pragma unroll
                for (int k = 0; k < 14; ++k) o[k] = qnan;
            } else {
                const double T2 = 2.0 * N;
                const double mu = S1 / T2;
                const double ei2 = (S2 + D2) / (4.0 * N);
                const double var = ei2 - mu * mu;
                const double eij = (S2 - D2) / (4.0 * N);
                const double con = D2 / N, dis = D1 / N, sa = S1 / N;
                const double sv = S2 / N - sa * sa;
                const double se = log(N) - HSS / N;
                const double ent = log(T2) - ENTS / T2;
                const double asm_ = ASMN / (T2 * T2);
                const double dv = D2 / N - dis * dis;
                const double hx = log(T2) - HXS / T2;
                const double hxy1 = 2.0 * hx; // = HXY2 for a symmetric matrix (marginalisation)
                double a2 = 1.0 - exp(-2.0 * (hxy1 - ent));
                a2 = a2 > 0.0 ? a2 : 0.0;
                o[0] = (float)((eij - mu * mu) / var);
                o[1] = (float)con;
                o[2] = (float)dis;
                o[3] = (float)ent;
                o[4] = (float)asm_;
                o[5] = (float)sa;
                o[6] = (float)sv;
                o[7] = (float)se;
                o[8] = (float)var;
                o[9] = (float)(IDM / N);
                o[10] = (float)dis;
                o[11] = (float)dv;
                o[12] = (float)((ent - hxy1) / hx);
                o[13] = (float)sqrt(a2);
            }
        }
        __syncthreads();
    }
}

hipError_t launch_glcm(nfx_ctx *ctx, const Job &job, int col0)
{
    if (job.n == 0) return hipSuccess;
    ScopedTiming t(ctx, KS_GLCM);
    const int64_t maxb = 256 * 2 * 2; // two resident workgroups per CU, x2 for tail balance
    const unsigned grid = (unsigned)(job.n < maxb ? job.n : maxb);
    if (job.patches_f32)
        hipLaunchKernelGGL(k_glcm<float>, dim3(grid), dim3(GL_THREADS), 0, ctx->stream, job.n, job.ps,
                           reinterpret_cast<const float *>(job.patches), job.mask_rows, job.out, job.F, col0);
    else
        hipLaunchKernelGGL(k_glcm<uint8_t>, dim3(grid), dim3(GL_THREADS), 0, ctx->stream, job.n, job.ps,
                           reinterpret_cast<const uint8_t *>(job.patches), job.mask_rows, job.out, job.F, col0);
    return hipGetLastError();
}

}  // namespace nfx
