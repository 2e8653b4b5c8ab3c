// k_glcm.hip -- the "glcm" feature set: 4 quantisation levels x 4 offsets x 14 Haralick features.
//
// Replaces (reference file:line):
//   GlcmFeatureSet::compute_features_batched     texture.rs:25-167 (GLCM_LEVELS, OFFSETS 19-20)
//   tch_utils::glcm::glcm + glcm_features         call site texture.rs:40-63
//
// The reference materialises a dense [N,L,L] f32 matrix per (level, offset) -- 258 KB per nucleus at
// L=254 -- and reduces it fourteen times.  A nucleus has at most ps*(ps-1) co-occurring pairs (a few
// thousand), so the matrix is >90 % zeros.  Here nothing dense is ever built or reduced:
//
//   * ONE WAVEFRONT PER NUCLEUS (single-wave workgroups, persistent): no cross-wave barrier or reduction,
//     lanes stay busy to within K mod 64.  The mask's pixels are compacted into an LDS list of
//     (position, four packed levels); the u8 patch is fetched with 16-byte coalesced loads into LDS (one HBM round
//     trip per nucleus) and every masked pixel is quantised exactly once;
//   * co-occurrence cells are counted with RETURNING LDS atomics.  The value the atomic returns is the
//     pair's rank k inside its cell, and
//         sum_cells G^2     = sum_pairs 2(2k+1)            (4(2k+1) on the diagonal)
//         sum_cells G ln G  = sum_pairs 2[h(k+1)-h(k)],  h(x) = x ln x   (diag: H(x) = 2x ln 2x)
//     telescope over the ranks (difference tables for every possible rank live in global memory, L1/L2
//     resident), so the multiplicities are never read back and no cell needs an owner;
//   * L=32 and L=64 use dense triangular tables of packed 16-bit counters (5 KB); L=128 and L=254 use
//     open-addressing HASH tables sized 2x the pair capacity (8 KB for masks up to 1024 pixels) instead
//     of 16.5 / 64.8 KB dense ones: a wave's whole state is 21 KB, seven nuclei in flight per CU.  Tables
//     are reset by sweeping them with 16-byte stores (a handful of instructions per wave);
//   * levels 32/64/128 share one pass over the pairs of an offset, level 254 takes a second pass in
//     the same table memory;
//   * every other feature is a closed form of four exact integer power sums over the pairs
//     (sum s, s^2, d, d^2 with s=a+b, d=|a-b|) plus the two 1-D histogram entropies.
// One instantiation per mask-size class (hash capacity).  The ten sums of every (level, offset) combination go
// to a scratch array; k_glcm_finish evaluates the 14 features with one lane per combination.
#include <cmath>
#include <cstdlib>
#include <vector>

#include "nfx_internal.h"
#include "wave_ops.h"

namespace nfx {

static constexpr int GL_NACC = 10;
static constexpr int GL_RANKS = 4096;      // rank / count range of the global difference tables
static constexpr int GL_HIST_WORDS = 768;
static constexpr unsigned GL_EMPTY = 0xFFFFFFFFu;
static constexpr int GL_D64_OFF = 272;     // pass A: d32 at 0 (264 words) | d64 at 272 (1040 words) | h128 at 1320
static constexpr int GL_H128_OFF = 1320;   // pass B: h254 at 0

__constant__ int kGlcmOffsets[4][2] = {{0, 1}, {1, 1}, {1, 0}, {1, -1}}; // texture.rs:20 (dy,dx)

struct GlcmLuts {           // device tables built on the host
    double dO[GL_RANKS];    // 2[h(k+1)-h(k)]
    double dD[GL_RANKS];    // H(k+1)-H(k)
    double xlnx[GL_RANKS];  // x ln x
};

struct LevelAcc {
    unsigned S1, S2, D1, D2, asmn;
    float idm;
    double ents;
};

static __device__ __forceinline__ unsigned quant(float gray, int L)
{
    int q = (int)floorf(gray * (float)L);
    q = q < 0 ? 0 : (q > L - 1 ? L - 1 : q);
    return (unsigned)q;
}

template <typename T>
static __device__ __forceinline__ float gray_at(const T *pp, int P, int p)
{
    float gs;
    if (sizeof(T) == 1) {
        const uint8_t *pb = reinterpret_cast<const uint8_t *>(pp);
        gs = (float)pb[p] / 255.0f + (float)pb[P + p] / 255.0f; // utils.rs:172
        gs = gs + (float)pb[2 * P + p] / 255.0f;
    } else {
        const float *pf = reinterpret_cast<const float *>(pp);
        gs = pf[p] + pf[P + p];
        gs = gs + pf[2 * P + p];
    }
    return gs / 3.0f; // texture.rs:36
}

// ---- per-pair update -----------------------------------------------------------------------------------
static __device__ __forceinline__ void pair_sums(unsigned a, unsigned b, unsigned rk, const GlcmLuts *__restrict__ lut,
                                                 unsigned *hx, unsigned *hs, LevelAcc &A)
{
    const unsigned lo = a < b ? a : b, hi = a < b ? b : a;
    atomicAdd(&hx[a], 1u);
    atomicAdd(&hx[b], 1u);
    atomicAdd(&hs[a + b], 1u);
    const unsigned s = a + b, d = hi - lo;
    A.S1 += s; A.S2 += s * s; A.D1 += d; A.D2 += d * d;
    A.idm += __builtin_amdgcn_rcpf(1.0f + (float)(d * d)); // 1/(1+d^2), 1 ulp
    const bool diag = lo == hi;
    A.asmn += (diag ? 4u : 2u) * (2u * rk + 1u);
    A.ents += diag ? lut->dD[rk] : lut->dO[rk];
}

static __device__ __forceinline__ unsigned tri_index(unsigned a, unsigned b)
{
    const unsigned lo = a < b ? a : b, hi = a < b ? b : a;
    return hi * (hi + 1u) / 2u + lo;
}

// dense triangular table of packed 16-bit counters: returns the pair's rank in its cell
static __device__ __forceinline__ unsigned dense_rank(unsigned *tab, unsigned idx)
{
    const unsigned sh = (idx & 1u) * 16u;
    return (atomicAdd(&tab[idx >> 1], 1u << sh) >> sh) & 0xFFFFu;
}

// open-addressing hash table, one word per cell: key << 16 | count (key <= 32384 < 0xFFFF = the EMPTY key)
template <int LOG2SLOTS>
static __device__ __forceinline__ unsigned hash_rank(unsigned *tab, unsigned key)
{
    constexpr unsigned MASK = (1u << LOG2SLOTS) - 1u;
    unsigned h = (key * 2654435761u) >> (32 - LOG2SLOTS);
    for (;;) {
        const unsigned cur = tab[h];
        if ((cur >> 16) == key) break;
        if (cur == GL_EMPTY) {
            const unsigned prev = atomicCAS(&tab[h], GL_EMPTY, key << 16);
            if (prev == GL_EMPTY || (prev >> 16) == key) break;
        }
        h = (h + 1u) & MASK;
    }
    return atomicAdd(&tab[h], 1u) & 0xFFFFu;
}

static __device__ __forceinline__ void sweep(unsigned *tab, int words, unsigned value, int lane)
{
    uint4 *t4 = reinterpret_cast<uint4 *>(tab);
    const uint4 v = make_uint4(value, value, value, value);
    for (int k = lane; k < words / 4; k += 64) t4[k] = v;
}

// histogram entropies (sum n ln n per level and kind), clearing the words; the NLV marginal histograms end at
// hx_end[], the sum histograms at hs_end[].  All twelve table look-ups of a lane are issued unconditionally and
// together (x ln x is 0 for 0 and 1), so the lane pays one L2 round trip instead of twelve dependent ones.
template <int NLV>
static __device__ __forceinline__ void hist_fold(unsigned *hist, const GlcmLuts *__restrict__ lut, int lane, const int (&hx_end)[NLV],
                                                 const int (&hs_end)[NLV], double (&hxs)[NLV], double (&hss)[NLV])
{
    constexpr int NB = GL_HIST_WORDS / 64;
    unsigned v[NB];
    double e[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) { v[j] = hist[lane + 64 * j]; hist[lane + 64 * j] = 0u; }
#pragma unroll
    for (int j = 0; j < NB; ++j) e[j] = lut->xlnx[v[j] < (unsigned)GL_RANKS ? v[j] : 0u];
#pragma unroll
    for (int l = 0; l < NLV; ++l) { hxs[l] = 0.0; hss[l] = 0.0; }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int b = lane + 64 * j;
        double ej = e[j];
        if (v[j] >= (unsigned)GL_RANKS) ej = (double)v[j] * log((double)v[j]); // beyond the table: never for ps <= 45
        bool done = false;
#pragma unroll
        for (int l = 0; l < NLV; ++l)
            if (!done && b < hx_end[l]) { hxs[l] += ej; done = true; }
#pragma unroll
        for (int l = 0; l < NLV; ++l)
            if (!done && b < hs_end[l]) { hss[l] += ej; done = true; }
    }
}

static __device__ __forceinline__ void publish(double *__restrict__ dst, unsigned np, const LevelAcc &A, double hxs, double hss, int lane)
{
    const double v1 = (double)wave_total_u32(A.S1), v2 = (double)wave_total_u32(A.S2);
    const double v3 = (double)wave_total_u32(A.D1), v4 = (double)wave_total_u32(A.D2);
    const double v5 = (double)wave_total_f32(A.idm), v6 = (double)wave_total_u32(A.asmn);
    const double v7 = wave_total_f64(A.ents), v8 = wave_total_f64(hxs), v9 = wave_total_f64(hss);
    if (lane == 0) {
        dst[0] = (double)np; dst[1] = v1; dst[2] = v2; dst[3] = v3; dst[4] = v4;
        dst[5] = v5; dst[6] = v6; dst[7] = v7; dst[8] = v8; dst[9] = v9;
    }
}

// ---- the kernel -------------------------------------------------------------------------------------------
// LOG2SLOTS: hash capacity; serves nuclei with k_lo < K <= SLOTS/2 mask pixels.
template <typename T, int LOG2SLOTS>
__global__ __launch_bounds__(64) void k_glcm(int64_t n, int ps, int k_lo, const T *__restrict__ patches,
                                             const uint64_t *__restrict__ mask_rows, const GlcmLuts *__restrict__ lut,
                                             double *__restrict__ sums)
{
    constexpr int SLOTS = 1 << LOG2SLOTS, QCAP = SLOTS / 2;
    constexpr int TAB_WORDS = GL_H128_OFF + SLOTS;
    __shared__ __attribute__((aligned(16))) unsigned s_tab[TAB_WORDS];
    __shared__ __attribute__((aligned(16))) unsigned s_hist[GL_HIST_WORDS];
    __shared__ unsigned s_q[QCAP];           // four packed levels per list entry
    __shared__ unsigned short s_pos[QCAP];   // (r << 8) | c per list entry
    __shared__ uint64_t s_rows[64];
    __shared__ unsigned short s_rowstart[64];

    const int lane = threadIdx.x;
    const int P = ps * ps;
    sweep(s_tab, GL_H128_OFF, 0u, lane);
    sweep(s_tab + GL_H128_OFF, SLOTS, GL_EMPTY, lane);
    sweep(s_hist, GL_HIST_WORDS, 0u, lane);
    __syncthreads();

    for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
        // ---- mask rows, row prefix, class test ------------------------------------------------------------
        const uint64_t myrow = lane < ps ? mask_rows[i * ps + lane] : 0ull;
        const int cnt = __popcll(myrow);
        int inc = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(inc, o, 64);
            if (lane >= o) inc += v;
        }
        const int K = __builtin_amdgcn_readlane(inc, 63);
        if (K <= k_lo || K > QCAP) continue; // another instantiation's nucleus (wave-uniform)
        const int excl = inc - cnt;
        s_rows[lane] = myrow;
        s_rowstart[lane] = (unsigned short)excl;
        const uint64_t anyrow = __ballot(myrow != 0ull);
        const T *pp = patches + (size_t)i * 3 * P;
        // u8 patches: the whole patch comes in with 16-byte loads (one HBM round trip) into the table memory,
        // which is idle until the list is built; the per-row loop below then reads LDS bytes
        const uint8_t *staged = nullptr;
        if (sizeof(T) == 1 && 3 * P <= TAB_WORDS * 4 && ((3 * P) & 15) == 0 && (reinterpret_cast<uintptr_t>(pp) & 15) == 0) {
            const uint4 *src = reinterpret_cast<const uint4 *>(pp);
            uint4 *dst = reinterpret_cast<uint4 *>(s_tab);
            for (int k = lane; k < (3 * P) / 16; k += 64) dst[k] = src[k];
            staged = reinterpret_cast<const uint8_t *>(s_tab);
            __syncthreads();
        }
        // ---- compact list: every masked pixel is read and quantised once --------------------------------------
        for (int r = 0; r < ps; ++r) {
            if (!((anyrow >> r) & 1ull)) continue;
            const uint64_t w = __shfl(myrow, r, 64);
            const int base = __builtin_amdgcn_readlane(excl, r);
            if (lane < ps && ((w >> lane) & 1ull)) {
                const float g = staged ? gray_at<uint8_t>(staged, P, r * ps + lane) : gray_at<T>(pp, P, r * ps + lane);
                const unsigned q4 = quant(g, 32) | (quant(g, 64) << 8) | (quant(g, 128) << 16) | (quant(g, 254) << 24); // texture.rs:19
                const int idx = base + __popcll(w & ((1ull << lane) - 1ull));
                s_q[idx] = q4;
                s_pos[idx] = (unsigned short)((r << 8) | lane);
            }
        }
        __syncthreads();
        if (staged) { // hand the table memory back: dense tables zero, hash table empty
            sweep(s_tab, GL_H128_OFF, 0u, lane);
            sweep(s_tab + GL_H128_OFF, SLOTS, GL_EMPTY, lane);
            __syncthreads();
        }
        double *sums_i = sums + i * 16 * GL_NACC;

        for (int oi = 0; oi < 4; ++oi) {
            const int dy = kGlcmOffsets[oi][0], dx = kGlcmOffsets[oi][1];
            // ================= pass A: levels 32, 64, 128 ==========================================================
            // histogram words: hx32 0..31 | hx64 32..95 | hx128 96..223 | hs32 224..286 | hs64 288..414 | hs128 416..670
            LevelAcc A0 = {0, 0, 0, 0, 0, 0.0f, 0.0}, A1 = A0, A2 = A0, A3 = A0;
            unsigned np = 0;
            for (int e = lane; e < K; e += 64) {
                const unsigned pos = s_pos[e];
                const int r2 = (int)(pos >> 8) + dy, c2 = (int)(pos & 255u) + dx;
                if (r2 >= ps || c2 < 0 || c2 >= ps) continue;
                const uint64_t w2 = s_rows[r2];
                if (!((w2 >> c2) & 1ull)) continue;
                const unsigned qa = s_q[e];
                const unsigned qb = s_q[s_rowstart[r2] + (unsigned)__popcll(w2 & ((1ull << c2) - 1ull))];
                np += 1u;
                {
                    const unsigned a = qa & 255u, b = qb & 255u;
                    pair_sums(a, b, dense_rank(s_tab, tri_index(a, b)), lut, s_hist, s_hist + 224, A0);
                }
                {
                    const unsigned a = (qa >> 8) & 255u, b = (qb >> 8) & 255u;
                    pair_sums(a, b, dense_rank(s_tab + GL_D64_OFF, tri_index(a, b)), lut, s_hist + 32, s_hist + 288, A1);
                }
                {
                    const unsigned a = (qa >> 16) & 255u, b = (qb >> 16) & 255u;
                    pair_sums(a, b, hash_rank<LOG2SLOTS>(s_tab + GL_H128_OFF, tri_index(a, b)), lut, s_hist + 96, s_hist + 416, A2);
                }
            }
            __syncthreads();
            const unsigned np_w = wave_total_u32(np);
            {
                const int hx_end[3] = {32, 96, 224}, hs_end[3] = {288, 416, GL_HIST_WORDS};
                double hxs[3], hss[3];
                hist_fold<3>(s_hist, lut, lane, hx_end, hs_end, hxs, hss);
                sweep(s_tab, SLOTS, GL_EMPTY, lane); // pass B's hash table lives at offset 0 (over d32, d64 and the head of h128)
                publish(sums_i + (0 * 4 + oi) * GL_NACC, np_w, A0, hxs[0], hss[0], lane);
                publish(sums_i + (1 * 4 + oi) * GL_NACC, np_w, A1, hxs[1], hss[1], lane);
                publish(sums_i + (2 * 4 + oi) * GL_NACC, np_w, A2, hxs[2], hss[2], lane);
            }
            __syncthreads();
            // ================= pass B: level 254 ===================================================================
            // histogram words: hx 0..253 | hs 256..762
            for (int e = lane; e < K; e += 64) {
                const unsigned pos = s_pos[e];
                const int r2 = (int)(pos >> 8) + dy, c2 = (int)(pos & 255u) + dx;
                if (r2 >= ps || c2 < 0 || c2 >= ps) continue;
                const uint64_t w2 = s_rows[r2];
                if (!((w2 >> c2) & 1ull)) continue;
                const unsigned a = s_q[e] >> 24;
                const unsigned b = s_q[s_rowstart[r2] + (unsigned)__popcll(w2 & ((1ull << c2) - 1ull))] >> 24;
                pair_sums(a, b, hash_rank<LOG2SLOTS>(s_tab, tri_index(a, b)), lut, s_hist, s_hist + 256, A3);
            }
            __syncthreads();
            {
                const int hx_end[1] = {256}, hs_end[1] = {GL_HIST_WORDS};
                double hxs[1], hss[1];
                hist_fold<1>(s_hist, lut, lane, hx_end, hs_end, hxs, hss);
                sweep(s_tab, GL_H128_OFF, 0u, lane);                 // pass A's dense tables start from zero
                sweep(s_tab + GL_H128_OFF, SLOTS, GL_EMPTY, lane);   // and its hash table from empty
                publish(sums_i + (3 * 4 + oi) * GL_NACC, np_w, A3, hxs[0], hss[0], lane);
            }
            __syncthreads();
        }
    }
}

// One lane per (nucleus, level, offset): the 14 Haralick features from the ten sums.
__global__ __launch_bounds__(256) void k_glcm_finish(int64_t n, const double *__restrict__ sums, float *__restrict__ out, int F, int col0)
{
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= n * 16) return;
    const int64_t i = g >> 4;
    const int comb_idx = (int)(g & 15);
    const double *a = sums + g * GL_NACC;
    const double N = a[0], S1 = a[1], S2 = a[2], D1 = a[3], D2 = a[4], IDM = a[5], ASMN = a[6], ENTS = a[7], HXS = a[8], HSS = a[9];
    float *o = out + i * (int64_t)F + col0 + comb_idx * 14;
    if (N == 0.0) {
        const float qnan = __builtin_nanf("");
#pragma unroll
        for (int k = 0; k < 14; ++k) o[k] = qnan;
        return;
    }
    const double T2 = 2.0 * N;
    const double mu = S1 / T2;
    const double ei2 = (S2 + D2) / (4.0 * N);
    const double var = ei2 - mu * mu;
    const double eij = (S2 - D2) / (4.0 * N);
    const double con = D2 / N, dis = D1 / N, sa = S1 / N;
    const double sv = S2 / N - sa * sa;
    const double lnT = log(T2);
    const double se = log(N) - HSS / N;
    const double ent = lnT - ENTS / T2;
    const double asm_ = ASMN / (T2 * T2);
    const double dv = D2 / N - dis * dis;
    const double hx = lnT - HXS / T2;
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

size_t glcm_scratch_bytes(const Job &job) { return sizeof(double) * 16 * GL_NACC * (size_t)(job.n > 0 ? job.n : 1); }

static hipError_t ensure_luts(nfx_ctx *ctx)
{
    if (ctx->glcm_lut) return hipSuccess;
    std::vector<double> h(3 * GL_RANKS);
    auto xl = [](double x) { return x > 0.0 ? x * std::log(x) : 0.0; };
    for (int k = 0; k < GL_RANKS; ++k) {
        h[k] = 2.0 * (xl(k + 1.0) - xl((double)k));                       // dO
        h[GL_RANKS + k] = xl(2.0 * (k + 1.0)) - xl(2.0 * (double)k);      // dD: H(x) = 2x ln 2x
        h[2 * GL_RANKS + k] = xl((double)k);                              // x ln x
    }
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&ctx->glcm_lut), sizeof(double) * h.size());
    if (e != hipSuccess) return e;
    return hipMemcpy(ctx->glcm_lut, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
}

template <typename T>
static void launch_classes(nfx_ctx *ctx, const Job &job, const GlcmLuts *lut, double *sums)
{
    const T *p = reinterpret_cast<const T *>(job.patches);
    const int P = job.ps * job.ps;
    auto grid = [&](int wgs_per_cu) { const int64_t m = 256ll * wgs_per_cu * 4; return (unsigned)(job.n < m ? job.n : m); };
    // K = 0 (empty mask) belongs to the first class: k_lo = -1.  Smaller classes keep more nuclei in flight per CU
    // (16 KB of LDS per wave for masks up to 512 pixels, 23 KB up to 1024, 37 KB up to 2048, 66 KB up to 4096).
    hipLaunchKernelGGL((k_glcm<T, 10>), dim3(grid(10)), dim3(64), 0, ctx->stream, job.n, job.ps, -1, p, job.mask_rows, lut, sums);
    if (P > 512) hipLaunchKernelGGL((k_glcm<T, 11>), dim3(grid(7)), dim3(64), 0, ctx->stream, job.n, job.ps, 512, p, job.mask_rows, lut, sums);
    if (P > 1024) hipLaunchKernelGGL((k_glcm<T, 12>), dim3(grid(4)), dim3(64), 0, ctx->stream, job.n, job.ps, 1024, p, job.mask_rows, lut, sums);
    if (P > 2048) hipLaunchKernelGGL((k_glcm<T, 13>), dim3(grid(2)), dim3(64), 0, ctx->stream, job.n, job.ps, 2048, p, job.mask_rows, lut, sums);
}

hipError_t launch_glcm(nfx_ctx *ctx, const Job &job, int col0, void *scratch)
{
    if (job.n == 0) return hipSuccess;
    hipError_t e = ensure_luts(ctx);
    if (e != hipSuccess) return e;
    ScopedTiming t(ctx, KS_GLCM);
    double *sums = reinterpret_cast<double *>(scratch);
    const GlcmLuts *lut = reinterpret_cast<const GlcmLuts *>(ctx->glcm_lut);
    if (job.patches_f32) launch_classes<float>(ctx, job, lut, sums);
    else launch_classes<uint8_t>(ctx, job, lut, sums);
    const int64_t items = job.n * 16;
    hipLaunchKernelGGL(k_glcm_finish, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, ctx->stream, job.n, sums, job.out, job.F, col0);
    return hipGetLastError();
}

}  // namespace nfx
