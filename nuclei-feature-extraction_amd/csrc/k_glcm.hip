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
//   * a persistent 256-thread workgroup owns one nucleus at a time.  The mask's pixels are compacted
//     once into a list (prefix sum of the row popcounts); each thread keeps its list entries'
//     coordinates and quantised levels (one byte per level, packed) in registers for the whole nucleus;
//   * the symmetric matrix is a TRIANGULAR table of packed 16-bit counters in LDS.  One pass of
//     RETURNING ds_add per pair is enough: the value the atomic returns is the pair's rank k inside
//     its cell, and
//         sum_cells G^2     = sum_pairs 2(2k+1)            (4(2k+1) on the diagonal)
//         sum_cells G ln G  = sum_pairs 2[h(k+1)-h(k)],  h(x) = x ln x   (diag: H(x) = 2x ln 2x)
//     telescope over the ranks.  The rank-0 pair OWNS its cell: after the barrier it reads the final
//     count m, adds m to the marginal / sum histograms (one weighted update per CELL instead of one per
//     pair), adds the tail f(m)-f(128) when the cell outgrew the rank tables, and clears the counter --
//     the table is never swept;
//   * two kernel variants share this code: SMALL handles levels 32, 64 and 128 in one pass over the pairs
//     of an offset (21 KB of tables -> four workgroups per CU), L254 handles level 254 (64.8 KB table ->
//     two workgroups per CU).  Per offset: atomics | barrier | owners | barrier, and the histogram
//     entropies + wave reductions of an offset run together with the next offset's atomics;
//   * every other feature is a closed form of four exact integer power sums over the pairs
//     (sum s, s^2, d, d^2 with s=a+b, d=|a-b|) plus the two 1-D histogram entropies.
// Instantiations per list-length class (entries per thread) keep the common small-mask case lean.  The ten sums of
// every (level, offset) combination go to a scratch array; k_glcm_finish evaluates the 14 features with one lane
// per combination (the f64 log/exp/divide chains would otherwise serialise 4-12 lanes of a workgroup per nucleus).
#include <cstdlib>

#include "nfx_internal.h"
#include "wave_ops.h"

namespace nfx {

static constexpr int GL_THREADS = 256;
static constexpr int GL_MAXK = (NFX_MAX_PATCH_SIZE * NFX_MAX_PATCH_SIZE) / GL_THREADS; // list entries per thread
static constexpr int GL_NACC = 10;
static constexpr int GL_LUT = 128;

__constant__ int kGlcmOffsets[4][2] = {{0, 1}, {1, 1}, {1, 0}, {1, -1}}; // texture.rs:20 (dy,dx)

static __device__ __forceinline__ double xlnx(unsigned m) { return m > 1u ? (double)m * log((double)m) : 0.0; }
static __device__ __forceinline__ double x2ln2x(unsigned m) { return m ? (double)(2u * m) * log((double)(2u * m)) : 0.0; }

struct LevelAcc {
    unsigned S1, S2, D1, D2, asmn;
    float idm;
    double ents;
};

// variant descriptors -----------------------------------------------------------------------------------
struct Small { // levels 32, 64, 128 (texture.rs:19), global level indices 0..2
    static constexpr int NL = 3;
    static constexpr int TAB_WORDS = 5456;               // 264 + 8 | 1040 + 8 | 4128 (+pad)
    static constexpr int HIST_WORDS = 768;               // hx32 0 | hx64 32 | hx128 96 | hs32 224 | hs64 288 | hs128 416
    static constexpr int WPS = 3;                        // waves per SIMD requested (46 KB of LDS per workgroup)
    static __device__ __forceinline__ int L(int l) { return 32 << l; }
    static __device__ __forceinline__ int tab_off(int l) { return l == 0 ? 0 : (l == 1 ? 272 : 1320); }
    static __device__ __forceinline__ int hx_off(int l) { return l == 0 ? 0 : (l == 1 ? 32 : 96); }
    static __device__ __forceinline__ int hs_off(int l) { return l == 0 ? 224 : (l == 1 ? 288 : 416); }
    static __device__ __forceinline__ int level_index(int l) { return l; }
    // which level a histogram word belongs to, and whether it is a marginal (hx) word
    static __device__ __forceinline__ int bin_level(int b, bool &is_hx)
    {
        is_hx = b < 224;
        if (b < 32) return 0;
        if (b < 96) return 1;
        if (b < 224) return 2;
        if (b < 288) return 0;
        if (b < 416) return 1;
        return 2;
    }
};
struct L254 { // level 254, global level index 3
    static constexpr int NL = 1;
    static constexpr int TAB_WORDS = (254 * 255 / 2 + 1) / 2 + 1;
    static constexpr int HIST_WORDS = 768;               // hx 0..253 | hs 256..762
    static constexpr int WPS = 2;
    static __device__ __forceinline__ int L(int) { return 254; }
    static __device__ __forceinline__ int tab_off(int) { return 0; }
    static __device__ __forceinline__ int hx_off(int) { return 0; }
    static __device__ __forceinline__ int hs_off(int) { return 256; }
    static __device__ __forceinline__ int level_index(int) { return 3; }
    static __device__ __forceinline__ int bin_level(int b, bool &is_hx) { is_hx = b < 256; return 0; }
};

static __device__ __forceinline__ unsigned quant(float gray, int L)
{
    int q = (int)floorf(gray * (float)L);
    q = q < 0 ? 0 : (q > L - 1 ? L - 1 : q);
    return (unsigned)q;
}

template <typename V>
static __device__ __forceinline__ unsigned quant_pack(float gray)
{
    unsigned p = 0;
#pragma unroll
    for (int l = 0; l < V::NL; ++l) p |= quant(gray, V::L(l)) << (8 * l);
    return p;
}

template <typename T>
static __device__ __forceinline__ float gray_at(const T *pp, int P, int p, const float *u8lut)
{
    float gs;
    if (sizeof(T) == 1) {
        const uint8_t *pb = reinterpret_cast<const uint8_t *>(pp);
        gs = u8lut[pb[p]] + u8lut[pb[P + p]];
        gs = gs + u8lut[pb[2 * P + p]];
    } else {
        const float *pf = reinterpret_cast<const float *>(pp);
        gs = pf[p] + pf[P + p];
        gs = gs + pf[2 * P + p];
    }
    return gs / 3.0f; // texture.rs:36
}

struct GlcmShared {
    unsigned *tab, *hist;
    unsigned *ovf; // [0] = count, [1..] = cells that outgrew the rank tables: m | diag << 16 | level << 17
    double *part;
    uint8_t *qmap; // [NL][ps*ps] quantised level of every masked pixel, written once per nucleus
    const double *dO, *dD;
};

// histogram entropies of the finished offset, wave totals of its accumulators -> per-wave partials
template <typename V>
static __device__ __forceinline__ void glcm_finish_offset(const GlcmShared &S, const LevelAcc (&A)[V::NL], unsigned np_w, int t)
{
    const int lane = t & 63, wv = t >> 6;
    double hxs[V::NL], hss[V::NL], tail[V::NL];
#pragma unroll
    for (int l = 0; l < V::NL; ++l) { hxs[l] = 0.0; hss[l] = 0.0; tail[l] = 0.0; }
    // cells with more than GL_LUT pairs: add f(m) - f(GL_LUT), the part the rank tables did not telescope
    {
        const unsigned n_ovf = S.ovf[0];
#pragma unroll 1
        for (unsigned e = t; e < n_ovf; e += GL_THREADS) {
            const unsigned rec = S.ovf[1 + e];
            const unsigned m = rec & 0xFFFFu;
            const int lv = (int)(rec >> 17);
            const double v = ((rec >> 16) & 1u) ? x2ln2x(m) - x2ln2x(GL_LUT) : 2.0 * (xlnx(m) - xlnx(GL_LUT));
#pragma unroll
            for (int l = 0; l < V::NL; ++l)
                if (l == lv) tail[l] += v;
        }
    }
#pragma unroll 1
    for (int j = 0; j < V::HIST_WORDS / GL_THREADS; ++j) {
        const int b = t + GL_THREADS * j;
        const unsigned v = S.hist[b];
        if (v > 1u) {
            bool is_hx;
            const int lv = V::bin_level(b, is_hx);
            const double e = xlnx(v);
#pragma unroll
            for (int l = 0; l < V::NL; ++l)
                if (l == lv) { if (is_hx) hxs[l] += e; else hss[l] += e; }
        }
        S.hist[b] = 0u;
    }
#pragma unroll
    for (int l = 0; l < V::NL; ++l) {
        const double v1 = (double)wave_total_u32(A[l].S1), v2 = (double)wave_total_u32(A[l].S2);
        const double v3 = (double)wave_total_u32(A[l].D1), v4 = (double)wave_total_u32(A[l].D2);
        const double v5 = (double)wave_total_f32(A[l].idm), v6 = (double)wave_total_u32(A[l].asmn);
        const double v7 = wave_total_f64(A[l].ents + tail[l]), v8 = wave_total_f64(hxs[l]), v9 = wave_total_f64(hss[l]);
        if (lane == 0) {
            double *pd = S.part + (wv * V::NL + l) * GL_NACC;
            pd[0] = (double)np_w; pd[1] = v1; pd[2] = v2; pd[3] = v3; pd[4] = v4; pd[5] = v5; pd[6] = v6; pd[7] = v7; pd[8] = v8; pd[9] = v9;
        }
    }
    if (t == 0) S.ovf[0] = 0u; // the next owners phase starts after a barrier
}

// All passes of one nucleus whose mask holds K <= MAXK*256 pixels; entries live in registers.
template <typename T, typename V, int MAXK>
static __device__ __forceinline__ void glcm_body(const GlcmShared &S, const unsigned short *list, const float *u8lut,
                                                 const uint64_t *rows, double *__restrict__ sums_i, const T *pp, int ps, int K, int t, int dbg)
{
    const int P = ps * ps;
    int rc[MAXK];
    unsigned qa[MAXK];
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
        rc[k] = -1; qa[k] = 0u;
        const int e = t + k * GL_THREADS;
        if (e < K) {
            const int pos = list[e];
            rc[k] = pos;
            const int p = (pos >> 8) * ps + (pos & 255);
            qa[k] = quant_pack<V>(gray_at<T>(pp, P, p, u8lut));
#pragma unroll
            for (int l = 0; l < V::NL; ++l) S.qmap[l * P + p] = (uint8_t)(qa[k] >> (8 * l));
        }
    }
    __syncthreads(); // level map complete; the list is in registers: its bytes become histograms + scratch
    // packed levels of every entry's four neighbours (0xFFFFFFFF: no in-mask neighbour) from the LDS level map
    unsigned qn[4][MAXK];
#pragma unroll
    for (int oi = 0; oi < 4; ++oi) {
        const int dy = kGlcmOffsets[oi][0], dx = kGlcmOffsets[oi][1];
#pragma unroll
        for (int k = 0; k < MAXK; ++k) {
            qn[oi][k] = 0xFFFFFFFFu;
            if (rc[k] >= 0) {
                const int r2 = (rc[k] >> 8) + dy, c2 = (rc[k] & 255) + dx;
                if (r2 < ps && c2 >= 0 && c2 < ps && ((rows[r2] >> c2) & 1ull)) {
                    unsigned q = 0;
#pragma unroll
                    for (int l = 0; l < V::NL; ++l) q |= (unsigned)S.qmap[l * P + r2 * ps + c2] << (8 * l);
                    qn[oi][k] = q;
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < V::HIST_WORDS / GL_THREADS; ++j) S.hist[t + GL_THREADS * j] = 0u;
    if (t == 0) S.ovf[0] = 0u;
    // (the barrier after the first offset's atomics orders this clear before the first histogram update)

    LevelAcc prev[V::NL];
    unsigned prev_np = 0;
#pragma unroll
    for (int oi = 0; oi < ((dbg & 2) ? 0 : 4); ++oi) {
        unsigned qb[MAXK];
        unsigned np = 0;
#pragma unroll
        for (int k = 0; k < MAXK; ++k) {
            qb[k] = qn[oi][k];
            np += qb[k] != 0xFFFFFFFFu ? 1u : 0u;
        }
        const unsigned np_w = wave_total_u32(np);
        // the previous offset's histograms are complete (barrier after its owners): fold them now
        if (oi > 0 && !(dbg & 8)) glcm_finish_offset<V>(S, prev, prev_np, t);

        // ---- phase 1: one returning atomic per pair and level ---------------------------------------------
        LevelAcc A[V::NL];
#pragma unroll
        for (int l = 0; l < V::NL; ++l) A[l] = LevelAcc{0, 0, 0, 0, 0, 0.0f, 0.0};
        unsigned long long own = 0ull; // bit NL*k+l: entry k holds rank 0 of its level-l cell
#pragma unroll
        for (int k = 0; k < MAXK; ++k) {
            if (qb[k] != 0xFFFFFFFFu && !(dbg & 16)) {
#pragma unroll
                for (int l = 0; l < V::NL; ++l) {
                    const unsigned a = (qa[k] >> (8 * l)) & 255u, b = (qb[k] >> (8 * l)) & 255u;
                    const unsigned lo = a < b ? a : b, hi = a < b ? b : a;
                    const unsigned idx = hi * (hi + 1u) / 2u + lo;
                    const unsigned sh = (idx & 1u) * 16u;
                    const unsigned rk = (atomicAdd(&S.tab[V::tab_off(l) + (idx >> 1)], 1u << sh) >> sh) & 0xFFFFu;
                    const unsigned s = a + b, d = hi - lo;
                    A[l].S1 += s; A[l].S2 += s * s; A[l].D1 += d; A[l].D2 += d * d;
                    A[l].idm += __builtin_amdgcn_rcpf(1.0f + (float)(d * d)); // 1/(1+d^2), 1 ulp
                    const bool diag = lo == hi;
                    A[l].asmn += (diag ? 4u : 2u) * (2u * rk + 1u);
                    if (rk < (unsigned)GL_LUT) A[l].ents += diag ? S.dD[rk] : S.dO[rk];
                    if (rk == 0u) own |= 1ull << (V::NL * k + l);
                }
            }
        }
        __syncthreads();
        // the previous offset's per-wave partials are published: sum them into the combination slot
        if (oi > 0 && t < V::NL * GL_NACC) {
            const int l = t / GL_NACC, a = t - l * GL_NACC;
            double v = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += S.part[(w * V::NL + l) * GL_NACC + a];
            sums_i[(V::level_index(l) * 4 + (oi - 1)) * GL_NACC + a] = v;
        }
        // ---- phase 2: cell owners ------------------------------------------------------------------------------
#pragma unroll
        for (int k = 0; k < MAXK; ++k) {
#pragma unroll
            for (int l = 0; l < V::NL; ++l) {
                if (((own >> (V::NL * k + l)) & 1ull) && !(dbg & 4)) {
                    const unsigned a = (qa[k] >> (8 * l)) & 255u, b = (qb[k] >> (8 * l)) & 255u;
                    const unsigned lo = a < b ? a : b, hi = a < b ? b : a;
                    unsigned short *cell = reinterpret_cast<unsigned short *>(S.tab + V::tab_off(l)) + (hi * (hi + 1u) / 2u + lo);
                    const unsigned m = *cell;
                    *cell = 0;
                    atomicAdd(&S.hist[V::hx_off(l) + a], m);
                    atomicAdd(&S.hist[V::hx_off(l) + b], m);
                    atomicAdd(&S.hist[V::hs_off(l) + a + b], m);
                    if (m > (unsigned)GL_LUT) {
                        const unsigned slot = atomicAdd(&S.ovf[0], 1u);
                        S.ovf[1 + slot] = m | ((lo == hi ? 1u : 0u) << 16) | ((unsigned)l << 17);
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int l = 0; l < V::NL; ++l) prev[l] = A[l];
        prev_np = np_w;
    }
    glcm_finish_offset<V>(S, prev, prev_np, t);
    __syncthreads();
    if (t < V::NL * GL_NACC) {
        const int l = t / GL_NACC, a = t - l * GL_NACC;
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) v += S.part[(w * V::NL + l) * GL_NACC + a];
        sums_i[(V::level_index(l) * 4 + 3) * GL_NACC + a] = v;
    }
    __syncthreads();
}

// One instantiation per variant and list-length class; a launch handles the nuclei with
// k_lo < K <= MAXK*256 and skips the others.
template <typename T, typename V, int MAXK>
__global__ __launch_bounds__(GL_THREADS, V::WPS) void k_glcm(int64_t n, int ps, int k_lo, const T *__restrict__ patches,
                                                             const uint64_t *__restrict__ mask_rows, double *__restrict__ sums, int dbg)
{
    __shared__ unsigned s_tab[V::TAB_WORDS];
    // s_list (pixel list, 8 KB) is dead once every thread holds its entries in registers; the histograms
    // and the reduction scratch live in the same bytes afterwards.
    __shared__ __attribute__((aligned(16))) unsigned char s_alias[NFX_MAX_PATCH_SIZE * NFX_MAX_PATCH_SIZE * 2];
    __shared__ float s_u8[256];       // u8 -> u8/255.0f (utils.rs:172)
    __shared__ double s_dO[GL_LUT];   // 2[h(k+1)-h(k)]
    __shared__ double s_dD[GL_LUT];   // H(k+1)-H(k)
    __shared__ uint64_t s_rows[64];
    __shared__ unsigned s_rowstart[65];
    __shared__ uint8_t s_q[V::NL * NFX_MAX_PATCH_SIZE * NFX_MAX_PATCH_SIZE]; // level map, one byte plane per level

    unsigned short *s_list = reinterpret_cast<unsigned short *>(s_alias);
    GlcmShared S;
    S.tab = s_tab;
    S.hist = reinterpret_cast<unsigned *>(s_alias);                       // V::HIST_WORDS words
    S.part = reinterpret_cast<double *>(s_alias + V::HIST_WORDS * 4);     // [4 waves][NL][GL_NACC]: 960 B
    S.ovf = reinterpret_cast<unsigned *>(s_alias + V::HIST_WORDS * 4 + 1024); // 1 + up to 4032/128 * NL records
    S.qmap = s_q; S.dO = s_dO; S.dD = s_dD;

    const int t = threadIdx.x, lane = t & 63;
    const int P = ps * ps;
    for (int w = t; w < V::TAB_WORDS; w += GL_THREADS) s_tab[w] = 0u;
    s_u8[t] = (float)t / 255.0f;
    if (t < GL_LUT) {
        s_dO[t] = 2.0 * (xlnx(t + 1) - xlnx(t));
        s_dD[t] = x2ln2x(t + 1) - x2ln2x(t);
    }
    // pixel p = t + 256k  ->  (r,c) by incremental carry, no division in the loops
    const int r_first = t / ps, c_first = t - r_first * ps;
    const int dr = GL_THREADS / ps, dc = GL_THREADS - dr * ps;
    __syncthreads();

    for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
        // ---- compact the mask's pixels into a list ----------------------------------------------------------
        if (t < 64) {
            const uint64_t w = t < ps ? mask_rows[i * ps + t] : 0ull;
            s_rows[t] = w;
            unsigned inc = (unsigned)__popcll(w); // inclusive scan over the 64 rows
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned v = __shfl_up(inc, o, 64);
                if (lane >= o) inc += v;
            }
            s_rowstart[t + 1] = inc;
            if (t == 0) s_rowstart[0] = 0u;
        }
        __syncthreads();
        const int K = (int)s_rowstart[64];
        if (K <= k_lo || K > MAXK * GL_THREADS) { // another instantiation's nucleus (uniform)
            __syncthreads();
            continue;
        }
        {
            int r = r_first, c = c_first;
#pragma unroll 1 // unrolled, the 16 (r,c) pairs and their masks become loop invariants of the nucleus loop: ~100 VGPRs
            for (int k = 0; k < GL_MAXK; ++k) {
                if (k * GL_THREADS >= P) break;
                if (r < ps) {
                    const uint64_t w = s_rows[r];
                    if ((w >> c) & 1ull)
                        s_list[s_rowstart[r] + (unsigned)__popcll(w & ((1ull << c) - 1ull))] = (unsigned short)((r << 8) | c);
                }
                c += dc; r += dr;
                if (c >= ps) { c -= ps; r += 1; }
            }
        }
        __syncthreads();
        const T *pp = patches + (size_t)i * 3 * P;
        if (!(dbg & 1)) glcm_body<T, V, MAXK>(S, s_list, s_u8, s_rows, sums + i * 16 * GL_NACC, pp, ps, K, t, dbg);

        // (the ten sums of every combination went straight to `sums`; k_glcm_finish turns them into the features)
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

template <typename T, typename V>
static void launch_variant(nfx_ctx *ctx, const Job &job, double *sums, unsigned grid)
{
    const T *p = reinterpret_cast<const T *>(job.patches);
    const char *dbg_env = getenv("NFX_GLCM_DBG"); // timing experiments only (phase skipping)
    const int dbg = dbg_env ? atoi(dbg_env) : 0;
    // K = 0 (empty mask) belongs to the first class: k_lo = -1
    hipLaunchKernelGGL((k_glcm<T, V, 4>), dim3(grid), dim3(GL_THREADS), 0, ctx->stream, job.n, job.ps, -1, p, job.mask_rows, sums, dbg);
    if (job.ps * job.ps > 4 * GL_THREADS)
        hipLaunchKernelGGL((k_glcm<T, V, 8>), dim3(grid), dim3(GL_THREADS), 0, ctx->stream, job.n, job.ps, 4 * GL_THREADS, p, job.mask_rows, sums, dbg);
    if (job.ps * job.ps > 8 * GL_THREADS)
        hipLaunchKernelGGL((k_glcm<T, V, GL_MAXK>), dim3(grid), dim3(GL_THREADS), 0, ctx->stream, job.n, job.ps, 8 * GL_THREADS, p, job.mask_rows, sums, dbg);
}

hipError_t launch_glcm(nfx_ctx *ctx, const Job &job, int col0, void *scratch)
{
    if (job.n == 0) return hipSuccess;
    ScopedTiming t(ctx, KS_GLCM);
    const int64_t maxb_small = 256 * 3 * 2, maxb_254 = 256 * 2 * 2; // resident workgroups per CU x2 for tail balance
    const unsigned grid_s = (unsigned)(job.n < maxb_small ? job.n : maxb_small);
    const unsigned grid_l = (unsigned)(job.n < maxb_254 ? job.n : maxb_254);
    double *sums = reinterpret_cast<double *>(scratch);
    if (job.patches_f32) {
        launch_variant<float, Small>(ctx, job, sums, grid_s);
        launch_variant<float, L254>(ctx, job, sums, grid_l);
    } else {
        launch_variant<uint8_t, Small>(ctx, job, sums, grid_s);
        launch_variant<uint8_t, L254>(ctx, job, sums, grid_l);
    }
    const int64_t items = job.n * 16;
    hipLaunchKernelGGL(k_glcm_finish, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, ctx->stream, job.n, sums, job.out, job.F, col0);
    return hipGetLastError();
}

}  // namespace nfx
