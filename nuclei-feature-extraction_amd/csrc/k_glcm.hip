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
//   * a persistent 256-thread workgroup owns one nucleus at a time.  The mask's pixels are
//     compacted once into a list (prefix sum of the row popcounts); each thread keeps its list
//     entries' coordinates and gray values in registers for the whole nucleus;
//   * the symmetric matrix is a TRIANGULAR table of packed 16-bit counters in LDS (64.8 KB at
//     L=254 -> two workgroups per CU).  One pass of RETURNING ds_add per pair is enough: the value
//     the atomic returns is the pair's rank k inside its cell, and
//         sum_cells G^2     = sum_pairs 2(2k+1)            (4(2k+1) on the diagonal)
//         sum_cells G ln G  = sum_pairs 2[h(k+1)-h(k)],  h(x) = x ln x   (diag: H(x) = 2x ln 2x)
//     telescope over the ranks, so the multiplicities are never read back;
//   * the three small levels (32, 64, 128) share one pass over the pairs of an offset; level 254 takes
//     a second one in the same table memory.  A pass is: atomics | barrier | histogram entropies +
//     the rank-0 pair of every cell clears its counter | barrier;
//   * every other feature is a closed form of four exact integer power sums over the pairs
//     (sum s, s^2, d, d^2 with s=a+b, d=|a-b|) plus two 1-D histogram entropies;
//   * wave reductions are DPP (row_shr / row_bcast), not LDS permutes.
// The 14 features of the 16 (level, offset) combinations are evaluated by 16 lanes at the end.
#include "nfx_internal.h"

namespace nfx {

static constexpr int GL_THREADS = 256;
static constexpr int GL_MAXK = (NFX_MAX_PATCH_SIZE * NFX_MAX_PATCH_SIZE) / GL_THREADS; // list entries per thread
static constexpr int GL_TAB_WORDS = (254 * 255 / 2 + 1) / 2 + 1;                       // packed u16 triangular, L=254
static constexpr int GL_NACC = 10;
static constexpr int GL_LUT = 128;

__constant__ int kGlcmOffsets[4][2] = {{0, 1}, {1, 1}, {1, 0}, {1, -1}}; // texture.rs:20 (dy,dx)

// ---- DPP wave reductions (gfx9 row_shr / row_bcast); the total is valid in lane 63 ---------------
template <int CTRL, int ROW_MASK>
static __device__ __forceinline__ int dpp_mov(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, false);
}
static __device__ __forceinline__ unsigned wave_total_u32(unsigned x)
{
    int v = (int)x;
    v += dpp_mov<0xB1, 0xF>(v);  // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E, 0xF>(v);  // quad_perm [2,3,0,1]
    v += dpp_mov<0x141, 0xF>(v); // row_half_mirror
    v += dpp_mov<0x140, 0xF>(v); // row_mirror        -> every lane holds its row's sum
    v += dpp_mov<0x142, 0xA>(v); // row_bcast15 into rows 1,3
    v += dpp_mov<0x143, 0xC>(v); // row_bcast31 into rows 2,3 -> lanes 48..63 hold the total
    return (unsigned)__builtin_amdgcn_readlane(v, 63);
}
template <int CTRL, int ROW_MASK>
static __device__ __forceinline__ double dpp_mov_f64(double v)
{
    const int lo = dpp_mov<CTRL, ROW_MASK>(__double2loint(v));
    const int hi = dpp_mov<CTRL, ROW_MASK>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
static __device__ __forceinline__ double wave_total_f64(double v)
{
    v += dpp_mov_f64<0xB1, 0xF>(v);
    v += dpp_mov_f64<0x4E, 0xF>(v);
    v += dpp_mov_f64<0x141, 0xF>(v);
    v += dpp_mov_f64<0x140, 0xF>(v);
    v += dpp_mov_f64<0x142, 0xA>(v);
    v += dpp_mov_f64<0x143, 0xC>(v);
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

static __device__ __forceinline__ double xlnx(unsigned m) { return m > 1u ? (double)m * log((double)m) : 0.0; }
static __device__ __forceinline__ double x2ln2x(unsigned m) { return m ? (double)(2u * m) * log((double)(2u * m)) : 0.0; }

// per-level accumulators of one pass
struct LevelAcc {
    unsigned S1, S2, D1, D2, asmn;
    double idm, ents;
};

// one pair (a,b) at level L: returning atomic on the packed triangular table + histograms.
// Ranks below GL_LUT telescope through the tables.  Returns true for the pair of rank 0: that pair OWNS the
// cell -- after the barrier it reads the final count m, adds the remainder f(m) - f(GL_LUT) if the cell
// outgrew the tables, and clears the counter (glcm_owner).  Nobody else touches the cell after the barrier.
static __device__ __forceinline__ bool glcm_pair(unsigned a, unsigned b, unsigned *tab, unsigned *hx, unsigned *hs,
                                                 const double *dO, const double *dD, const float *idm_lut, LevelAcc &A)
{
    const unsigned lo = a < b ? a : b, hi = a < b ? b : a;
    const unsigned idx = hi * (hi + 1u) / 2u + lo;
    const unsigned sh = (idx & 1u) * 16u;
    const unsigned k = (atomicAdd(&tab[idx >> 1], 1u << sh) >> sh) & 0xFFFFu; // rank of this pair in its cell
    atomicAdd(&hx[a], 1u);
    atomicAdd(&hx[b], 1u);
    atomicAdd(&hs[a + b], 1u);
    const unsigned s = a + b, d = hi - lo;
    A.S1 += s; A.S2 += s * s; A.D1 += d; A.D2 += d * d;
    A.idm += (double)idm_lut[d];
    const bool diag = lo == hi;
    A.asmn += (diag ? 4u : 2u) * (2u * k + 1u);
    if (k < (unsigned)GL_LUT) A.ents += diag ? dD[k] : dO[k];
    return k == 0u;
}

// after the barrier, by the cell's owner only
static __device__ __forceinline__ void glcm_owner(unsigned a, unsigned b, unsigned *tab, LevelAcc &A)
{
    const unsigned lo = a < b ? a : b, hi = a < b ? b : a;
    unsigned short *cell = reinterpret_cast<unsigned short *>(tab) + (hi * (hi + 1u) / 2u + lo);
    const unsigned m = *cell;
    *cell = 0;
    if (m > (unsigned)GL_LUT) A.ents += lo == hi ? x2ln2x(m) - x2ln2x(GL_LUT) : 2.0 * (xlnx(m) - xlnx(GL_LUT));
}

static __device__ __forceinline__ unsigned quant(float gray, int L)
{
    int q = (int)floorf(gray * (float)L);
    q = q < 0 ? 0 : (q > L - 1 ? L - 1 : q);
    return (unsigned)q;
}

struct GlcmShared {
    unsigned *tab, *hist;
    double *part;
    const unsigned short *list;
    const float *u8, *idm;
    const double *dO, *dD;
    const uint64_t *rows;
    double (*comb)[GL_NACC];
};

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

// All passes of one nucleus whose mask holds K <= MAXK*256 pixels; entries live in registers.
template <typename T, int MAXK>
static __device__ __forceinline__ void glcm_body(const GlcmShared &S, const T *pp, int ps, int K, int t)
{
    const int lane = t & 63, wv = t >> 6;
    const int P = ps * ps;
    int rc[MAXK];
    float ga[MAXK];
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
        rc[k] = -1; ga[k] = 0.0f;
        const int e = t + k * GL_THREADS;
        if (e < K) {
            const int pos = S.list[e];
            rc[k] = pos;
            ga[k] = gray_at<T>(pp, P, (pos >> 8) * ps + (pos & 255), S.u8);
        }
    }
    __syncthreads(); // the list is in registers: its bytes become histograms + scratch
    S.hist[t] = 0u; S.hist[t + 256] = 0u; S.hist[t + 512] = 0u;
    __syncthreads();

    for (int oi = 0; oi < 4; ++oi) {
        const int dy = kGlcmOffsets[oi][0], dx = kGlcmOffsets[oi][1];
        // neighbour gray of every entry for this offset (NaN when it has no in-mask neighbour)
        float gb[MAXK];
        unsigned np = 0;
#pragma unroll
        for (int k = 0; k < MAXK; ++k) {
            gb[k] = __builtin_nanf("");
            if (rc[k] >= 0) {
                const int r2 = (rc[k] >> 8) + dy, c2 = (rc[k] & 255) + dx;
                if (r2 < ps && c2 >= 0 && c2 < ps && ((S.rows[r2] >> c2) & 1ull)) {
                    gb[k] = gray_at<T>(pp, P, r2 * ps + c2, S.u8);
                    np += 1u;
                }
            }
        }
        const unsigned np_w = wave_total_u32(np);

        // ================= pass A: levels 32, 64, 128 =====================================================
        {
            // table sub-ranges (words): L=32 at 0 (264 w), L=64 at 272 (1040 w), L=128 at 1320 (4128 w)
            unsigned *tab0 = S.tab, *tab1 = S.tab + 272, *tab2 = S.tab + 1320;
            // histogram words: hx32 0..31 | hx64 32..95 | hx128 96..223 | hs32 224..286 | hs64 288..414 | hs128 416..670
            unsigned *hx0 = S.hist, *hx1 = S.hist + 32, *hx2 = S.hist + 96;
            unsigned *hs0 = S.hist + 224, *hs1 = S.hist + 288, *hs2 = S.hist + 416;
            LevelAcc A0 = {0, 0, 0, 0, 0, 0.0, 0.0}, A1 = A0, A2 = A0;
            unsigned long long tail = 0ull; // bit 3k+l: entry k owns its level-l cell
#pragma unroll
            for (int k = 0; k < MAXK; ++k) {
                if (gb[k] == gb[k]) {
                    if (glcm_pair(quant(ga[k], 32), quant(gb[k], 32), tab0, hx0, hs0, S.dO, S.dD, S.idm, A0)) tail |= 1ull << (3 * k);
                    if (glcm_pair(quant(ga[k], 64), quant(gb[k], 64), tab1, hx1, hs1, S.dO, S.dD, S.idm, A1)) tail |= 2ull << (3 * k);
                    if (glcm_pair(quant(ga[k], 128), quant(gb[k], 128), tab2, hx2, hs2, S.dO, S.dD, S.idm, A2)) tail |= 4ull << (3 * k);
                }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < MAXK; ++k) {
                if ((tail >> (3 * k)) & 1ull) glcm_owner(quant(ga[k], 32), quant(gb[k], 32), tab0, A0);
                if ((tail >> (3 * k)) & 2ull) glcm_owner(quant(ga[k], 64), quant(gb[k], 64), tab1, A1);
                if ((tail >> (3 * k)) & 4ull) glcm_owner(quant(ga[k], 128), quant(gb[k], 128), tab2, A2);
            }
            // histogram entropies: bin t, t+256, t+512 of the 671 words; then clear
            double hxs0 = 0, hxs1 = 0, hxs2 = 0, hss0 = 0, hss1 = 0, hss2 = 0;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int b = t + 256 * j;
                const unsigned v = S.hist[b];
                if (v > 1u) {
                    const double e = xlnx(v);
                    if (b < 32) hxs0 += e; else if (b < 96) hxs1 += e; else if (b < 224) hxs2 += e;
                    else if (b < 288) hss0 += e; else if (b < 416) hss1 += e; else hss2 += e;
                }
                S.hist[b] = 0u;
            }
            const LevelAcc *AA[3] = {&A0, &A1, &A2};
            const double hx3[3] = {hxs0, hxs1, hxs2}, hs3[3] = {hss0, hss1, hss2};
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                const double v0 = (double)np_w;
                const double v1 = (double)wave_total_u32(AA[l]->S1), v2 = (double)wave_total_u32(AA[l]->S2);
                const double v3 = (double)wave_total_u32(AA[l]->D1), v4 = (double)wave_total_u32(AA[l]->D2);
                const double v5 = wave_total_f64(AA[l]->idm), v6 = (double)wave_total_u32(AA[l]->asmn);
                const double v7 = wave_total_f64(AA[l]->ents), v8 = wave_total_f64(hx3[l]), v9 = wave_total_f64(hs3[l]);
                if (lane == 0) {
                    double *pd = S.part + (wv * 3 + l) * GL_NACC;
                    pd[0] = v0; pd[1] = v1; pd[2] = v2; pd[3] = v3; pd[4] = v4; pd[5] = v5; pd[6] = v6; pd[7] = v7; pd[8] = v8; pd[9] = v9;
                }
            }
            __syncthreads();
            if (t < 3 * GL_NACC) {
                const int l = t / GL_NACC, a = t - l * GL_NACC;
                S.comb[l * 4 + oi][a] = S.part[(0 * 3 + l) * GL_NACC + a] + S.part[(1 * 3 + l) * GL_NACC + a] +
                                        S.part[(2 * 3 + l) * GL_NACC + a] + S.part[(3 * 3 + l) * GL_NACC + a];
            }
        }
        // ================= pass B: level 254 ===============================================================
        {
            unsigned *hx = S.hist, *hs = S.hist + 256; // hx 0..253 | hs 256..762
            LevelAcc A = {0, 0, 0, 0, 0, 0.0, 0.0};
            // no barrier needed here: pass A cleared the histograms before its last barrier, and the scratch is
            // rewritten only after this pass's own post-atomics barrier
            unsigned tail = 0u;
#pragma unroll
            for (int k = 0; k < MAXK; ++k)
                if (gb[k] == gb[k] && glcm_pair(quant(ga[k], 254), quant(gb[k], 254), S.tab, hx, hs, S.dO, S.dD, S.idm, A)) tail |= 1u << k;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < MAXK; ++k)
                if ((tail >> k) & 1u) glcm_owner(quant(ga[k], 254), quant(gb[k], 254), S.tab, A);
            double hxs = 0, hss = 0;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int b = t + 256 * j;
                const unsigned v = S.hist[b];
                if (v > 1u) { if (b < 256) hxs += xlnx(v); else hss += xlnx(v); }
                S.hist[b] = 0u;
            }
            const double v0 = (double)np_w;
            const double v1 = (double)wave_total_u32(A.S1), v2 = (double)wave_total_u32(A.S2);
            const double v3 = (double)wave_total_u32(A.D1), v4 = (double)wave_total_u32(A.D2);
            const double v5 = wave_total_f64(A.idm), v6 = (double)wave_total_u32(A.asmn);
            const double v7 = wave_total_f64(A.ents), v8 = wave_total_f64(hxs), v9 = wave_total_f64(hss);
            if (lane == 0) {
                double *pd = S.part + wv * GL_NACC;
                pd[0] = v0; pd[1] = v1; pd[2] = v2; pd[3] = v3; pd[4] = v4; pd[5] = v5; pd[6] = v6; pd[7] = v7; pd[8] = v8; pd[9] = v9;
            }
            __syncthreads();
            if (t < GL_NACC)
                S.comb[3 * 4 + oi][t] = S.part[t] + S.part[GL_NACC + t] + S.part[2 * GL_NACC + t] + S.part[3 * GL_NACC + t];
        }
    }
    __syncthreads();
}

// One instantiation per list-length class (MAXK entries per thread) so the common small-mask case gets its own
// register allocation; a launch handles the nuclei with k_lo < K <= MAXK*256 and skips the others.
template <typename T, int MAXK>
__global__ __launch_bounds__(GL_THREADS, 2) void k_glcm(int64_t n, int ps, int k_lo, const T *__restrict__ patches,
                                                        const uint64_t *__restrict__ mask_rows, float *__restrict__ out,
                                                        int F, int col0)
{
    __shared__ unsigned s_tab[GL_TAB_WORDS];
    // s_list (pixel list, 8 KB) is dead once every thread holds its entries in registers; the histograms
    // and the reduction scratch live in the same bytes afterwards.
    __shared__ __attribute__((aligned(16))) unsigned char s_alias[NFX_MAX_PATCH_SIZE * NFX_MAX_PATCH_SIZE * 2];
    __shared__ float s_u8[256];       // u8 -> u8/255.0f (utils.rs:172)
    __shared__ float s_idm[256];      // 1/(1+d^2)
    __shared__ double s_dO[GL_LUT];   // 2[h(k+1)-h(k)]
    __shared__ double s_dD[GL_LUT];   // H(k+1)-H(k)
    __shared__ uint64_t s_rows[64];
    __shared__ unsigned s_rowstart[65];
    __shared__ double s_comb[16][GL_NACC];

    unsigned short *s_list = reinterpret_cast<unsigned short *>(s_alias);
    GlcmShared S;
    S.tab = s_tab;
    S.hist = reinterpret_cast<unsigned *>(s_alias);            // 768 words: hx | hs per level
    S.part = reinterpret_cast<double *>(s_alias + 768 * 4);    // [4 waves][3 levels][GL_NACC]
    S.list = s_list; S.u8 = s_u8; S.idm = s_idm; S.dO = s_dO; S.dD = s_dD; S.rows = s_rows; S.comb = s_comb;

    const int t = threadIdx.x, lane = t & 63;
    const int P = ps * ps;
    for (int w = t; w < GL_TAB_WORDS; w += GL_THREADS) s_tab[w] = 0u;
    s_u8[t] = (float)t / 255.0f;
    s_idm[t] = 1.0f / (1.0f + (float)t * (float)t);
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
#pragma unroll
            for (int k = 0; k < GL_MAXK; ++k) {
                if (k * GL_THREADS < P && r < ps) {
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
        glcm_body<T, MAXK>(S, pp, ps, K, t);

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
        }
        // no trailing barrier: the next nucleus touches s_comb / the scratch only several barriers later
    }
}

template <typename T>
static void launch_glcm_t(nfx_ctx *ctx, const Job &job, int col0, unsigned grid)
{
    const T *p = reinterpret_cast<const T *>(job.patches);
    // K = 0 (empty mask) belongs to the first class: k_lo = -1
    hipLaunchKernelGGL((k_glcm<T, 4>), dim3(grid), dim3(GL_THREADS), 0, ctx->stream, job.n, job.ps, -1, p, job.mask_rows, job.out, job.F, col0);
    if (job.ps * job.ps > 4 * GL_THREADS)
        hipLaunchKernelGGL((k_glcm<T, 8>), dim3(grid), dim3(GL_THREADS), 0, ctx->stream, job.n, job.ps, 4 * GL_THREADS, p, job.mask_rows, job.out, job.F, col0);
    if (job.ps * job.ps > 8 * GL_THREADS)
        hipLaunchKernelGGL((k_glcm<T, GL_MAXK>), dim3(grid), dim3(GL_THREADS), 0, ctx->stream, job.n, job.ps, 8 * GL_THREADS, p, job.mask_rows, job.out, job.F, col0);
}

hipError_t launch_glcm(nfx_ctx *ctx, const Job &job, int col0)
{
    if (job.n == 0) return hipSuccess;
    ScopedTiming t(ctx, KS_GLCM);
    const int64_t maxb = 256 * 2 * 2; // two resident workgroups per CU, x2 for tail balance
    const unsigned grid = (unsigned)(job.n < maxb ? job.n : maxb);
    if (job.patches_f32) launch_glcm_t<float>(ctx, job, col0, grid);
    else launch_glcm_t<uint8_t>(ctx, job, col0, grid);
    return hipGetLastError();
}

}  // namespace nfx
