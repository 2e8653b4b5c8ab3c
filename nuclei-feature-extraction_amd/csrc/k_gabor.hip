// k_gabor.hip -- the "gabor" feature set: 8 angles x 6 frequencies, masked mean / variance of the response.
//
// Replaces (reference file:line):
//   GaborFilterFeatureSet::compute_features_batched  texture.rs:323-369 (ANGLES_COUNT, FREQUENCIES 319-320)
//   tch_utils::gabor::apply_gabor_filter              call site texture.rs:333-334 (conv2d 1 -> 48 channels)
//
// The reference convolves every patch with 48 dense 30x30 kernels (354 MFLOP per 64x64 patch) and
// then keeps only the masked pixels.  The Gabor kernel with an isotropic envelope factorises:
//     G(x)G(y) cos(Ax + By) = [G(x)cos Ax][G(y)cos By] - [G(x)sin Ax][G(y)sin By]
// so each filter is two 30-tap row passes (shared by every filter with the same |A|: 13 distinct
// row passes serve the whole bank) and one 30-tap column pass evaluated ONLY at masked pixels.
// Angles t and t+180 degrees give the same kernel (cos is even), so 24 column passes fill the 48
// channels.  About 3 MFLOP per nucleus instead of 354.
//
// One 256-thread workgroup per nucleus (persistent): gray patch with a zero apron, and the two
// row-pass planes U, W (with zero aprons, odd row stride) live in LDS (78 KB, two workgroups per CU).
// Both passes are REGISTER-TILED and WAVE-STRIPED: wave w owns strip w (8, 12 or 16 outputs wide, chosen per
// nucleus from the mask's bounding box so that four strips cover it); a lane loads a (TS+29)-value window once
// and produces TS outputs, the 30 taps arrive as LDS broadcasts from a double-buffered per-group table.  Every
// pass is exactly one round for every wave.  Row passes cover only the mask's column range and the rows within
// 15 of the mask; the 45/135-degree filter pairs share one column pass (f+- = P1 -+ P2); per-strip sums are f32
// about a per-filter shift (DC gain x mean masked gray), accumulated in f64 and reduced with DPP.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "exact_div.h"
#include "nfx_internal.h"
#include "wave_ops.h"

namespace nfx {

static constexpr int GB_SIZE = 30, GB_PAD = 15, GB_THREADS = 256;
static constexpr int GB_GW = NFX_MAX_PATCH_SIZE + GB_SIZE - 1; // gray row with apron: 93 floats
static constexpr int GB_UW = NFX_MAX_PATCH_SIZE + 1;           // U/W row stride (odd: conflict-free both ways)
static constexpr int GB_UH = NFX_MAX_PATCH_SIZE + GB_SIZE - 1; // U/W rows incl. zero aprons: 93
static constexpr int GB_MAX_GROUPS = 16, GB_MAX_FILTERS = 48;
static constexpr int GB_WSLOT = 32;                            // one 30-tap kernel padded to 32 floats
static constexpr int GB_WBUF = GB_WSLOT * 14;                  // rc, rs + up to 6 x (cc, cs)

struct GaborGroup {
    float rc[GB_SIZE];
    float rs[GB_SIZE];
    int has_w, first_filter, n_filters, paired;
    float image[GB_WBUF]; // the group's taps exactly as they sit in LDS: [rc | rs | cc0 | cs0 | cc1 | cs1 | ...] x 32 floats
};
struct GaborFilter {
    float cc[GB_SIZE];
    float cs[GB_SIZE];
    float wsign;
    int ch0, ch1, has_s;
    float dc; // sum of the 2-D kernel: response to a constant image of 1
};
struct GaborTables {
    int n_groups, n_filters, pad0, pad1;
    GaborGroup groups[GB_MAX_GROUPS];
    GaborFilter filters[GB_MAX_FILTERS];
};

template <typename T> struct BPx;
template <> struct BPx<uint8_t> {
    static __device__ __forceinline__ float ld(const uint8_t *p, size_t i) { return u8_unit(p[i]); }
};
template <> struct BPx<float> {
    static __device__ __forceinline__ float ld(const float *p, size_t i) { return p[i]; }
};

// ---- DPP wave reduction; the total is returned in every lane -------------------------------------------
template <int CTRL, int ROW_MASK>
static __device__ __forceinline__ double gb_dpp(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
static __device__ __forceinline__ double bsum(double v)
{
    v += gb_dpp<0xB1, 0xF>(v);
    v += gb_dpp<0x4E, 0xF>(v);
    v += gb_dpp<0x141, 0xF>(v);
    v += gb_dpp<0x140, 0xF>(v);
    v += gb_dpp<0x142, 0xA>(v);
    v += gb_dpp<0x143, 0xC>(v);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// TS outputs of a 30-tap correlation from a (TS+29)-value register window; taps come from LDS
// (every lane reads the same address: a broadcast, one LDS cycle per ds_read_b128)
template <int TS>
static __device__ __forceinline__ void taps(const float (&x)[TS + GB_SIZE - 1], const float *w, float (&acc)[TS])
{
    float wr[GB_WSLOT];
#pragma unroll
    for (int j = 0; j < GB_WSLOT; j += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(w + j);
        wr[j] = v.x; wr[j + 1] = v.y; wr[j + 2] = v.z; wr[j + 3] = v.w;
    }
#pragma unroll
    for (int o = 0; o < TS; ++o) acc[o] = 0.0f;
    // tap-major order, pinned: the TS FMAs of one tap hit TS different accumulators (independent, back-to-back
    // issue).  Left alone, the scheduler serialises all 30 taps of one accumulator into a dependent chain.
#pragma unroll
    for (int j = 0; j < GB_SIZE; ++j) {
#pragma unroll
        for (int o = 0; o < TS; ++o) acc[o] = fmaf(wr[j], x[o + j], acc[o]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// masked, shifted partial sums of one strip's TS outputs (f32 inside the strip, f64 across strips)
template <int TS>
static __device__ __forceinline__ void accumulate(const float (&v)[TS], unsigned mbits, float shift, double &s1, double &s2)
{
    float a = 0.0f, b = 0.0f;
#pragma unroll
    for (int q = 0; q < TS; ++q) {
        const float d = ((mbits >> q) & 1u) ? v[q] - shift : 0.0f;
        a += d;
        b = fmaf(d, d, b);
    }
    s1 += (double)a;
    s2 += (double)b;
}

struct GaborShared {
    float *g, *U, *W;
    const float *wt; // current group's taps: [rc | rs | cc0 | cs0 | cc1 | cs1 | ...] x 32 floats
    const uint64_t *rows;
};

// One group (one row pass + its column passes) with strip size TS: wave w owns column strip w of the row pass
// and row strip w of the column pass, so each pass is exactly one round for every wave.
template <int TS>
static __device__ __forceinline__ void gabor_group(const GaborShared &S, const GaborTables *__restrict__ tab, int g, int ps,
                                                   int gw, int rlo, int nrow_rp, int rmin, int rmax, int cmin, int cmax,
                                                   float mu_gray, int lane, int wv, double (&s1)[6], double (&s2)[6], int dbg)
{
    constexpr int WIN = TS + GB_SIZE - 1;
    const GaborGroup &G = tab->groups[g];
    const bool has_w = G.has_w != 0;
    // ---- row pass: lanes run along rows (LDS stride gw, odd), the wave's strip along columns ----------------
    {
        const int c0 = cmin + wv * TS;
        if (c0 <= cmax && lane < nrow_rp && !(dbg & 1)) {
            const int r = rlo + lane;
            int c0c = c0 < ps - TS ? c0 : ps - TS; // keep the window inside the apron row
            c0c = c0c > 0 ? c0c : 0;
            const float *src = &S.g[r * gw + c0c]; // gray column c-15 sits at apron index c
            float x[WIN], acc[TS];
#pragma unroll
            for (int j = 0; j < WIN; ++j) x[j] = src[j];
            taps<TS>(x, S.wt, acc);
            float *du = &S.U[(r + GB_PAD) * GB_UW + c0c];
            // unconditional stores (c0c <= ps - TS keeps them inside the row; for ps < TS they land in the row's unused
            // tail): a guarded store makes the compiler sink each output's 30 FMAs into its own branch -- one
            // dependent chain per output instead of TS independent ones
#pragma unroll
            for (int q = 0; q < TS; ++q) du[q] = acc[q];
            if (has_w) {
                taps<TS>(x, S.wt + GB_WSLOT, acc);
                float *dw = &S.W[(r + GB_PAD) * GB_UW + c0c];
#pragma unroll
                for (int q = 0; q < TS; ++q) dw[q] = acc[q];
            }
        }
    }
    __syncthreads();
    // ---- column pass: lanes run along columns, the wave's strip along rows ------------------------------------
    {
        const int r0 = rmin + wv * TS;
        const int c = cmin + lane;
        if (r0 <= rmax && c <= cmax && !(dbg & 2)) {
            int r0c = r0 < ps - TS ? r0 : ps - TS; // clamped strips overlap: rows below r0 are masked out
            r0c = r0c > 0 ? r0c : 0;
            unsigned mbits = 0;
#pragma unroll
            for (int q = 0; q < TS; ++q)
                if (r0c + q >= r0) mbits |= (unsigned)((S.rows[r0c + q] >> c) & 1ull) << q;
            if (mbits != 0u) {
                const int nf = G.n_filters;
                float x[WIN], acc[TS];
                const float *su = &S.U[r0c * GB_UW + c]; // row (r0c - 15) + 15
#pragma unroll
                for (int j = 0; j < WIN; ++j) x[j] = su[j * GB_UW];
                if (G.paired) { // two filters share cc, cs: f+ = P1 - P2, f- = P1 + P2
                    float p2[TS], v[TS];
                    taps<TS>(x, S.wt + 2 * GB_WSLOT, acc);
                    const float *sw = &S.W[r0c * GB_UW + c];
#pragma unroll
                    for (int j = 0; j < WIN; ++j) x[j] = sw[j * GB_UW];
                    taps<TS>(x, S.wt + 3 * GB_WSLOT, p2);
#pragma unroll
                    for (int q = 0; q < TS; ++q) v[q] = acc[q] - p2[q];
                    accumulate<TS>(v, mbits, tab->filters[G.first_filter].dc * mu_gray, s1[0], s2[0]);
#pragma unroll
                    for (int q = 0; q < TS; ++q) v[q] = acc[q] + p2[q];
                    accumulate<TS>(v, mbits, tab->filters[G.first_filter + 1].dc * mu_gray, s1[1], s2[1]);
                } else if (!has_w) { // A = 0: every filter of the group is a plain column pass over U
#pragma unroll
                    for (int f = 0; f < 6; ++f) {
                        if (f < nf) {
                            taps<TS>(x, S.wt + (2 + 2 * f) * GB_WSLOT, acc);
                            accumulate<TS>(acc, mbits, tab->filters[G.first_filter + f].dc * mu_gray, s1[f], s2[f]);
                        }
                    }
                } else { // general: out = U (*) cc - wsign * W (*) cs
                    float xw[WIN], p2[TS];
                    const float *sw = &S.W[r0c * GB_UW + c];
#pragma unroll
                    for (int j = 0; j < WIN; ++j) xw[j] = sw[j * GB_UW];
#pragma unroll
                    for (int f = 0; f < 6; ++f) {
                        if (f < nf) {
                            const GaborFilter &Fl = tab->filters[G.first_filter + f];
                            taps<TS>(x, S.wt + (2 + 2 * f) * GB_WSLOT, acc);
                            if (Fl.has_s) {
                                taps<TS>(xw, S.wt + (3 + 2 * f) * GB_WSLOT, p2);
                                const float ws = Fl.wsign;
#pragma unroll
                                for (int q = 0; q < TS; ++q) acc[q] = acc[q] - ws * p2[q];
                            }
                            accumulate<TS>(acc, mbits, Fl.dc * mu_gray, s1[f], s2[f]);
                        }
                    }
                }
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(GB_THREADS, 2) void k_gabor(int64_t n, int ps, const T *__restrict__ patches,
                                                         const uint64_t *__restrict__ mask_rows,
                                                         const GaborTables *__restrict__ tab, float *__restrict__ out,
                                                         int F, int col0, int dbg)
{
    __shared__ __attribute__((aligned(16))) float s_wt[2][GB_WBUF];
    __shared__ float s_g[NFX_MAX_PATCH_SIZE * GB_GW + 16];
    __shared__ float s_U[GB_UH * GB_UW]; // row r lives at index r + GB_PAD; aprons stay zero
    __shared__ float s_W[GB_UH * GB_UW];
    __shared__ uint64_t s_rows[64 + 16];
    __shared__ double s_part[4][24][2];
    __shared__ float s_gsum[4];

    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int P = ps * ps;
    const int gw = ps + GB_SIZE - 1;
    for (int k = t; k < NFX_MAX_PATCH_SIZE * GB_GW + 16; k += GB_THREADS) s_g[k] = 0.0f; // apron stays zero
    for (int k = t; k < GB_UH * GB_UW; k += GB_THREADS) { s_U[k] = 0.0f; s_W[k] = 0.0f; }
    for (int k = t; k < 2 * GB_WBUF; k += GB_THREADS) (&s_wt[0][0])[k] = 0.0f;
    if (t < 64 + 16) s_rows[t] = 0ull;
    __syncthreads();
    const int n_groups = tab->n_groups, n_filters = tab->n_filters;
    GaborShared S;
    S.g = s_g; S.U = s_U; S.W = s_W; S.rows = s_rows;

    for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
        if (t < 64) s_rows[t] = t < ps ? mask_rows[i * ps + t] : 0ull;
        __syncthreads();
        const T *pp = patches + (size_t)i * 3 * P;
        float gsum = 0.0f; // masked gray sum -> the shift that keeps the f32 partial sums well conditioned
        {
            int r = t / ps, c = t - r * ps;
            const int dr = GB_THREADS / ps, dc = GB_THREADS - dr * ps;
            for (int p = t; p < P; p += GB_THREADS) {
                float gs = BPx<T>::ld(pp, p) + BPx<T>::ld(pp, P + p);
                gs = gs + BPx<T>::ld(pp, 2 * P + p);
                gs = sizeof(T) == 1 ? third_of_u8_sum(gs) : gs / 3.0f; // texture.rs:332
                s_g[r * gw + c + GB_PAD] = gs;
                if ((s_rows[r] >> c) & 1ull) gsum += gs;
                c += dc; r += dr;
                if (c >= ps) { c -= ps; r += 1; }
            }
        }
        gsum = (float)bsum((double)gsum);
        if (lane == 0) s_gsum[wv] = gsum;
        // mask bounding box and area: lane r looks at row r, the wave votes
        const uint64_t wrow = s_rows[lane];
        const uint64_t rowsany = __ballot(wrow != 0ull);
        const int area = wave_total_i32(__popcll(wrow));
        int olo = (int)(unsigned)wrow, ohi = (int)(unsigned)(wrow >> 32);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { olo |= __shfl_xor(olo, o, 64); ohi |= __shfl_xor(ohi, o, 64); }
        const uint64_t orr = ((uint64_t)(unsigned)ohi << 32) | (uint64_t)(unsigned)olo;
        const int rmin = rowsany ? __ffsll((long long)rowsany) - 1 : ps, rmax = rowsany ? 63 - __clzll((long long)rowsany) : -1;
        float *o = out + i * (int64_t)F + col0;
        if (area == 0) { // 0/0 in texture.rs:340-344
            if (t < 96) o[t] = __builtin_nanf("");
            __syncthreads();
            continue;
        }
        const int cmin = __ffsll((long long)orr) - 1, cmax = 63 - __clzll((long long)orr);
        const int ncol = cmax - cmin + 1, nrow = rmax - rmin + 1;
        const int rlo = rmin - GB_PAD > 0 ? rmin - GB_PAD : 0;
        const int rhi = rmax + (GB_SIZE - 1 - GB_PAD) < ps - 1 ? rmax + (GB_SIZE - 1 - GB_PAD) : ps - 1;
        const int nrow_rp = rhi - rlo + 1;
        const int span = ncol > nrow ? ncol : nrow;   // both passes use the same strip size
        // group 0's taps -> buffer 0 (the barrier inside the first gabor_group call is too late for the row pass,
        // so this load is followed by its own barrier, which also publishes s_g and s_gsum)
        for (int g = 0; g < ((dbg & 4) ? 0 : n_groups); ++g) {
            const GaborGroup &G = tab->groups[g];
            float *wt = s_wt[g & 1];
            if (t < GB_WBUF / 4) reinterpret_cast<float4 *>(wt)[t] = reinterpret_cast<const float4 *>(G.image)[t];
            __syncthreads(); // taps visible; every wave has left the previous group's column pass (U/W reusable)
            const float mu_gray = (s_gsum[0] + s_gsum[1] + s_gsum[2] + s_gsum[3]) / (float)area;
            S.wt = wt;
            double s1[6], s2[6];
#pragma unroll
            for (int f = 0; f < 6; ++f) { s1[f] = 0.0; s2[f] = 0.0; }
            if (span <= 32) gabor_group<8>(S, tab, g, ps, gw, rlo, nrow_rp, rmin, rmax, cmin, cmax, mu_gray, lane, wv, s1, s2, dbg);
            else if (span <= 48) gabor_group<12>(S, tab, g, ps, gw, rlo, nrow_rp, rmin, rmax, cmin, cmax, mu_gray, lane, wv, s1, s2, dbg);
            else gabor_group<16>(S, tab, g, ps, gw, rlo, nrow_rp, rmin, rmax, cmin, cmax, mu_gray, lane, wv, s1, s2, dbg);
#pragma unroll
            for (int f = 0; f < 6; ++f) {
                if (f < G.n_filters) {
                    // each lane holds one strip's shifted sums (small numbers): an f32 wave reduction is exact enough
                    const float a = wave_total_f32((float)s1[f]), b = wave_total_f32((float)s2[f]);
                    if (lane == 0) { s_part[wv][G.first_filter + f][0] = (double)a; s_part[wv][G.first_filter + f][1] = (double)b; }
                }
            }
        }
        __syncthreads();
        // ---- mean / variance (texture.rs:339-344) from the shifted sums: mean = c + s1/K, var = s2/K - (s1/K)^2 -----
        if (t < n_filters) {
            const GaborFilter &Fl = tab->filters[t];
            const double K = (double)area;
            const float mu_gray = (s_gsum[0] + s_gsum[1] + s_gsum[2] + s_gsum[3]) / (float)area;
            const double shift = (double)(Fl.dc * mu_gray);
            const double a = s_part[0][t][0] + s_part[1][t][0] + s_part[2][t][0] + s_part[3][t][0];
            const double b = s_part[0][t][1] + s_part[1][t][1] + s_part[2][t][1] + s_part[3][t][1];
            const double m1 = a / K;
            double var = b / K - m1 * m1;
            var = var > 0.0 ? var : 0.0;
            const double mean = shift + m1;
            if (Fl.ch0 >= 0) { o[2 * Fl.ch0] = (float)mean; o[2 * Fl.ch0 + 1] = (float)var; }
            if (Fl.ch1 >= 0) { o[2 * Fl.ch1] = (float)mean; o[2 * Fl.ch1 + 1] = (float)var; }
        }
        __syncthreads();
    }
}

// ---- host: factorised filter bank (DESIGN.md "Gabor bank") ----------------------------------------------
static void build_tables(GaborTables &T)
{
    const double PI = 3.14159265358979323846, sigma = 0.45;
    const double freqs[6] = {0.5, 1.0, 2.0, 4.0, 6.0, 8.0}; // texture.rs:320
    memset(&T, 0, sizeof(T));
    struct Item { double A, B; int ch; };
    std::vector<Item> items;
    for (int a = 0; a < 8; ++a)
        for (int f = 0; f < 6; ++f) {
            const double th = (double)a * PI / 4.0;
            double A = 2.0 * PI * freqs[f] * cos(th), B = 2.0 * PI * freqs[f] * sin(th);
            if (fabs(A) < 1e-9) A = 0.0;
            if (fabs(B) < 1e-9) B = 0.0;
            items.push_back({A, B, a * 6 + f});
        }
    auto tk = [](int k) { return -1.0 + 2.0 * (double)k / (double)(GB_SIZE - 1); };
    auto env = [&](double x) { return exp(-(x * x) / (2.0 * sigma * sigma)); };
    auto close = [](double x, double y) { return fabs(x - y) < 1e-9; };
    std::vector<char> used(items.size(), 0);
    // unique filters: cos(Ax+By) == cos(-Ax-By)
    struct Uniq { double A, B; int ch0, ch1; };
    std::vector<Uniq> uniq;
    for (size_t i = 0; i < items.size(); ++i) {
        if (used[i]) continue;
        Uniq u{items[i].A, items[i].B, items[i].ch, -1};
        used[i] = 1;
        for (size_t j = i + 1; j < items.size(); ++j)
            if (!used[j] && ((close(items[j].A, u.A) && close(items[j].B, u.B)) || (close(items[j].A, -u.A) && close(items[j].B, -u.B)))) {
                u.ch1 = items[j].ch;
                used[j] = 1;
                break;
            }
        uniq.push_back(u);
    }
    // groups by |A|
    std::vector<double> gA;
    for (auto &u : uniq) {
        bool found = false;
        for (double a : gA) found = found || close(a, fabs(u.A));
        if (!found) gA.push_back(fabs(u.A));
    }
    T.n_groups = (int)gA.size();
    int nf = 0;
    for (int g = 0; g < T.n_groups; ++g) {
        GaborGroup &G = T.groups[g];
        for (int j = 0; j < GB_SIZE; ++j) {
            G.rc[j] = (float)(env(tk(j)) * cos(gA[g] * tk(j)));
            G.rs[j] = (float)(env(tk(j)) * sin(gA[g] * tk(j)));
        }
        G.has_w = gA[g] != 0.0;
        G.first_filter = nf;
        for (auto &u : uniq) {
            if (!close(fabs(u.A), gA[g])) continue;
            GaborFilter &Fl = T.filters[nf++];
            for (int k = 0; k < GB_SIZE; ++k) {
                Fl.cc[k] = (float)(env(tk(k)) * cos(u.B * tk(k)));
                Fl.cs[k] = (float)(env(tk(k)) * sin(u.B * tk(k)));
            }
            Fl.wsign = u.A < 0.0 ? -1.0f : 1.0f; // sin(Ax) = sign(A) sin(|A|x)
            {   // DC gain: sum_ij G(x_j)G(y_i) cos(A x_j + B y_i)
                double dc = 0.0;
                for (int ii = 0; ii < GB_SIZE; ++ii)
                    for (int jj = 0; jj < GB_SIZE; ++jj) dc += env(tk(ii)) * env(tk(jj)) * cos(u.A * tk(jj) + u.B * tk(ii));
                Fl.dc = (float)dc;
            }
            Fl.ch0 = u.ch0; Fl.ch1 = u.ch1;
            Fl.has_s = u.B != 0.0;
        }
        G.n_filters = nf - G.first_filter;
        for (int j = 0; j < GB_SIZE; ++j) { G.image[j] = G.rc[j]; G.image[GB_WSLOT + j] = G.rs[j]; }
        for (int f = 0; f < G.n_filters && f < 6; ++f)
            for (int j = 0; j < GB_SIZE; ++j) {
                G.image[(2 + 2 * f) * GB_WSLOT + j] = T.filters[G.first_filter + f].cc[j];
                G.image[(3 + 2 * f) * GB_WSLOT + j] = T.filters[G.first_filter + f].cs[j];
            }
        // two filters with the same column kernels and opposite sin sign (45 / 135 degrees) share one pass
        G.paired = 0;
        if (G.n_filters == 2 && G.has_w) {
            const GaborFilter &f0 = T.filters[G.first_filter], &f1 = T.filters[G.first_filter + 1];
            bool same = f0.has_s && f1.has_s && f0.wsign == 1.0f && f1.wsign == -1.0f;
            for (int k = 0; k < GB_SIZE && same; ++k) same = f0.cc[k] == f1.cc[k] && f0.cs[k] == f1.cs[k];
            G.paired = same ? 1 : 0;
        }
    }
    T.n_filters = nf;
}

static hipError_t ensure_tables(nfx_ctx *ctx)
{
    if (ctx->gabor_tab) return hipSuccess;
    static GaborTables T;
    build_tables(T);
    if (T.n_groups > GB_MAX_GROUPS || T.n_filters > 24) return hipErrorInvalidValue;
    for (int g = 0; g < T.n_groups; ++g)
        if (T.groups[g].n_filters > 6) return hipErrorInvalidValue;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&ctx->gabor_tab), sizeof(GaborTables));
    if (e != hipSuccess) return e;
    return hipMemcpy(ctx->gabor_tab, &T, sizeof(GaborTables), hipMemcpyHostToDevice);
}

hipError_t launch_gabor(nfx_ctx *ctx, const Job &job, int col0)
{
    if (job.n == 0) return hipSuccess;
    hipError_t e = ensure_tables(ctx);
    if (e != hipSuccess) return e;
    ScopedTiming t(ctx, KS_GABOR);
    const int64_t maxb = 256 * 2 * 2;
    const unsigned grid = (unsigned)(job.n < maxb ? job.n : maxb);
    const GaborTables *tab = reinterpret_cast<const GaborTables *>(ctx->gabor_tab);
    const char *dbg_env = getenv("NFX_GABOR_DBG"); // timing experiments only: 1 skip row pass, 2 skip column pass, 4 skip all groups
    const int dbg = dbg_env ? atoi(dbg_env) : 0;
    if (job.patches_f32)
        hipLaunchKernelGGL(k_gabor<float>, dim3(grid), dim3(GB_THREADS), 0, ctx->stream, job.n, job.ps,
                           reinterpret_cast<const float *>(job.patches), job.mask_rows, tab, job.out, job.F, col0, dbg);
    else
        hipLaunchKernelGGL(k_gabor<uint8_t>, dim3(grid), dim3(GB_THREADS), 0, ctx->stream, job.n, job.ps,
                           reinterpret_cast<const uint8_t *>(job.patches), job.mask_rows, tab, job.out, job.F, col0, dbg);
    return hipGetLastError();
}

}  // namespace nfx
