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
// Both passes are REGISTER-TILED: a thread loads a 37-value window once and produces 8 outputs
// (480 FMAs per 37 LDS reads, filter taps in SGPRs); row passes cover only the mask's column range
// and the rows within 15 of the mask; the 45/135-degree filter pairs share one column pass
// (f+- = P1 -+ P2); per-filter sums are f64, reduced with DPP.
#include <cmath>
#include <cstring>
#include <vector>

#include "nfx_internal.h"

namespace nfx {

static constexpr int GB_SIZE = 30, GB_PAD = 15, GB_THREADS = 256;
static constexpr int GB_GW = NFX_MAX_PATCH_SIZE + GB_SIZE - 1; // gray row with apron: 93 floats
static constexpr int GB_TS = 8;                                // outputs per thread in a strip
static constexpr int GB_WIN = GB_TS + GB_SIZE - 1;             // 37 inputs feed 8 outputs
static constexpr int GB_UW = NFX_MAX_PATCH_SIZE + 1;           // U/W row stride (odd: conflict-free both ways)
static constexpr int GB_UH = NFX_MAX_PATCH_SIZE + GB_SIZE - 1 + GB_TS; // U/W rows incl. zero aprons: 101
static constexpr int GB_MAX_GROUPS = 16, GB_MAX_FILTERS = 48;

struct GaborGroup {
    float rc[GB_SIZE];
    float rs[GB_SIZE];
    int has_w, first_filter, n_filters, paired;
};
struct GaborFilter {
    float cc[GB_SIZE];
    float cs[GB_SIZE];
    float wsign;
    int ch0, ch1, has_s;
};
struct GaborTables {
    int n_groups, n_filters, pad0, pad1;
    GaborGroup groups[GB_MAX_GROUPS];
    GaborFilter filters[GB_MAX_FILTERS];
};

template <typename T> struct BPx;
template <> struct BPx<uint8_t> {
    static __device__ __forceinline__ float ld(const uint8_t *p, size_t i) { return (float)p[i] / 255.0f; }
};
template <> struct BPx<float> {
    static __device__ __forceinline__ float ld(const float *p, size_t i) { return p[i]; }
};

// ---- DPP wave reduction (total valid in every lane via readlane 63) ------------------------------------
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

// 8 outputs of a 30-tap correlation from a 37-value register window
static __device__ __forceinline__ void taps8(const float (&x)[GB_WIN], const float *__restrict__ w, float (&acc)[GB_TS])
{
#pragma unroll
    for (int o = 0; o < GB_TS; ++o) acc[o] = 0.0f;
#pragma unroll
    for (int j = 0; j < GB_SIZE; ++j) {
        const float wj = w[j]; // wave-uniform: scalar load
#pragma unroll
        for (int o = 0; o < GB_TS; ++o) acc[o] = fmaf(wj, x[o + j], acc[o]);
    }
}

template <typename T>
__global__ __launch_bounds__(GB_THREADS, 2) void k_gabor(int64_t n, int ps, const T *__restrict__ patches,
                                                         const uint64_t *__restrict__ mask_rows,
                                                         const GaborTables *__restrict__ tab, float *__restrict__ out,
                                                         int F, int col0)
{
    __shared__ float s_g[NFX_MAX_PATCH_SIZE * GB_GW + 16];
    __shared__ float s_U[GB_UH * GB_UW]; // row r lives at index r + GB_PAD; aprons stay zero
    __shared__ float s_W[GB_UH * GB_UW];
    __shared__ uint64_t s_rows[64 + GB_TS];
    __shared__ double s_part[4][24][2];

    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int P = ps * ps;
    const int gw = ps + GB_SIZE - 1;
    for (int k = t; k < NFX_MAX_PATCH_SIZE * GB_GW + 16; k += GB_THREADS) s_g[k] = 0.0f; // apron stays zero
    for (int k = t; k < GB_UH * GB_UW; k += GB_THREADS) { s_U[k] = 0.0f; s_W[k] = 0.0f; }
    if (t < 64 + GB_TS) s_rows[t] = 0ull;
    __syncthreads();
    const int n_groups = tab->n_groups, n_filters = tab->n_filters;

    for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
        if (t < 64) s_rows[t] = t < ps ? mask_rows[i * ps + t] : 0ull;
        const T *pp = patches + (size_t)i * 3 * P;
        for (int p = t; p < P; p += GB_THREADS) {
            const int r = p / ps, c = p - r * ps;
            float gs = BPx<T>::ld(pp, p) + BPx<T>::ld(pp, P + p);
            gs = gs + BPx<T>::ld(pp, 2 * P + p);
            s_g[r * gw + c + GB_PAD] = gs / 3.0f; // texture.rs:332
        }
        __syncthreads();
        // mask bounding box and area
        uint64_t orr = 0;
        int rmin = ps, rmax = -1, area = 0;
        for (int r = 0; r < ps; ++r) {
            const uint64_t w = s_rows[r];
            if (w) { orr |= w; rmin = r < rmin ? r : rmin; rmax = r; area += __popcll(w); }
        }
        float *o = out + i * (int64_t)F + col0;
        if (area == 0) { // 0/0 in texture.rs:340-344
            if (t < 96) o[t] = __builtin_nanf("");
            __syncthreads();
            continue;
        }
        const int cmin = __ffsll((long long)orr) - 1, cmax = 63 - __clzll((long long)orr);
        const int ncol = cmax - cmin + 1;
        const int rlo = rmin - GB_PAD > 0 ? rmin - GB_PAD : 0;
        const int rhi = rmax + (GB_SIZE - 1 - GB_PAD) < ps - 1 ? rmax + (GB_SIZE - 1 - GB_PAD) : ps - 1;
        const int nrow_rp = rhi - rlo + 1;
        const int nstrip_c = (ncol + GB_TS - 1) / GB_TS;              // column strips of the row pass
        const int nstrip_r = (rmax - rmin + 1 + GB_TS - 1) / GB_TS;   // row strips of the column pass

        for (int g = 0; g < n_groups; ++g) {
            const GaborGroup &G = tab->groups[g];
            const bool has_w = G.has_w != 0;
            // ---- row pass: item = (row, strip of 8 columns); lanes run along rows (LDS stride gw, odd) ------
            for (int k = t; k < nrow_rp * nstrip_c; k += GB_THREADS) {
                const int st = k / nrow_rp;
                const int r = rlo + (k - st * nrow_rp), c0 = cmin + st * GB_TS;
                const float *src = &s_g[r * gw + c0]; // gray column c-15 sits at apron index c
                float x[GB_WIN];
#pragma unroll
                for (int j = 0; j < GB_WIN; ++j) x[j] = src[j];
                float acc[GB_TS];
                taps8(x, G.rc, acc);
                float *du = &s_U[(r + GB_PAD) * GB_UW + c0];
#pragma unroll
                for (int q = 0; q < GB_TS; ++q)
                    if (c0 + q < ps) du[q] = acc[q];
                if (has_w) {
                    taps8(x, G.rs, acc);
                    float *dw = &s_W[(r + GB_PAD) * GB_UW + c0];
#pragma unroll
                    for (int q = 0; q < GB_TS; ++q)
                        if (c0 + q < ps) dw[q] = acc[q];
                }
            }
            __syncthreads();
            // ---- column pass: item = (column, strip of 8 rows); lanes run along columns ---------------------
            const int nf = G.n_filters;
            double s1[6], s2[6];
#pragma unroll
            for (int f = 0; f < 6; ++f) { s1[f] = 0.0; s2[f] = 0.0; }
            for (int k = t; k < ncol * nstrip_r; k += GB_THREADS) {
                const int st = k / ncol;
                const int c = cmin + (k - st * ncol), r0 = rmin + st * GB_TS;
                unsigned mbits = 0; // mask bits of the 8 output pixels
#pragma unroll
                for (int q = 0; q < GB_TS; ++q) mbits |= (unsigned)((s_rows[r0 + q] >> c) & 1ull) << q;
                if (mbits == 0u) continue;
                float x[GB_WIN], acc[GB_TS];
                const float *su = &s_U[r0 * GB_UW + c]; // row (r0 - 15) + 15
#pragma unroll
                for (int j = 0; j < GB_WIN; ++j) x[j] = su[j * GB_UW];
                if (G.paired) { // two filters share cc, cs: f+ = P1 - P2, f- = P1 + P2
                    const GaborFilter &Fl = tab->filters[G.first_filter];
                    float p1[GB_TS], p2[GB_TS];
                    taps8(x, Fl.cc, p1);
                    const float *sw = &s_W[r0 * GB_UW + c];
#pragma unroll
                    for (int j = 0; j < GB_WIN; ++j) x[j] = sw[j * GB_UW];
                    taps8(x, Fl.cs, p2);
#pragma unroll
                    for (int q = 0; q < GB_TS; ++q) {
                        if ((mbits >> q) & 1u) {
                            const double a = (double)(p1[q] - p2[q]), b = (double)(p1[q] + p2[q]);
                            s1[0] += a; s2[0] += a * a;
                            s1[1] += b; s2[1] += b * b;
                        }
                    }
                } else if (!has_w) { // A = 0: every filter of the group is a plain column pass over U
#pragma unroll
                    for (int f = 0; f < 6; ++f) {
                        if (f < nf) {
                            taps8(x, tab->filters[G.first_filter + f].cc, acc);
#pragma unroll
                            for (int q = 0; q < GB_TS; ++q)
                                if ((mbits >> q) & 1u) { const double a = (double)acc[q]; s1[f] += a; s2[f] += a * a; }
                        }
                    }
                } else { // general: out = U (*) cc - wsign * W (*) cs
                    float xw[GB_WIN];
                    const float *sw = &s_W[r0 * GB_UW + c];
#pragma unroll
                    for (int j = 0; j < GB_WIN; ++j) xw[j] = sw[j * GB_UW];
                    for (int f = 0; f < nf && f < 6; ++f) {
                        const GaborFilter &Fl = tab->filters[G.first_filter + f];
                        float p2[GB_TS];
                        taps8(x, Fl.cc, acc);
                        if (Fl.has_s) {
                            taps8(xw, Fl.cs, p2);
#pragma unroll
                            for (int q = 0; q < GB_TS; ++q) acc[q] = acc[q] - Fl.wsign * p2[q];
                        }
#pragma unroll
                        for (int q = 0; q < GB_TS; ++q)
                            if ((mbits >> q) & 1u) {
                                const double a = (double)acc[q];
                                if (f == 0) { s1[0] += a; s2[0] += a * a; } else if (f == 1) { s1[1] += a; s2[1] += a * a; }
                                else if (f == 2) { s1[2] += a; s2[2] += a * a; } else if (f == 3) { s1[3] += a; s2[3] += a * a; }
                                else if (f == 4) { s1[4] += a; s2[4] += a * a; } else { s1[5] += a; s2[5] += a * a; }
                            }
                    }
                }
            }
#pragma unroll
            for (int f = 0; f < 6; ++f) {
                if (f < nf) {
                    const double a = bsum(s1[f]), b = bsum(s2[f]);
                    if (lane == 0) { s_part[wv][G.first_filter + f][0] = a; s_part[wv][G.first_filter + f][1] = b; }
                }
            }
            __syncthreads();
        }
        // ---- mean / variance (texture.rs:339-344): var = sum f^2/K - mean^2 ---------------------------------
        if (t < n_filters) {
            const GaborFilter &Fl = tab->filters[t];
            const double K = (double)area;
            const double a = s_part[0][t][0] + s_part[1][t][0] + s_part[2][t][0] + s_part[3][t][0];
            const double b = s_part[0][t][1] + s_part[1][t][1] + s_part[2][t][1] + s_part[3][t][1];
            const double mean = a / K;
            double var = b / K - mean * mean;
            var = var > 0.0 ? var : 0.0;
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
            Fl.ch0 = u.ch0; Fl.ch1 = u.ch1;
            Fl.has_s = u.B != 0.0;
        }
        G.n_filters = nf - G.first_filter;
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
    if (job.patches_f32)
        hipLaunchKernelGGL(k_gabor<float>, dim3(grid), dim3(GB_THREADS), 0, ctx->stream, job.n, job.ps,
                           reinterpret_cast<const float *>(job.patches), job.mask_rows, tab, job.out, job.F, col0);
    else
        hipLaunchKernelGGL(k_gabor<uint8_t>, dim3(grid), dim3(GB_THREADS), 0, ctx->stream, job.n, job.ps,
                           reinterpret_cast<const uint8_t *>(job.patches), job.mask_rows, tab, job.out, job.F, col0);
    return hipGetLastError();
}

}  // namespace nfx
