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
// row-pass planes U, W live in LDS (57 KB, two workgroups per CU); row passes cover only the
// mask's column range and the rows within 15 of the mask; per-filter sums are f64.
#include <cmath>
#include <cstring>
#include <vector>

#include "nfx_internal.h"

namespace nfx {

static constexpr int GB_SIZE = 30, GB_PAD = 15, GB_THREADS = 256;
static constexpr int GB_GW = NFX_MAX_PATCH_SIZE + GB_SIZE - 1; // gray row with apron: 93 floats
static constexpr int GB_MAX_GROUPS = 16, GB_MAX_FILTERS = 48;

struct GaborGroup {
    float rc[GB_SIZE];
    float rs[GB_SIZE];
    int has_w, first_filter, n_filters, pad;
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

static __device__ __forceinline__ double bsum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <typename T>
__global__ __launch_bounds__(GB_THREADS, 2) void k_gabor(int64_t n, int ps, const T *__restrict__ patches,
                                                         const uint64_t *__restrict__ mask_rows,
                                                         const GaborTables *__restrict__ tab, float *__restrict__ out,
                                                         int F, int col0)
{
    __shared__ float s_g[NFX_MAX_PATCH_SIZE * GB_GW];
    __shared__ float s_U[NFX_MAX_PATCH_SIZE * NFX_MAX_PATCH_SIZE];
    __shared__ float s_W[NFX_MAX_PATCH_SIZE * NFX_MAX_PATCH_SIZE];
    __shared__ uint64_t s_rows[64];
    __shared__ double s_part[4][GB_MAX_FILTERS][2];

    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int P = ps * ps;
    const int gw = ps + GB_SIZE - 1;
    for (int k = t; k < NFX_MAX_PATCH_SIZE * GB_GW; k += GB_THREADS) s_g[k] = 0.0f; // apron stays zero
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
        const int nrow_rp = rhi - rlo + 1, nrow_cp = rmax - rmin + 1;

        for (int g = 0; g < n_groups; ++g) {
            const GaborGroup &G = tab->groups[g];
            const bool has_w = G.has_w != 0;
            // ---- row pass: U = gray (*) rc, W = gray (*) rs over rows [rlo,rhi], columns [cmin,cmax] ----
            for (int k = t; k < nrow_rp * ncol; k += GB_THREADS) {
                const int rr = k / ncol;
                const int r = rlo + rr, c = cmin + (k - rr * ncol);
                const float *src = &s_g[r * gw + c]; // gray column c-15 sits at apron index c
                float u = 0.0f, w = 0.0f;
                if (has_w) {
#pragma unroll
                    for (int j = 0; j < GB_SIZE; ++j) {
                        const float x = src[j];
                        u = fmaf(G.rc[j], x, u);
                        w = fmaf(G.rs[j], x, w);
                    }
                    s_W[r * ps + c] = w;
                } else {
#pragma unroll
                    for (int j = 0; j < GB_SIZE; ++j) u = fmaf(G.rc[j], src[j], u);
                }
                s_U[r * ps + c] = u;
            }
            __syncthreads();
            // ---- column passes at masked pixels ---------------------------------------------------------
            for (int fi = 0; fi < G.n_filters; ++fi) {
                const int f = G.first_filter + fi;
                const GaborFilter &Fl = tab->filters[f];
                const bool use_w = has_w && Fl.has_s != 0;
                const float ws = Fl.wsign;
                double s1 = 0.0, s2 = 0.0;
                for (int k = t; k < nrow_cp * ncol; k += GB_THREADS) {
                    const int rr = k / ncol;
                    const int r = rmin + rr, c = cmin + (k - rr * ncol);
                    if ((s_rows[r] >> c) & 1ull) {
                        const int i0 = r - GB_PAD < 0 ? GB_PAD - r : 0;                       // first tap with r+i-15 >= 0
                        const int i1 = r - GB_PAD + GB_SIZE > ps ? ps - (r - GB_PAD) : GB_SIZE; // one past the last tap
                        float acc = 0.0f;
                        if (use_w) {
                            for (int ii = i0; ii < i1; ++ii) {
                                const int q = (r + ii - GB_PAD) * ps + c;
                                acc = fmaf(Fl.cc[ii], s_U[q], acc);
                                acc = fmaf(-ws * Fl.cs[ii], s_W[q], acc);
                            }
                        } else {
                            for (int ii = i0; ii < i1; ++ii) acc = fmaf(Fl.cc[ii], s_U[(r + ii - GB_PAD) * ps + c], acc);
                        }
                        const double a = (double)acc;
                        s1 += a;
                        s2 += a * a;
                    }
                }
                s1 = bsum(s1);
                s2 = bsum(s2);
                if (lane == 0) { s_part[wv][f][0] = s1; s_part[wv][f][1] = s2; }
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
    }
    T.n_filters = nf;
}

static hipError_t ensure_tables(nfx_ctx *ctx)
{
    if (ctx->gabor_tab) return hipSuccess;
    static GaborTables T;
    build_tables(T);
    if (T.n_groups > GB_MAX_GROUPS || T.n_filters > GB_MAX_FILTERS) return hipErrorInvalidValue;
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
