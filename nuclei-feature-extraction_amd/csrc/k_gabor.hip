// k_gabor.hip -- the "gabor" feature set: 8 angles x 6 frequencies, masked mean / variance of the response.
//
// Replaces (reference file:line):
//   GaborFilterFeatureSet::compute_features_batched  texture.rs:323-369 (ANGLES_COUNT, FREQUENCIES 319-320)
//   tch_utils::gabor::apply_gabor_filter              call site texture.rs:333-334 (conv2d 1 -> 48 channels)
//
// The reference convolves every patch with 48 dense 30x30 kernels (354 MFLOP per 64x64 patch) and
// then keeps only the masked pixels.  The Gabor kernel with an isotropic envelope factorises:
//     G(x)G(y) cos(Ax + By) = [G(x)cos Ax][G(y)cos By] - [G(x)sin Ax][G(y)sin By]
// i.e. 30-tap row passes followed by 30-tap column passes evaluated ONLY at masked pixels; angles t and t+180
// degrees give the same kernel (cos is even), so 24 filters fill the 48 channels.
//
// EVERYTHING IS PAIRED so that each FMA instruction is a v_pk_fma_f32 (two f32 FMAs per lane and issue slot --
// the only way to the chip's FP32 peak):
//   * a row pass produces a plane PAIR (a, b) = x (*) (ra, rb), stored interleaved (float2) in LDS;
//   * a column pass correlates the pair with a weight pair: (accA, accB) = (a, b) (*) (ca, cb), one packed FMA per
//     tap and output, and yields two filter responses:
//         45 / 135 degrees, |A| = |B|:  (ra, rb) = (G cos Ax, G sin Ax), (ca, cb) = (G cos By, G sin By),
//                                       responses accA - accB and accA + accB;
//         0 degrees (B = 0), two frequencies f0, f1 at once: (ra, rb) = (G cos A0 x, G cos A1 x), (ca, cb) = (G, G);
//         90 degrees (A = 0): (ra, rb) = (G, G) once, then three column passes with (ca, cb) = (G cos B0 y, G cos B1 y).
//   10 row passes and 12 column passes serve the whole bank (25 + 24 unpaired ones before).
//
// One 256-thread workgroup per nucleus (persistent): gray patch with a zero apron and the plane pair (zero aprons,
// odd row stride) live in LDS (76 KB, two workgroups per CU).  Both passes are REGISTER-TILED and WAVE-STRIPED:
// wave w owns strip w (8, 12 or 16 outputs wide, chosen per nucleus from the mask's bounding box so that four
// strips cover it); a lane loads a (TS+29)-value window once and produces TS outputs, the 30 tap pairs arrive as
// LDS broadcasts from a double-buffered per-group table.  Row passes cover only the mask's column range and the
// rows within 15 of the mask; per-strip sums are f32 about a per-filter shift (DC gain x mean masked gray),
// accumulated in f64 and reduced with DPP.
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
static constexpr int GB_MAX_GROUPS = 16, GB_MAX_FILTERS = 24, GB_MAX_PASSES = 3;
static constexpr int GB_WSLOT = 64;                            // one pass: 30 weight pairs padded to 32 float2
static constexpr int GB_WBUF = GB_WSLOT * (1 + GB_MAX_PASSES); // row pass + up to three column passes

typedef float f2 __attribute__((ext_vector_type(2)));

struct GaborGroup {
    int n_pass, pad0, pad1, pad2;
    int combine[4];            // per column pass: 0 -> responses (accA, accB); 1 -> (accA - accB, accA + accB)
    int filt[4][2];            // per column pass: the two filter slots its responses belong to (-1: none)
    float image[GB_WBUF];      // the group's taps exactly as they sit in LDS: [row (ra,rb) x 32 | pass0 (ca,cb) x 32 | pass1 | pass2]
};
struct GaborFilter {
    int ch0, ch1;
    float dc; // sum of the 2-D kernel: response to a constant image of 1
    int pad;
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

// TS output pairs of a 30-tap correlation from a (TS+29)-value register window; the tap pairs come from LDS
// (every lane reads the same address: a broadcast).  Tap-major order, pinned: the TS packed FMAs of one tap hit TS
// different accumulators (independent, back-to-back issue); left alone, the scheduler serialises all 30 taps of
// one accumulator into a dependent chain.
// The pairs are fetched six taps at a time, one block ahead of their use, so only two blocks (24 registers, not 64)
// are live and the LDS latency of the next block hides behind the current block's FMAs.
static constexpr int GB_TBLK = 6, GB_NBLK = GB_SIZE / GB_TBLK;

static __device__ __forceinline__ void load_tap_block(const float *w, int blk, f2 (&wr)[GB_TBLK])
{
    const float4 *w4 = reinterpret_cast<const float4 *>(w + 2 * GB_TBLK * blk);
#pragma unroll
    for (int j = 0; j < GB_TBLK; j += 2) {
        const float4 v = w4[j / 2];
        wr[j] = (f2){v.x, v.y};
        wr[j + 1] = (f2){v.z, v.w};
    }
}

// PAIRWIN = false: row pass, one gray window, two kernels: (a, b)[o] = sum_j (ra, rb)[j] * x[o + j]
// PAIRWIN = true:  column pass, a window of plane pairs:   (accA, accB)[o] = sum_j (ca, cb)[j] * (a, b)[o + j]
template <int TS, typename X>
static __device__ __forceinline__ void taps2(const X (&x)[TS + GB_SIZE - 1], const float *w, f2 (&acc)[TS])
{
    f2 wr[2][GB_TBLK];
    load_tap_block(w, 0, wr[0]);
#pragma unroll
    for (int o = 0; o < TS; ++o) acc[o] = (f2){0.0f, 0.0f};
#pragma unroll
    for (int blk = 0; blk < GB_NBLK; ++blk) {
        if (blk + 1 < GB_NBLK) load_tap_block(w, blk + 1, wr[(blk + 1) & 1]);
#pragma unroll
        for (int jj = 0; jj < GB_TBLK; ++jj) {
            const int j = blk * GB_TBLK + jj;
#pragma unroll
            for (int o = 0; o < TS; ++o) {
                f2 xv;
                if constexpr (sizeof(X) == sizeof(float)) xv = (f2){x[o + j], x[o + j]};
                else xv = x[o + j];
                acc[o] = __builtin_elementwise_fma(wr[blk & 1][jj], xv, acc[o]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

struct GaborShared {
    float *g;
    f2 *UW;          // plane pair, interleaved
    const float *wt; // current group's taps
    const uint64_t *rows;
};

// One group (one row pass + its column passes) with strip size TS: wave w owns column strip w of the row pass
// and row strip w of the column passes, so each pass is exactly one round for every wave.
template <int TS>
static __device__ __forceinline__ void gabor_group(const GaborShared &S, const GaborTables *__restrict__ tab, int g, int ps,
                                                   int gw, int rlo, int nrow_rp, int rmin, int rmax, int cmin, int cmax,
                                                   float mu_gray, int lane, int wv, float (&s1)[6], float (&s2)[6], int dbg)
{
    constexpr int WIN = TS + GB_SIZE - 1;
    const GaborGroup &G = tab->groups[g];
    // ---- row pass: lanes run along rows (LDS stride gw, odd), the wave's strip along columns ----------------
    {
        const int c0 = cmin + wv * TS;
        if (c0 <= cmax && lane < nrow_rp && !(dbg & 1)) {
            const int r = rlo + lane;
            int c0c = c0 < ps - TS ? c0 : ps - TS; // keep the window inside the apron row
            c0c = c0c > 0 ? c0c : 0;
            const float *src = &S.g[r * gw + c0c]; // gray column c-15 sits at apron index c
            float x[WIN];
            f2 acc[TS];
#pragma unroll
            for (int j = 0; j < WIN; ++j) x[j] = src[j];
            taps2<TS, float>(x, S.wt, acc);
            f2 *du = &S.UW[(r + GB_PAD) * GB_UW + c0c];
            // unconditional stores (c0c <= ps - TS keeps them inside the row; for ps < TS they land in the row's unused
            // tail): a guarded store makes the compiler sink each output's 30 FMAs into its own branch -- one
            // dependent chain per output instead of TS independent ones
#pragma unroll
            for (int q = 0; q < TS; ++q) du[q] = acc[q];
        }
    }
    __syncthreads();
    // ---- column passes: lanes run along columns, the wave's strip along rows -----------------------------------
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
                f2 x[WIN];
                const f2 *su = &S.UW[r0c * GB_UW + c]; // row (r0c - 15) + 15
#pragma unroll
                for (int j = 0; j < WIN; ++j) x[j] = su[j * GB_UW];
#pragma unroll
                for (int p = 0; p < GB_MAX_PASSES; ++p) {
                    if (p < G.n_pass) {
                        f2 acc[TS];
                        taps2<TS, f2>(x, S.wt + (1 + p) * GB_WSLOT, acc);
                        const int f0 = G.filt[p][0], f1 = G.filt[p][1];
                        const f2 shift = {f0 >= 0 ? tab->filters[f0].dc * mu_gray : 0.0f, f1 >= 0 ? tab->filters[f1].dc * mu_gray : 0.0f};
                        const bool mix = G.combine[p] != 0;
                        // masked, shifted partial sums of the strip's TS responses (f32 inside the strip and the wave, f64 across waves)
                        f2 a = {0.0f, 0.0f}, b = {0.0f, 0.0f};
#pragma unroll
                        for (int q = 0; q < TS; ++q) {
                            const f2 v = mix ? (f2){acc[q].x - acc[q].y, acc[q].x + acc[q].y} : acc[q];
                            const float mq = ((mbits >> q) & 1u) ? 1.0f : 0.0f; // the mask as a 0/1 factor: packed multiply
                            const f2 d = (v - shift) * (f2){mq, mq};
                            a += d;
                            b = __builtin_elementwise_fma(d, d, b);
                        }
                        // every wave runs exactly one strip per pass: the strip's sums ARE the lane's sums
                        s1[2 * p] = a.x; s2[2 * p] = b.x;
                        s1[2 * p + 1] = a.y; s2[2 * p + 1] = b.y;
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
    __shared__ f2 s_UW[GB_UH * GB_UW]; // plane pair; row r lives at index r + GB_PAD; aprons stay zero
    __shared__ uint64_t s_rows[64 + 16];
    __shared__ double s_part[4][24][2];
    __shared__ float s_gsum[4];

    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int P = ps * ps;
    const int gw = ps + GB_SIZE - 1;
    for (int k = t; k < NFX_MAX_PATCH_SIZE * GB_GW + 16; k += GB_THREADS) s_g[k] = 0.0f; // apron stays zero
    for (int k = t; k < GB_UH * GB_UW; k += GB_THREADS) s_UW[k] = (f2){0.0f, 0.0f};
    for (int k = t; k < 2 * GB_WBUF; k += GB_THREADS) (&s_wt[0][0])[k] = 0.0f;
    if (t < 64 + 16) s_rows[t] = 0ull;
    __syncthreads();
    const int n_groups = tab->n_groups, n_filters = tab->n_filters;
    GaborShared S;
    S.g = s_g; S.UW = s_UW; S.rows = s_rows;

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
            float s1[6], s2[6];
#pragma unroll
            for (int f = 0; f < 6; ++f) { s1[f] = 0.0f; s2[f] = 0.0f; }
            if (span <= 32) gabor_group<8>(S, tab, g, ps, gw, rlo, nrow_rp, rmin, rmax, cmin, cmax, mu_gray, lane, wv, s1, s2, dbg);
            else if (span <= 48) gabor_group<12>(S, tab, g, ps, gw, rlo, nrow_rp, rmin, rmax, cmin, cmax, mu_gray, lane, wv, s1, s2, dbg);
            else gabor_group<16>(S, tab, g, ps, gw, rlo, nrow_rp, rmin, rmax, cmin, cmax, mu_gray, lane, wv, s1, s2, dbg);
#pragma unroll
            for (int f = 0; f < 6; ++f) {
                const int slot = (f >> 1) < G.n_pass ? G.filt[f >> 1][f & 1] : -1;
                if (slot >= 0) {
                    // each lane holds one strip's shifted sums (small numbers): an f32 wave reduction is exact enough
                    const float a = wave_total_f32(s1[f]), b = wave_total_f32(s2[f]);
                    if (lane == 0) { s_part[wv][slot][0] = (double)a; s_part[wv][slot][1] = (double)b; }
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

// ---- host: factorised, paired filter bank (DESIGN.md "Gabor bank") --------------------------------------
static void build_tables(GaborTables &T)
{
    const double PI = 3.14159265358979323846, sigma = 0.45;
    const double freqs[6] = {0.5, 1.0, 2.0, 4.0, 6.0, 8.0}; // texture.rs:320
    memset(&T, 0, sizeof(T));
    auto tk = [](int k) { return -1.0 + 2.0 * (double)k / (double)(GB_SIZE - 1); };
    auto env = [&](double x) { return exp(-(x * x) / (2.0 * sigma * sigma)); };
    auto close = [](double x, double y) { return fabs(x - y) < 1e-9; };
    // unique filters: cos(Ax+By) == cos(-Ax-By); channel = angle_index * 6 + frequency_index (texture.rs:346-361)
    struct Uniq { double A, B; int ch0, ch1, slot; bool placed; };
    std::vector<Uniq> uniq;
    for (int a = 0; a < 8; ++a)
        for (int f = 0; f < 6; ++f) {
            const double th = (double)a * PI / 4.0;
            double A = 2.0 * PI * freqs[f] * cos(th), B = 2.0 * PI * freqs[f] * sin(th);
            if (fabs(A) < 1e-9) A = 0.0;
            if (fabs(B) < 1e-9) B = 0.0;
            bool merged = false;
            for (auto &u : uniq)
                if (u.ch1 < 0 && ((close(A, u.A) && close(B, u.B)) || (close(A, -u.A) && close(B, -u.B)))) { u.ch1 = a * 6 + f; merged = true; break; }
            if (!merged) uniq.push_back({A, B, a * 6 + f, -1, -1, false});
        }
    int nf = 0;
    for (auto &u : uniq) {
        if (nf >= GB_MAX_FILTERS) { T.n_filters = GB_MAX_FILTERS + 1; return; }
        u.slot = nf;
        GaborFilter &Fl = T.filters[nf++];
        Fl.ch0 = u.ch0; Fl.ch1 = u.ch1;
        double dc = 0.0; // DC gain: sum_ij G(x_j)G(y_i) cos(A x_j + B y_i)
        for (int ii = 0; ii < GB_SIZE; ++ii)
            for (int jj = 0; jj < GB_SIZE; ++jj) dc += env(tk(ii)) * env(tk(jj)) * cos(u.A * tk(jj) + u.B * tk(ii));
        Fl.dc = (float)dc;
    }
    T.n_filters = nf;
    int ng = 0;
    auto new_group = [&]() -> GaborGroup * {
        if (ng >= GB_MAX_GROUPS) return nullptr;
        GaborGroup &G = T.groups[ng++];
        for (int p = 0; p < 4; ++p) { G.filt[p][0] = G.filt[p][1] = -1; G.combine[p] = 0; }
        return &G;
    };
    auto set_row = [&](GaborGroup &G, int j, double ra, double rb) { G.image[2 * j] = (float)ra; G.image[2 * j + 1] = (float)rb; };
    auto set_col = [&](GaborGroup &G, int p, int j, double ca, double cb) { G.image[(1 + p) * GB_WSLOT + 2 * j] = (float)ca; G.image[(1 + p) * GB_WSLOT + 2 * j + 1] = (float)cb; };
    bool ok = true;
    // (1) A != 0, B != 0: a filter and its mirror (-A, B) share both passes: out = accA -+ accB
    for (auto &u : uniq) {
        if (u.placed || u.A == 0.0 || u.B == 0.0) continue;
        Uniq *mir = nullptr;
        for (auto &v : uniq)
            if (!v.placed && &v != &u && close(v.A, -u.A) && close(v.B, u.B)) { mir = &v; break; }
        GaborGroup *G = new_group();
        if (!G) { ok = false; break; }
        const double Aa = fabs(u.A);
        // cos(Ax + By) = cos|A|x cos By - sign(A) sin|A|x sin By
        Uniq *plus = u.A > 0.0 ? &u : mir, *minus = u.A > 0.0 ? mir : &u; // plus: sign(A) = +1 -> accA - accB
        for (int j = 0; j < GB_SIZE; ++j) {
            set_row(*G, j, env(tk(j)) * cos(Aa * tk(j)), env(tk(j)) * sin(Aa * tk(j)));
            set_col(*G, 0, j, env(tk(j)) * cos(u.B * tk(j)), env(tk(j)) * sin(u.B * tk(j)));
        }
        G->n_pass = 1; G->combine[0] = 1;
        G->filt[0][0] = plus ? plus->slot : -1;
        G->filt[0][1] = minus ? minus->slot : -1;
        u.placed = true;
        if (mir) mir->placed = true;
    }
    // (2) B = 0: the column kernel is the bare envelope; two frequencies ride in one plane pair
    for (size_t i = 0; i < uniq.size() && ok; ++i) {
        Uniq &u = uniq[i];
        if (u.placed || u.B != 0.0) continue;
        Uniq *v = nullptr;
        for (size_t k = i + 1; k < uniq.size(); ++k)
            if (!uniq[k].placed && uniq[k].B == 0.0) { v = &uniq[k]; break; }
        GaborGroup *G = new_group();
        if (!G) { ok = false; break; }
        for (int j = 0; j < GB_SIZE; ++j) {
            set_row(*G, j, env(tk(j)) * cos(u.A * tk(j)), v ? env(tk(j)) * cos(v->A * tk(j)) : 0.0);
            set_col(*G, 0, j, env(tk(j)), env(tk(j)));
        }
        G->n_pass = 1;
        G->filt[0][0] = u.slot; G->filt[0][1] = v ? v->slot : -1;
        u.placed = true;
        if (v) v->placed = true;
    }
    // (3) A = 0: one row pass with the bare envelope (in both halves), two frequencies per column pass
    {
        std::vector<Uniq *> col;
        for (auto &u : uniq)
            if (!u.placed && u.A == 0.0) col.push_back(&u);
        for (size_t i = 0; i < col.size() && ok; i += 2 * GB_MAX_PASSES) {
            GaborGroup *G = new_group();
            if (!G) { ok = false; break; }
            for (int j = 0; j < GB_SIZE; ++j) set_row(*G, j, env(tk(j)), env(tk(j)));
            for (int p = 0; p < GB_MAX_PASSES && i + 2 * p < col.size(); ++p) {
                Uniq *u = col[i + 2 * p], *v = i + 2 * p + 1 < col.size() ? col[i + 2 * p + 1] : nullptr;
                for (int j = 0; j < GB_SIZE; ++j) set_col(*G, p, j, env(tk(j)) * cos(u->B * tk(j)), v ? env(tk(j)) * cos(v->B * tk(j)) : 0.0);
                G->filt[p][0] = u->slot; G->filt[p][1] = v ? v->slot : -1;
                u->placed = true;
                if (v) v->placed = true;
                G->n_pass = p + 1;
            }
        }
    }
    for (auto &u : uniq) ok = ok && u.placed;
    T.n_groups = ok ? ng : GB_MAX_GROUPS + 1;
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
