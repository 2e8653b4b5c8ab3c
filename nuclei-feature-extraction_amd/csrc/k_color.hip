// k_color.hip -- the "color" feature set (masked RGB / HSV / HED statistics) on gfx950.
//
// Replaces (reference file:line):
//   ColorFeatureSet::compute_features_batched   color.rs:11-102
//   mean_std                                    color.rs:117-134
//   circular_mean                               color.rs:144-155 (batch-coupled by its broadcast)
//   tch_utils::color::{hsv_from_rgb,hed_from_rgb} call sites color.rs:45-46
//
// Two kernels:
//   k_hue_batch  one lane per (reference batch, pixel): S = sum_j sin(hue_j), C = sum_j cos(hue_j)
//                over every patch j of the batch.  This is the cheap form of the reference's
//                [N,N,ps,ps] broadcast: nucleus i's circular mean is atan2(mask_i . S, mask_i . C).
//   k_color      one wavefront per nucleus.  The masked pixels' positions are compacted into a per-wave LDS list
//                (a mask row fills well under half a wave), then ONE pass walks the list with every lane busy,
//                two pixels per lane in flight: it dots the mask with S and C (-> mean_h), converts
//                RGB->HSV/HED in registers and accumulates sum x and sum x^2 of nine channels (exact integers for the
//                byte-valued r, g, b, v of u8 pixels, f64 for the rest); the
//                variance about the f32-rounded mean is then sum x^2 - 2 mu sum x + K mu^2 (the two-pass
//                result of color.rs:123-128 without a second pass over the pixels).  The hue statistics are
//                taken on h itself: the shift by mean_h (color.rs:52) only reaches the standard deviation,
//                which is shift-invariant, so mean_h is not needed before the pixels are read.
//   Blocks of one reference batch share blockIdx % 8, i.e. one XCD, so the batch's S/C images are
//   served by that XCD's L2 to all of its nuclei.
#include "exact_div.h"
#include "nfx_internal.h"
#include "wave_ops.h"

namespace nfx {

static constexpr int COLOR_LIST_CAP = 2048; // list entries per wave

template <typename T> struct Px;
template <> struct Px<uint8_t> {
    static __device__ __forceinline__ float ld(const uint8_t *p, size_t i) { return u8_unit(p[i]); } // utils.rs:172 (exact_div.h)
};
template <> struct Px<float> {
    static __device__ __forceinline__ float ld(const float *p, size_t i) { return p[i]; }
};

static __device__ __forceinline__ double wave_sum_f64(double v) { return wave_total_f64(v); }

// Smith hexcone, H in degrees (DESIGN.md "colour transforms").  The two quotients use the hardware reciprocal (1 ulp)
// instead of the eleven-instruction IEEE division: hue and saturation are float statistics compared at 1e-5, and
// the conversion is a build decision of this repository (tch-utils' own arithmetic is not vendored), not a pinned one.
static __device__ __forceinline__ float quot(float x, float y) { return x * __builtin_amdgcn_rcpf(y); }

static __device__ __forceinline__ void hsv_from_rgb(float r, float g, float b, float &h, float &s, float &v)
{
    const float mx = fmaxf(r, fmaxf(g, b));
    const float mn = fminf(r, fminf(g, b));
    const float d = mx - mn;
    float hh;
    if (d == 0.0f) hh = 0.0f;
    else if (mx == r) { hh = quot(g - b, d); if (hh < 0.0f) hh += 6.0f; } // |hh| <= 1: fmod(hh, 6) is the identity
    else if (mx == g) hh = quot(b - r, d) + 2.0f;
    else hh = quot(r - g, d) + 4.0f;
    h = hh * 60.0f;
    s = (mx == 0.0f) ? 0.0f : quot(d, mx);
    v = mx;
}

// sin and cos of the hue angle without a general argument reduction: the hexcone hue is 120 s + 60 f degrees with a
// sector s in {0,1,2} and f = (difference of the two smaller channels) / (max - min) in [-1,1], so
//     sin(hue) = sin(120 s) cos(x) + cos(120 s) sin(x),  x = f pi/3 in [-1.05, 1.05]
// and two short Taylor polynomials (errors 4e-8 and 4e-9 at the interval ends) replace sincosf.  The reference
// rounds 60 f and the degree-to-radian product to f32 first (color.rs:147), which moves the angle by up to 7e-7 rad;
// the sums of 100 patches x K pixels that follow are compared at 1e-5.  (The hardware sine / cosine on the angle in
// revolutions are 12 % faster for this kernel and 1e-6 off: a batch whose hues spread round the circle has a short
// resultant vector, and its circular mean moved by 6e-3 degrees -- outside the tolerance -- in tests/test_geometry_extremes_gpu.py.)
static __device__ __forceinline__ void hue_sincos(float r, float g, float b, float &sn, float &cs)
{
    const float mx = fmaxf(r, fmaxf(g, b));
    const float mn = fminf(r, fminf(g, b));
    const float d = mx - mn;
    // same tie order as hsv_from_rgb: red first, then green
    const bool isr = mx == r, isg = !isr && mx == g;
    const float num = isr ? g - b : (isg ? b - r : r - g);
    const float f = d == 0.0f ? 0.0f : quot(num, d);
    const bool s0 = isr || d == 0.0f;
    const float ss = s0 ? 0.0f : (isg ? 0.86602540378443865f : -0.86602540378443865f); // sin(120 s)
    const float sc = s0 ? 1.0f : -0.5f;                                                  // cos(120 s)
    const float x = f * 1.0471975511965976f, x2 = x * x;
    float ps = fmaf(x2, 2.7557319223985893e-06f, -1.9841269841269841e-04f);
    ps = fmaf(x2, ps, 8.3333333333333333e-03f);
    ps = fmaf(x2, ps, -1.6666666666666667e-01f);
    ps = fmaf(x2 * x, ps, x);
    float pc = fmaf(x2, -2.7557319223985888e-07f, 2.4801587301587302e-05f);
    pc = fmaf(x2, pc, -1.3888888888888889e-03f);
    pc = fmaf(x2, pc, 4.1666666666666664e-02f);
    pc = fmaf(x2, pc, -0.5f);
    pc = fmaf(x2, pc, 1.0f);
    sn = fmaf(ss, pc, sc * ps);
    cs = fmaf(sc, pc, -(ss * ps));
}

// optical density log(max(x,1e-6))/log(1e-6), evaluated in f64 and rounded (f32 inputs only;
// u8 inputs use the 256-entry table built on the host with the same formula)
static __device__ __forceinline__ float od_f32(float x)
{
    const float xc = x > 1e-6f ? x : 1e-6f;
    return (float)(log((double)xc) / log(1e-6));
}

__constant__ float kHedFromRgb[3][3] = {
    {1.87798273563385f, -1.0076786279678345f, -0.5561158061027527f},
    {-0.06590805947780609f, 1.134730339050293f, -0.135521799325943f},
    {-0.6019073724746704f, -0.48041418194770813f, 1.5735880136489868f},
};

static __device__ __forceinline__ void hed_from_od(float o0, float o1, float o2, float e[3])
{
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float s = o0 * kHedFromRgb[0][k];
        s = s + o1 * kHedFromRgb[1][k];
        s = s + o2 * kHedFromRgb[2][k];
        e[k] = s > 0.0f ? s : 0.0f;
    }
}

// ---- k_hue_batch ---------------------------------------------------------------------------------
// VEC pixels per lane (one VEC-byte load per channel and patch; VEC = 4 when ps*ps is a multiple of 4)
template <typename T, int VEC>
__global__ __launch_bounds__(256) void k_hue_batch(int64_t n, int ps, int batch_size, const T *__restrict__ patches,
                                                   double *__restrict__ SC)
{
    const int P = ps * ps;
    const int64_t b = blockIdx.y;
    const int px = (blockIdx.x * blockDim.x + threadIdx.x) * VEC;
    if (px >= P) return;
    const int64_t j0 = b * batch_size;
    const int64_t j1 = (j0 + batch_size) < n ? (j0 + batch_size) : n;
    double S[VEC], C[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) { S[q] = 0.0; C[q] = 0.0; }
    for (int64_t j = j0; j < j1; ++j) {
        const T *p = patches + (size_t)j * 3 * P;
        float r[VEC], g[VEC], bl[VEC];
        if (sizeof(T) == 1 && VEC == 4) {
            const uint8_t *pb = reinterpret_cast<const uint8_t *>(p);
            const uchar4 r4 = *reinterpret_cast<const uchar4 *>(pb + px), g4 = *reinterpret_cast<const uchar4 *>(pb + P + px),
                         b4 = *reinterpret_cast<const uchar4 *>(pb + 2 * P + px);
            const uint8_t rr[4] = {r4.x, r4.y, r4.z, r4.w}, gg[4] = {g4.x, g4.y, g4.z, g4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            // the hue angle depends on the channels only through order and the ratio of two differences: the raw byte
            // values serve as well as value/255 (the quotient moves by 2e-7 relative, inside hue_sincos' own deviation)
            for (int q = 0; q < VEC; ++q) { r[q] = (float)rr[q]; g[q] = (float)gg[q]; bl[q] = (float)bb[q]; }
        } else {
#pragma unroll
            for (int q = 0; q < VEC; ++q) { r[q] = Px<T>::ld(p, px + q); g[q] = Px<T>::ld(p, P + px + q); bl[q] = Px<T>::ld(p, 2 * P + px + q); }
        }
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            float sn, cs;
            hue_sincos(r[q], g[q], bl[q], sn, cs); // color.rs:147-149
            S[q] += (double)sn;
            C[q] += (double)cs;
        }
    }
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
        SC[((size_t)b * 2 + 0) * P + px + q] = S[q];
        SC[((size_t)b * 2 + 1) * P + px + q] = C[q];
    }
}

// nine channels of one pixel: r g b h' s v he eo da (color.rs:45-53)
template <typename T>
static __device__ __forceinline__ void pixel_channels(float r, float g, float b, unsigned b0, unsigned b1, unsigned b2, float mh,
                                                      const float *s_od, float (&x)[9])
{
    x[0] = r; x[1] = g; x[2] = b;
    float h;
    hsv_from_rgb(r, g, b, h, x[4], x[5]);
    x[3] = h - mh; // color.rs:52
    float o0, o1, o2;
    if (sizeof(T) == 1) {
        o0 = s_od[b0]; o1 = s_od[b1]; o2 = s_od[b2];
    } else {
        o0 = od_f32(r); o1 = od_f32(g); o2 = od_f32(b);
    }
    hed_from_od(o0, o1, o2, &x[6]);
}

// ---- k_color ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_color(int64_t n, int ps, int batch_size, int64_t n_batches, int blocks_per_batch,
                                               const T *__restrict__ patches, const uint64_t *__restrict__ mask_rows,
                                               const double *__restrict__ SC, const float *__restrict__ od_lut,
                                               float *__restrict__ out, int F, int col0)
{
    __shared__ float s_od[256];
    extern __shared__ __attribute__((aligned(16))) uint8_t s_patch_all[]; // 4 waves x min(ps*ps, COLOR_LIST_CAP) list entries
    s_od[threadIdx.x] = od_lut[threadIdx.x];
    __syncthreads();
    // XCD-aware block -> (batch, slot) map: every block of a batch has the same blockIdx % 8
    const int64_t v = blockIdx.x;
    const int64_t xcd = v & 7, slot = v >> 3;
    const int64_t batch = (slot / blocks_per_batch) * 8 + xcd;
    const int within = (int)(slot % blocks_per_batch);
    if (batch >= n_batches) return;
    const int lane = threadIdx.x & 63;
    const int k_in_batch = within * 4 + (threadIdx.x >> 6);
    const int64_t i = batch * batch_size + k_in_batch;
    if (k_in_batch >= batch_size || i >= n) return; // wave-uniform

    const int P = ps * ps;
    const T *pp = patches + (size_t)i * 3 * P;
    auto ld3 = [&](int p, unsigned &b0, unsigned &b1, unsigned &b2) { // the three channel bytes of pixel p (u8 paths)
        const uint8_t *pb = reinterpret_cast<const uint8_t *>(pp);
        b0 = pb[p]; b1 = pb[P + p]; b2 = pb[2 * P + p];
    };
    // the masked pixels' positions, compacted: masks beyond the list's capacity (only the largest nuclei of a 64x64
    // patch) leave their last pixels to a row-wise tail, so the LDS footprint stays at four waves x 4 KB
    const int cap = P < COLOR_LIST_CAP ? P : COLOR_LIST_CAP;
    unsigned short *s_list = reinterpret_cast<unsigned short *>(s_patch_all) + (threadIdx.x >> 6) * cap;
    const uint64_t myrow = lane < ps ? mask_rows[i * ps + lane] : 0ull;
    const uint64_t anyrow = __ballot(myrow != 0ull); // rows that hold mask pixels
    const double *S = SC + (size_t)batch * 2 * P;
    const double *Cc = S + P;

    const int rcnt = __popcll(myrow);
    const int incl = wave_inclusive_scan_i32(rcnt); // list position of the first pixel of this lane's row = exclusive prefix over the rows
    const int K = __builtin_amdgcn_readlane(incl, 63);
    const int rowstart = incl - rcnt;
    {   // position list: lane = column, stores only (no round trip in the loop)
        const unsigned rlo = (unsigned)myrow, rhi = (unsigned)(myrow >> 32);
        uint64_t rows_left = anyrow;
        while (rows_left) {
            const int r = __ffsll((long long)rows_left) - 1;
            rows_left &= rows_left - 1ull;
            const unsigned wlo = __builtin_amdgcn_readlane(rlo, r), whi = __builtin_amdgcn_readlane(rhi, r);
            const int base = __builtin_amdgcn_readlane(rowstart, r);
            const uint64_t w = ((uint64_t)whi << 32) | wlo;
            const int at = base + (int)__builtin_amdgcn_mbcnt_hi(whi, __builtin_amdgcn_mbcnt_lo(wlo, 0u));
            if (((w >> lane) & 1ull) && at < cap) s_list[at] = (unsigned short)(r * ps + lane);
        }
    }
    wave_lds_fence();

    // ONE pass over the list, every lane busy: the circular sums (color.rs:148-152) and sum x, sum x^2 of
    // r g b h s v he eo da.  The hue statistics are taken on h itself: the reference shifts h by mean_h first
    // (color.rs:52), which only the standard deviation sees -- and that is shift-invariant -- so mean_h is not needed
    // before the pixels are read, and one pass replaces two.
    // u8 pixels: r, g, b and v = max(r, g, b) are byte / 255, so their sums are kept as exact integers (two integer
    // operations per channel and pixel instead of three f64 ones) and scaled once at the end
    constexpr bool INT4 = sizeof(T) == 1;
    double ss = 0.0, cc = 0.0;
    // The float-valued channels are summed about a PIVOT -- an estimate of the channel's mean over the nucleus, the same
    // in every lane -- in f32 per lane (at most 64 terms) and in f64 across lanes: d = x - pivot is small against x, so
    // sum d and sum d^2 carry no cancellation worth f64 (three f32 instructions per channel and pixel instead of four
    // f64 ones at twice the issue cost); mean and variance are put together in f64 at the end.
    float f1[9], f2s[9], pv[9];
    unsigned iu[4] = {0u, 0u, 0u, 0u}, iu2[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < 9; ++k) { f1[k] = 0.0f; f2s[k] = 0.0f; pv[k] = 0.0f; }
    {   // pivot = mean of 64 pixels spread evenly over the list (one per lane; a wave total per channel)
        const int Kc = K < cap ? K : cap;
        const int pm = Kc > 0 ? s_list[(lane * Kc) >> 6] : 0;
        float smp[9];
        if (sizeof(T) == 1) {
            unsigned b0, b1, b2;
            ld3(pm, b0, b1, b2);
            pixel_channels<T>(u8_unit(b0), u8_unit(b1), u8_unit(b2), b0, b1, b2, 0.0f, s_od, smp);
        } else {
            pixel_channels<T>(Px<T>::ld(pp, pm), Px<T>::ld(pp, P + pm), Px<T>::ld(pp, 2 * P + pm), 0u, 0u, 0u, 0.0f, s_od, smp);
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            if (INT4 && (k < 3 || k == 5)) continue;
            const float m = wave_total_f32(smp[k]) * (1.0f / 64.0f);
            pv[k] = (m == m && fabsf(m) < 3.0e38f) ? m : 0.0f; // a NaN / infinite pixel must not poison every difference
        }
    }
    auto pixel = [&](int p, bool on, unsigned b0, unsigned b1, unsigned b2, float f0, float f1_, float f2_, double sv, double cv) {
        float x[9];
        if (sizeof(T) == 1) pixel_channels<T>(u8_unit(b0), u8_unit(b1), u8_unit(b2), b0, b1, b2, 0.0f, s_od, x); // utils.rs:172
        else pixel_channels<T>(f0, f1_, f2_, 0u, 0u, 0u, 0.0f, s_od, x);
        ss += on ? sv : 0.0;
        cc += on ? cv : 0.0;
        if (INT4) {
            const unsigned m = b0 > b1 ? (b0 > b2 ? b0 : b2) : (b1 > b2 ? b1 : b2);
            const unsigned u4[4] = {on ? b0 : 0u, on ? b1 : 0u, on ? b2 : 0u, on ? m : 0u};
#pragma unroll
            for (int c = 0; c < 4; ++c) { iu[c] += u4[c]; iu2[c] += u4[c] * u4[c]; }
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            if (INT4 && (k < 3 || k == 5)) continue;
            const float d = on ? x[k] - pv[k] : 0.0f;
            f1[k] += d;
            f2s[k] = fmaf(d, d, f2s[k]);
        }
    };
    const int Kl = K < cap ? K : cap;
    for (int e = lane; e < Kl; e += 128) { // two entries per lane: ten loads in flight before the first use
        const bool on1 = e + 64 < Kl;
        const int p0 = s_list[e], p1 = on1 ? s_list[e + 64] : p0;
        unsigned a0 = 0, a1 = 0, a2 = 0, c0 = 0, c1 = 0, c2 = 0;
        float fa0 = 0, fa1 = 0, fa2 = 0, fc0 = 0, fc1 = 0, fc2 = 0;
        if (sizeof(T) == 1) {
            ld3(p0, a0, a1, a2);
            ld3(p1, c0, c1, c2);
        } else {
            fa0 = Px<T>::ld(pp, p0); fa1 = Px<T>::ld(pp, P + p0); fa2 = Px<T>::ld(pp, 2 * P + p0);
            fc0 = Px<T>::ld(pp, p1); fc1 = Px<T>::ld(pp, P + p1); fc2 = Px<T>::ld(pp, 2 * P + p1);
        }
        const double sv0 = S[p0], cv0 = Cc[p0], sv1 = S[p1], cv1 = Cc[p1];
        pixel(p0, true, a0, a1, a2, fa0, fa1, fa2, sv0, cv0);
        pixel(p1, on1, c0, c1, c2, fc0, fc1, fc2, sv1, cv1);
    }
    if (K > cap) { // tail: pixels that found no room in the list, row by row
        for (int r = 0; r < ps; ++r) {
            const uint64_t w = __shfl(myrow, r, 64);
            const int at = __shfl(rowstart, r, 64) + __popcll(w & ((1ull << lane) - 1ull));
            const bool on = lane < ps && ((w >> lane) & 1ull) && at >= cap;
            const int p = on ? r * ps + lane : 0;
            unsigned a0 = 0, a1 = 0, a2 = 0;
            float fa0 = 0, fa1 = 0, fa2 = 0;
            if (sizeof(T) == 1) {
                ld3(p, a0, a1, a2);
            } else {
                fa0 = Px<T>::ld(pp, p); fa1 = Px<T>::ld(pp, P + p); fa2 = Px<T>::ld(pp, 2 * P + p);
            }
            pixel(p, on, a0, a1, a2, fa0, fa1, fa2, S[p], Cc[p]);
        }
    }
    const float area = (float)K;
    float *o = out + i * (int64_t)F + col0; // column order color.rs:80-99: mean rgb, std rgb, mean hsv, std hsv, mean hed, std hed
    // mean and population standard deviation of channel k, one lane per channel, from the sums about the pivot p:
    // a = sum (x - p), b = sum (x - p)^2.  mean = (float)(sum x) / K as color.rs:119-121 has it; the deviations are taken about
    // that f32 mean, sum (x - mu)^2 = b - 2 (mu - p) a + K (mu - p)^2 (the two-pass result of color.rs:123-128)
    auto finish = [&](int k, double a, double b, double p, float mh) {
        const float mu = (float)(a + (double)K * p) / area;
        const double md = (double)mu - p;
        double D = b - 2.0 * md * a + (double)K * md * md;
        D = D > 0.0 ? D : (D == D ? 0.0 : D);                 // rounding can leave -1e-17; keep NaN
        const float sd = sqrtf((float)D / area);
        o[k < 3 ? k : (k < 6 ? 3 + k : 6 + k)] = k == 3 ? mh : mu; // mean_h is the circular mean
        o[k < 3 ? 3 + k : (k < 6 ? 6 + k : 9 + k)] = sd;
    };
    auto circular_mean = [&](double ssum, double csum) {
        const float rad2deg = (float)(180.0 / 3.14159265358979323846);
        const float sinm = (float)ssum / area, cosm = (float)csum / area;
        float mh = (float)atan2((double)sinm, (double)cosm);
        mh = mh * rad2deg;
        mh = mh + 360.0f;
        return fmodf(mh, 360.0f); // color.rs:154
    };
    // all totals in reduce-scatter butterflies (wave_ops.h); lane l ends up with the pair (sum, sum of squares) of one channel
    if (INT4) {
        unsigned A4[4] = {iu[0], iu[1], iu[2], iu[3]}, B4[4] = {iu2[0], iu2[1], iu2[2], iu2[3]};
        const unsigned ua = wave_reduce_scatter(A4, lane), ub = wave_reduce_scatter(B4, lane);
        double T1[8] = {(double)f1[3], (double)f1[4], (double)f1[6], (double)f1[7], (double)f1[8], ss, cc, 0.0};
        double T2[8] = {(double)f2s[3], (double)f2s[4], (double)f2s[6], (double)f2s[7], (double)f2s[8], 0.0, 0.0, 0.0};
        const double a = wave_reduce_scatter(T1, lane), b = wave_reduce_scatter(T2, lane);
        // scatter_index<8> reverses three bits: slot 5 (101b) sits in lane 5, slot 6 (110b) in lane 3
        ss = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a), 5), __builtin_amdgcn_readlane(__double2loint(a), 5));
        cc = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a), 3), __builtin_amdgcn_readlane(__double2loint(a), 3));
        const float mh = circular_mean(ss, cc);
        if (lane < 4) { // r, g, b, v: exact sums of bytes -> sums of byte / 255 (pivot 0)
            const int c = scatter_index<4>(lane);
            finish(c < 3 ? c : 5, (double)ua / 255.0, (double)ub / 65025.0, 0.0, mh);
        }
        const int j = scatter_index<8>(lane & 7);
        if (lane < 8 && j < 5) { // h, s, he, eo, da
            const int k = j < 2 ? 3 + j : 4 + j;
            float pk = pv[3];
#pragma unroll
            for (int q = 4; q < 9; ++q) pk = k == q ? pv[q] : pk;
            finish(k, a, b, (double)pk, mh);
        }
    } else {
        double T1[16] = {(double)f1[0], (double)f1[1], (double)f1[2], (double)f1[3], (double)f1[4], (double)f1[5], (double)f1[6], (double)f1[7], (double)f1[8], ss, cc, 0.0, 0.0, 0.0, 0.0, 0.0};
        double T2[16] = {(double)f2s[0], (double)f2s[1], (double)f2s[2], (double)f2s[3], (double)f2s[4], (double)f2s[5], (double)f2s[6], (double)f2s[7], (double)f2s[8], 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        const double a = wave_reduce_scatter(T1, lane), b = wave_reduce_scatter(T2, lane);
        // scatter_index<16> reverses the four bits: slot 9 (1001b) sits in lane 9, slot 10 (1010b) in lane 5
        ss = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a), 9), __builtin_amdgcn_readlane(__double2loint(a), 9));
        cc = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a), 5), __builtin_amdgcn_readlane(__double2loint(a), 5));
        const float mh = circular_mean(ss, cc);
        const int k = scatter_index<16>(lane & 15);
        if (lane < 16 && k < 9) {
            float pk = pv[0];
#pragma unroll
            for (int q = 1; q < 9; ++q) pk = k == q ? pv[q] : pk;
            finish(k, a, b, (double)pk, mh);
        }
    }
}

// ---- patch sizes 65..512: one 256-thread workgroup per nucleus (see k_geometry.hip, "patch sizes 65..128" and "129..512") ----
// Same per-pixel arithmetic (pixel_channels), same finishing formulas; the threads stride over the patch, test the mask bit
// in the [ps][W] row words (W = ceil(ps / 64)), and the twenty sums are reduced per wave (DPP) and across the four waves through LDS.
template <typename T>
__global__ __launch_bounds__(256) void k_color_wide(int64_t n, int ps, int batch_size, const T *__restrict__ patches,
                                                    const uint64_t *__restrict__ mask_rows, const double *__restrict__ SC,
                                                    const float *__restrict__ od_lut, float *__restrict__ out, int F, int col0)
{
    __shared__ float s_od[256];
    __shared__ double s_red[4][20];
    __shared__ int s_K[4];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    s_od[t] = od_lut[t];
    __syncthreads();
    const int P = ps * ps;
    for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
        const T *pp = patches + (size_t)i * 3 * P;
        const int W = (ps + 63) >> 6;
        const uint64_t *rows = mask_rows + (size_t)i * ps * W;
        const double *S = SC + (size_t)(i / batch_size) * 2 * P;
        const double *Cc = S + P;
        double acc[20]; // sum x (9), sum x^2 (9), sum sin, sum cos
#pragma unroll
        for (int k = 0; k < 20; ++k) acc[k] = 0.0;
        int K = 0;
        for (int p = t; p < P; p += 256) {
            const int r = p / ps, c = p - r * ps;
            if (!((rows[W * r + (c >> 6)] >> (c & 63)) & 1ull)) continue;
            ++K;
            float x[9];
            if (sizeof(T) == 1) {
                const uint8_t *pb = reinterpret_cast<const uint8_t *>(pp);
                const unsigned b0 = pb[p], b1 = pb[P + p], b2 = pb[2 * P + p];
                pixel_channels<T>(u8_unit(b0), u8_unit(b1), u8_unit(b2), b0, b1, b2, 0.0f, s_od, x); // utils.rs:172
            } else {
                pixel_channels<T>(Px<T>::ld(pp, p), Px<T>::ld(pp, P + p), Px<T>::ld(pp, 2 * P + p), 0u, 0u, 0u, 0.0f, s_od, x);
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) { const double xd = (double)x[k]; acc[k] += xd; acc[9 + k] += xd * xd; }
            acc[18] += S[p];
            acc[19] += Cc[p];
        }
#pragma unroll
        for (int k = 0; k < 20; ++k) {
            const double tot = wave_total_f64(acc[k]);
            if (lane == 0) s_red[wv][k] = tot;
        }
        const int Kw = wave_total_i32(K);
        if (lane == 0) s_K[wv] = Kw;
        __syncthreads();
        if (t < 9) {
            const int Kt = s_K[0] + s_K[1] + s_K[2] + s_K[3];
            const float area = (float)Kt;
            auto tot = [&](int k) { return (s_red[0][k] + s_red[1][k]) + (s_red[2][k] + s_red[3][k]); };
            const float rad2deg = (float)(180.0 / 3.14159265358979323846);
            const float sinm = (float)tot(18) / area, cosm = (float)tot(19) / area;
            float mh = (float)atan2((double)sinm, (double)cosm);
            mh = mh * rad2deg;
            mh = mh + 360.0f;
            mh = fmodf(mh, 360.0f); // color.rs:154
            const int k = t;
            const double a = tot(k), b = tot(9 + k);
            const float mu = (float)a / area;
            const double mud = (double)mu;
            double D = b - 2.0 * mud * a + (double)Kt * mud * mud; // sum (x-mu)^2 about the f32 mean
            D = D > 0.0 ? D : (D == D ? 0.0 : D);
            const float sd = sqrtf((float)D / area);
            float *o = out + i * (int64_t)F + col0;
            o[k < 3 ? k : (k < 6 ? 3 + k : 6 + k)] = k == 3 ? mh : mu;
            o[k < 3 ? 3 + k : (k < 6 ? 6 + k : 9 + k)] = sd;
        }
        __syncthreads();
    }
}

// ---- host side -------------------------------------------------------------------------------------
size_t color_scratch_bytes(const Job &job)
{
    const int64_t nb = job.batch_size > 0 ? (job.n + job.batch_size - 1) / job.batch_size : 0;
    return sizeof(double) * 2 * (size_t)(nb > 0 ? nb : 1) * job.ps * job.ps + 256;
}

hipError_t prepare_color(nfx_ctx *ctx)
{
    if (ctx->od_lut) return hipSuccess;
    if (tables_locked(ctx)) return hipErrorNotReady; // device-memory calls on a reserved context and captures never allocate: nfx_ctx_reserve builds the tables
    float h[256];
    for (int u = 0; u < 256; ++u) {
        const float x = (float)u / 255.0f;
        const float xc = x > 1e-6f ? x : 1e-6f;
        h[u] = (float)(log((double)xc) / log(1e-6));
    }
    float *d = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d), sizeof(h));
    if (e != hipSuccess) return e;
    e = hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d); return e; }
    ctx->od_lut = d; // published only once it is complete
    return hipSuccess;
}

hipError_t launch_color(nfx_ctx *ctx, const Job &job, int col0, void *scratch)
{
    if (job.n == 0) return hipSuccess;
    hipError_t e = prepare_color(ctx);
    if (e != hipSuccess) return e;
    const int P = job.ps * job.ps;
    const int64_t nb = (job.n + job.batch_size - 1) / job.batch_size;
    double *SC = reinterpret_cast<double *>(scratch);
    {
        ScopedTiming t(ctx, KS_HUE_BATCH);
        // gridDim.y is limited to 65535: walk the batches in slabs
        for (int64_t b0 = 0; b0 < nb; b0 += 65535) {
            const int64_t nbs = (nb - b0) < 65535 ? (nb - b0) : 65535;
            const int64_t j0 = b0 * job.batch_size;
            const int64_t nn = job.n - j0;
            dim3 grid((unsigned)((P + 255) / 256), (unsigned)nbs);
            if (job.patches_f32)
                hipLaunchKernelGGL((k_hue_batch<float, 1>), grid, dim3(256), 0, ctx->stream, nn, job.ps, job.batch_size,
                                   reinterpret_cast<const float *>(job.patches) + (size_t)j0 * 3 * P, SC + (size_t)b0 * 2 * P);
            else if ((P & 3) == 0 && (reinterpret_cast<uintptr_t>(job.patches) & 3) == 0) {
                // block size: the multiple of 64 (up to 256) that wastes the fewest lanes on a patch's P/4 pixel groups
                // (48x48: 576 groups = 3 blocks of 192, not 2.25 blocks of 256)
                const int groups = P / 4;
                int bs = 256, waste = (groups + 255) / 256 * 256 - groups;
                for (int cand = 192; cand >= 64; cand -= 64) {
                    const int w = (groups + cand - 1) / cand * cand - groups;
                    if (w < waste) { waste = w; bs = cand; }
                }
                dim3 grid4((unsigned)((groups + bs - 1) / bs), (unsigned)nbs);
                hipLaunchKernelGGL((k_hue_batch<uint8_t, 4>), grid4, dim3(bs), 0, ctx->stream, nn, job.ps, job.batch_size,
                                   reinterpret_cast<const uint8_t *>(job.patches) + (size_t)j0 * 3 * P, SC + (size_t)b0 * 2 * P);
            } else
                hipLaunchKernelGGL((k_hue_batch<uint8_t, 1>), grid, dim3(256), 0, ctx->stream, nn, job.ps, job.batch_size,
                                   reinterpret_cast<const uint8_t *>(job.patches) + (size_t)j0 * 3 * P, SC + (size_t)b0 * 2 * P);
        }
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (job.ps > 64) {
        ScopedTiming t(ctx, KS_COLOR);
        const unsigned wgrid = (unsigned)(job.n < 256 * 8 ? job.n : 256 * 8);
        if (job.patches_f32)
            hipLaunchKernelGGL(k_color_wide<float>, dim3(wgrid), dim3(256), 0, ctx->stream, job.n, job.ps, job.batch_size,
                               reinterpret_cast<const float *>(job.patches), job.mask_rows, SC, ctx->od_lut, job.out, job.F, col0);
        else
            hipLaunchKernelGGL(k_color_wide<uint8_t>, dim3(wgrid), dim3(256), 0, ctx->stream, job.n, job.ps, job.batch_size,
                               reinterpret_cast<const uint8_t *>(job.patches), job.mask_rows, SC, ctx->od_lut, job.out, job.F, col0);
        return hipGetLastError();
    }
    {
        ScopedTiming t(ctx, KS_COLOR);
        const int bpb = (job.batch_size + 3) / 4;
        const int64_t nb8 = (nb + 7) / 8 * 8;
        const int64_t blocks = nb8 * bpb;
        if (blocks > 0x7FFFFFFFll) return hipErrorInvalidValue;
        const size_t lds = 4 * (size_t)(P < COLOR_LIST_CAP ? P : COLOR_LIST_CAP) * sizeof(unsigned short);
        if (job.patches_f32)
            hipLaunchKernelGGL(k_color<float>, dim3((unsigned)blocks), dim3(256), lds, ctx->stream, job.n, job.ps, job.batch_size, nb, bpb,
                               reinterpret_cast<const float *>(job.patches), job.mask_rows, SC, ctx->od_lut, job.out, job.F, col0);
        else
            hipLaunchKernelGGL(k_color<uint8_t>, dim3((unsigned)blocks), dim3(256), lds, ctx->stream, job.n, job.ps, job.batch_size, nb, bpb,
                               reinterpret_cast<const uint8_t *>(job.patches), job.mask_rows, SC, ctx->od_lut, job.out, job.F, col0);
        e = hipGetLastError();
    }
    return e;
}

}  // namespace nfx
