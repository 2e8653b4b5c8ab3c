// k_geometry.hip -- polygon -> mask rasterisation and the "geometry" feature set on gfx950.
//
// Replaces (reference file:line):
//   tch_utils::shapes::polygon call sites           utils.rs:90-95,152-157
//   ShapeFeatureSet::compute_features_batched       shape.rs:16-130  (center_of_mass 219-226,
//     major_minor_axes_w_angle 141-203, eccentricity 205-207, ellipse 80-87, eliptic_deviation 209-217,
//     geometric_features::* 89-97)
//
// Layout: ONE WAVEFRONT PER NUCLEUS, ONE LANE PER MASK ROW.  A mask row (ps <= 64 pixels) is a
// single 64-bit word held in a VGPR pair, so the whole mask of a nucleus lives in the wave's
// registers: the scanline rasteriser XORs "all pixels left of the crossing" words edge by edge
// (no cross-lane traffic at all), the image moments are popcounts of the row word against six
// constant bit patterns, and only six integers per nucleus go through a wave reduction.  The mask
// never touches LDS or HBM unless another feature set needs it (then 8 B per row are stored once).
//
// Compiled with -ffp-contract=off: the mask bits and the ellipse XOR count are integer results
// that must match the CPU oracle bit for bit, so the f64/f32 expression trees below are
// evaluated exactly as written.
#include <type_traits>

#include "nfx_internal.h"
#include "wave_ops.h"

namespace nfx {

static __device__ __forceinline__ int wave_sum(int v) { return wave_total_i32(v); } // DPP, no LDS round trips

static __device__ __forceinline__ float readlane_f(float v, int l)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// ---- scanline XOR rasteriser: returns this lane's row word ------------------------------------
// pixel (r,c) sits at (c - ps/2, r - ps/2); edge (x1,y1)->(x2,y2) toggles it iff
// (y1 > py) != (y2 > py) && px < (x2-x1)*(py-y1)/(y2-y1) + x1   (all f64).
// An edge crosses only the few rows between its end points, so "every row tests every edge" would keep about
// three lanes in 64 busy.  Instead lane e takes edge e, counts the rows it crosses (py in [ymin, ymax): an exact
// integer range), a wave scan lays the (edge, row) crossings out as a work list in LDS, and the lanes then share
// the crossings evenly -- one f64 division per crossing, not per (edge, row) -- XORing "all pixels left of the
// crossing" into the row words with LDS atomics.  Lists longer than RS_CAP (zig-zag polygons) fall back to the
// every-row-tests-every-edge loop.
static constexpr int RS_CAP = 512;
struct RasterLds {
    unsigned short items[RS_CAP]; // edge << 8 | row
    unsigned lo[64], hi[64];      // row words under construction
    float ex[4][64];              // x1, y1, x2, y2 of the chunk's edges
};

// number of leading pixels of row py left of the crossing of edge (x1,y1)-(x2,y2); the caller knows the edge crosses
static __device__ __forceinline__ int crossing_count(double x1, double y1, double x2, double y2, double py, double half, int ps)
{
    const double xint = (x2 - x1) * (py - y1) / (y2 - y1) + x1;
    // k = #{c in [0,ps) : (double)c - half < xint}; estimate by ceil, then make it exact
    const double kf = ceil(xint + half);
    int k = !(kf > 0.0) ? 0 : (kf >= (double)ps ? ps : (int)kf);
    if (k > 0 && !((double)(k - 1) - half < xint)) --k;
    if (k < ps && ((double)k - half < xint)) ++k;
    return k;
}

static __device__ __forceinline__ uint64_t raster_row(const uint32_t *__restrict__ poly_off,
                                                      const float *__restrict__ poly_xy, int64_t i, int ps, int lane, RasterLds &L)
{
    const uint32_t v0 = poly_off[i];
    const int V = (int)(poly_off[i + 1] - v0);
    const double half = (double)ps / 2.0;
    uint64_t row = 0; // fallback path only
    L.lo[lane] = 0u;
    L.hi[lane] = 0u;
    for (int base = 0; base < V; base += 64) {
        const int cnt = (V - base) < 64 ? (V - base) : 64;
        float fx1 = 0.f, fy1 = 0.f, fx2 = 0.f, fy2 = 0.f;
        int ra = 0, nrows = 0;
        if (lane < cnt) {
            const int a = base + lane;
            const int b = (a + 1 == V) ? 0 : a + 1;
            const float2 p1 = reinterpret_cast<const float2 *>(poly_xy)[v0 + a];
            const float2 p2 = reinterpret_cast<const float2 *>(poly_xy)[v0 + b];
            fx1 = p1.x; fy1 = p1.y; fx2 = p2.x; fy2 = p2.y;
            // rows with ymin <= r - half < ymax, i.e. (y1 > py) != (y2 > py): r in [ceil(ymin + half), ceil(ymax + half))
            const double y1 = (double)fy1, y2 = (double)fy2;
            const double ylo = y1 < y2 ? y1 : y2, yhi = y1 < y2 ? y2 : y1; // a NaN leaves an empty range below
            double da = ceil(ylo + half), db = ceil(yhi + half);
            da = da > 0.0 ? da : 0.0; db = db > 0.0 ? db : 0.0;
            da = da < (double)ps ? da : (double)ps; db = db < (double)ps ? db : (double)ps;
            ra = (int)da;
            nrows = (int)db - ra;
            nrows = (nrows > 0 && ylo < yhi) ? nrows : 0;
        }
        const int incl = wave_inclusive_scan_i32(nrows);
        const int total = __builtin_amdgcn_readlane(incl, 63);
        if (total > RS_CAP) {
            const double py = (double)lane - half;
            for (int e = 0; e < cnt; ++e) {
                const double x1 = (double)readlane_f(fx1, e), y1 = (double)readlane_f(fy1, e);
                const double x2 = (double)readlane_f(fx2, e), y2 = (double)readlane_f(fy2, e);
                if ((y1 > py) != (y2 > py)) {
                    const int k = crossing_count(x1, y1, x2, y2, py, half, ps);
                    row ^= (k >= 64) ? ~0ull : ((1ull << k) - 1ull);
                }
            }
            continue;
        }
        L.ex[0][lane] = fx1; L.ex[1][lane] = fy1; L.ex[2][lane] = fx2; L.ex[3][lane] = fy2;
        {
            const int first = incl - nrows;
            int longest = nrows;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(longest, o, 64); longest = v > longest ? v : longest; }
            for (int j = 0; j < longest; ++j)
                if (j < nrows) L.items[first + j] = (unsigned short)((lane << 8) | (ra + j));
        }
        wave_lds_fence();
        for (int w = lane; w < total; w += 64) {
            const unsigned it = L.items[w];
            const int e = (int)(it >> 8), r = (int)(it & 255u);
            const int k = crossing_count((double)L.ex[0][e], (double)L.ex[1][e], (double)L.ex[2][e], (double)L.ex[3][e], (double)r - half, half, ps);
            const uint64_t span = (k >= 64) ? ~0ull : ((1ull << k) - 1ull);
            atomicXor(&L.lo[r], (unsigned)span);
            atomicXor(&L.hi[r], (unsigned)(span >> 32));
        }
        wave_lds_fence();
    }
    wave_lds_fence();
    row ^= ((uint64_t)L.hi[lane] << 32) | (uint64_t)L.lo[lane];
    return lane < ps ? row : 0ull;
}

// ---- LAPACK slanv2 (f32) as reached by sgeev on a 2x2 block; see DESIGN.md "orientation" -------
static __device__ __forceinline__ float sign1(float x) { return __builtin_signbit(x) ? -1.0f : 1.0f; }

static __device__ __forceinline__ float slapy2(float x, float y)
{
    const float xa = fabsf(x), ya = fabsf(y);
    const float w = xa > ya ? xa : ya;
    const float z = xa > ya ? ya : xa;
    if (z == 0.0f) return w;
    const float t = z / w;
    return w * sqrtf(1.0f + t * t);
}

static __device__ __forceinline__ void slanv2(float a, float b, float c, float d, float &rt1, float &rt2, float &cs, float &sn)
{
    const float eps = 1.1920929e-07f, multpl = 4.0f;
    if (c == 0.0f) {
        cs = 1.0f; sn = 0.0f;
    } else if (b == 0.0f) {
        cs = 0.0f; sn = 1.0f;
        const float t = d; d = a; a = t; b = -c; c = 0.0f;
    } else if ((a - d) == 0.0f && sign1(b) != sign1(c)) {
        cs = 1.0f; sn = 0.0f;
    } else {
        float temp = a - d;
        const float p = 0.5f * temp;
        const float bcmax = fmaxf(fabsf(b), fabsf(c));
        const float bcmis = fminf(fabsf(b), fabsf(c)) * sign1(b) * sign1(c);
        const float scale = fmaxf(fabsf(p), bcmax);
        float z = p / scale * p + bcmax / scale * bcmis;
        if (z >= multpl * eps) {
            const float sq = sqrtf(scale) * sqrtf(z);
            z = p + (__builtin_signbit(p) ? -fabsf(sq) : fabsf(sq));
            a = d + z;
            d = d - (bcmax / z) * bcmis;
            const float tau = slapy2(c, z);
            cs = z / tau;
            sn = c / tau;
            b = b - c;
            c = 0.0f;
        } else {
            const float sigma = b + c;
            float tau = slapy2(sigma, temp);
            cs = sqrtf(0.5f * (1.0f + fabsf(sigma) / tau));
            sn = -(p / (tau * cs)) * sign1(sigma);
            const float aa = a * cs + b * sn, bb = -a * sn + b * cs;
            const float cc = c * cs + d * sn, dd = -c * sn + d * cs;
            a = aa * cs + cc * sn; b = bb * cs + dd * sn;
            c = -aa * sn + cc * cs; d = -bb * sn + dd * cs;
            temp = 0.5f * (a + d);
            a = temp; d = temp;
            if (c != 0.0f) {
                if (b != 0.0f) {
                    if (sign1(b) == sign1(c)) {
                        const float sab = sqrtf(fabsf(b)), sac = sqrtf(fabsf(c));
                        const float pp = __builtin_signbit(c) ? -fabsf(sab * sac) : fabsf(sab * sac);
                        tau = 1.0f / sqrtf(fabsf(b + c));
                        a = temp + pp; d = temp - pp;
                        b = b - c; c = 0.0f;
                        const float cs1 = sab * tau, sn1 = sac * tau;
                        temp = cs * cs1 - sn * sn1;
                        sn = cs * sn1 + sn * cs1;
                        cs = temp;
                    }
                } else {
                    b = -c; c = 0.0f;
                    temp = cs; cs = -sn; sn = temp;
                }
            }
        }
    }
    rt1 = a; rt2 = d;
}

// ---- mask-derived geometry columns ------------------------------------------------------------------------------------
// per-word moments by popcount against the bit patterns "bit index has bit j set": count, sum of bit indices, sum of squares
static __device__ __forceinline__ void word_moments(uint64_t w, int &pc, int &sc, int &scc)
{
    constexpr uint64_t PAT[6] = {0xAAAAAAAAAAAAAAAAull, 0xCCCCCCCCCCCCCCCCull, 0xF0F0F0F0F0F0F0F0ull,
                                 0xFF00FF00FF00FF00ull, 0xFFFF0000FFFF0000ull, 0xFFFFFFFF00000000ull};
    pc = __popcll(w);
    sc = 0; scc = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const uint64_t mj = w & PAT[j];
        const int pj = __popcll(mj);
        sc += pj << j;
        scc += pj << (2 * j);
#pragma unroll
        for (int k = j + 1; k < 6; ++k) scc += __popcll(mj & PAT[k]) << (j + k + 1);
    }
}

// center_of_mass, major_minor_axes_w_angle, eccentricity (shape.rs:141-226) from the six integer moments of the mask
struct ShapeAxes {
    float major, minor, angle, ecc, cen0, cen1; // cen0 / cen1: centre of mass (row, column) minus ps / 2 (shape.rs:71-73)
};

// (I = int for patches up to 128 x 128, whose sums stay below 2^31; long long above)
template <typename I>
static __device__ __forceinline__ ShapeAxes shape_axes(I K, I sr, I scs, I srr, I src, I sccs, int ps)
{
    ShapeAxes A;
    // center_of_mass (shape.rs:219-226): integer sums < 2^24 are exact in f32
    const float mu_r = (float)sr / (float)K; // K == 0 -> NaN
    const float mu_c = (float)scs / (float)K;
    const float halfps = (float)ps / 2.0f;
    A.cen0 = mu_r - halfps;
    A.cen1 = mu_c - halfps;
    float major, minor, angle;
    if (K == 0) {
        major = minor = angle = __builtin_nanf("");
    } else {
        // covariance about the f32 means: exact integer moments, f64 closed form, rounded once
        const double mr = (double)mu_r, mc = (double)mu_c, kd = (double)K;
        const double Crr = (double)srr - 2.0 * mr * (double)sr + kd * mr * mr;
        const double Crc = (double)src - mr * (double)scs - mc * (double)sr + kd * mr * mc;
        const double Ccc = (double)sccs - 2.0 * mc * (double)scs + kd * mc * mc;
        const float p = (float)Crr / (float)K;
        const float q = (float)Crc / (float)K;
        const float r = (float)Ccc / (float)K;
        float w0, w1, cs, sn;
        if (q == 0.0f) { w0 = p; w1 = r; cs = 1.0f; sn = 0.0f; }
        else slanv2(p, q, q, r, w0, w1, cs, sn);
        // eigenvectors V = [[cs,-sn],[sn,cs]]; the reference reads a ROW of V (shape.rs:191-195)
        float mx, my, ea, eb;
        if (w0 > w1) { ea = sqrtf(w0); eb = sqrtf(w1); mx = cs; my = -sn; }
        else         { ea = sqrtf(w1); eb = sqrtf(w0); mx = sn; my = cs; }
        angle = (float)atan2((double)mx, (double)my);
        major = ea * 2.0f;
        minor = eb * 2.0f;
    }
    const float hM = major * 0.5f, hm = minor * 0.5f;
    A.major = major; A.minor = minor; A.angle = angle;
    A.ecc = sqrtf(hM * hM - hm * hm) / hM;
    return A;
}

// ellipse (shape.rs:80-87), one row: the columns that pass the reference test form an interval [lo, lo + len) (the ellipse is
// convex).  Its ends are estimated from the quadratic in x and then made EXACT by evaluating the reference test itself at the
// ends, so the result equals the per-pixel evaluation bit for bit at ~6 tests per row instead of ps.  len = 0: empty row.
struct EllipseFrame {
    double ct, st, half, xc, yc, a, b;
};
static __device__ __forceinline__ EllipseFrame ellipse_frame(const ShapeAxes &A, int ps)
{
    EllipseFrame E;
    E.ct = cos((double)A.angle); E.st = sin((double)A.angle);
    E.half = (double)ps / 2.0;
    E.xc = (double)A.cen1; E.yc = (double)A.cen0; E.a = (double)A.major; E.b = (double)A.minor;
    return E;
}
static __device__ __forceinline__ void ellipse_row_span(const EllipseFrame &E, int ps, int row, int &lo, int &len)
{
    lo = 0; len = 0;
    const double ct = E.ct, st = E.st, half = E.half, xc = E.xc, yc = E.yc, a = E.a, b = E.b;
    if (!(a > 0.0 && b > 0.0)) return; // a or b zero/NaN: every test is false
    const double y = ((double)row - half) - yc;
    auto inside = [&](int c) -> bool {
        const double x = ((double)c - half) - xc;
        const double xrr = x * ct + y * st;
        const double yrr = y * ct - x * st;
        const double u = xrr / a, v = yrr / b;
        return (u * u + v * v) <= 1.0;
    };
    const double ia2 = 1.0 / (a * a), ib2 = 1.0 / (b * b);
    const double qa = ct * ct * ia2 + st * st * ib2;
    const double qb = 2.0 * y * ct * st * (ia2 - ib2);
    const double qc = y * y * (st * st * ia2 + ct * ct * ib2) - 1.0;
    const double disc = qb * qb - 4.0 * qa * qc;
    const double shift = half + xc; // column = x + shift
    const double xv = -qb / (2.0 * qa);
    auto clampc = [&](double f) -> int { return f < 0.0 ? 0 : (f > (double)(ps - 1) ? ps - 1 : (int)f); };
    // a pixel inside the interval, if there is one: the column nearest to the quadratic's vertex (or its
    // neighbours, when rounding put the vertex on the other side of a pixel centre)
    const int cm = clampc(floor(xv + shift + 0.5));
    int seed = -1;
#pragma unroll 1
    for (int d = 0; d < 3 && seed < 0; ++d) {
        const int c = cm + (d == 0 ? 0 : (d == 1 ? -1 : 1));
        if (c >= 0 && c < ps && inside(c)) seed = c;
    }
    if (seed < 0) return;
    const double sq = disc > 0.0 ? sqrt(disc) / (2.0 * qa) : 0.0;
    int bound[2];
#pragma unroll 1
    for (int side = 0; side < 2; ++side) {
        const int dir = side == 0 ? -1 : 1;
        int e = side == 0 ? clampc(ceil(xv - sq + shift)) : clampc(floor(xv + sq + shift));
        e = side == 0 ? (e < seed ? e : seed) : (e > seed ? e : seed);
        bool grow = false; // first walk towards the seed until inside, then outwards while inside
        for (;;) {
            if (!grow && e == seed) grow = true;
            const int probe = grow ? e + dir : e;
            if (grow && (probe < 0 || probe >= ps)) break;
            const bool in = inside(probe);
            if (!grow) { if (in) grow = true; else e -= dir; }
            else { if (in) e = probe; else break; }
        }
        bound[side] = e;
    }
    lo = bound[0];
    len = bound[1] - bound[0] + 1;
}

// one wave per nucleus, lane = row (ps <= 64): writes out[1]=major_axis [2]=minor_axis [3]=eccentricity [4]=orientation
// [8]=eliptic_deviation
static __device__ __forceinline__ void shape_from_row(uint64_t row, int ps, int lane, float *__restrict__ o)
{
    int kr, sc, scc;
    word_moments(row, kr, sc, scc);
    // seven totals in one reduce-scatter butterfly (wave_ops.h); lane l holds the total of M[scatter_index<8>(l & 7)]
    unsigned M[8] = {(unsigned)kr, (unsigned)(lane * kr), (unsigned)sc, (unsigned)(lane * lane * kr), (unsigned)(lane * sc), (unsigned)scc, 0u, 0u};
    const int tot = (int)wave_reduce_scatter(M, lane);
    // scatter_index<8> reverses three bits: slots 0..5 sit in lanes 0, 4, 2, 6, 1, 5
    const int K = __builtin_amdgcn_readlane(tot, 0), sr = __builtin_amdgcn_readlane(tot, 4), scs = __builtin_amdgcn_readlane(tot, 2);
    const int srr = __builtin_amdgcn_readlane(tot, 6), src = __builtin_amdgcn_readlane(tot, 1), sccs = __builtin_amdgcn_readlane(tot, 5);
    const ShapeAxes A = shape_axes(K, sr, scs, srr, src, sccs, ps);
    // ellipse (shape.rs:80-87) XOR mask, one row per lane
    int xr = 0;
    {
        const EllipseFrame E = ellipse_frame(A, ps);
        uint64_t ell = 0;
        if (lane < ps) {
            int lo, len;
            ellipse_row_span(E, ps, lane, lo, len);
            if (len > 0) ell = (len >= 64 ? ~0ull : ((1ull << len) - 1ull)) << lo;
        }
        xr = wave_sum(__popcll(ell ^ row));
    }
    if (lane == 0) {
        o[1] = A.major;
        o[2] = A.minor;
        o[3] = A.ecc;
        o[4] = A.angle;
        o[8] = (float)xr / (float)K;
    }
}

// (eight waves per SIMD: with the shape arithmetic inlined the kernel needs 62 registers, and it is a chain of dependent f64
// divisions and transcendental expansions that only other waves can hide: -11 % against five waves at 86 registers)
template <bool SHAPE>
__global__ __launch_bounds__(256, 8) void k_raster_shape(int64_t n, const uint32_t *__restrict__ poly_off,
                                                      const float *__restrict__ poly_xy, int ps,
                                                      uint64_t *__restrict__ rows_out, float *__restrict__ out, int F,
                                                      int col0, int *__restrict__ areas_out)
{
    __shared__ RasterLds s_raster[4];
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return; // wave-uniform
    const uint64_t row = raster_row(poly_off, poly_xy, i, ps, lane, s_raster[threadIdx.x >> 6]);
    if (rows_out && lane < ps) rows_out[i * ps + lane] = row;
    if (areas_out) { // the pixel kernels size their work by the mask's area: handed over instead of re-read from the rows
        const int K = wave_sum(__popcll(row));
        if (lane == 0) areas_out[i] = K;
    }
    if (SHAPE) shape_from_row(row, ps, lane, out + i * (int64_t)F + col0);
}

__global__ __launch_bounds__(256) void k_shape_from_rows(int64_t n, const uint64_t *__restrict__ rows, int ps,
                                                         float *__restrict__ out, int F, int col0)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const uint64_t row = lane < ps ? rows[i * ps + lane] : 0ull;
    shape_from_row(row, ps, lane, out + i * (int64_t)F + col0);
}

// ---- patch sizes 65..128 (args.rs:93-94 takes any u32; `-p 128` works in the reference) ---------------------------------
// The row-word kernels above hold a mask row in one lane's 64-bit word.  Larger patches take this plainer family (here and in
// the other k_*.hip files): ONE 256-THREAD WORKGROUP PER NUCLEUS, thread = mask row, a row is W = 2 words.  Same arithmetic,
// same expression trees (the pieces above are shared), no attempt at the narrow path's speed: every row tests every edge,
// integer moments and the ellipse XOR count are reduced through LDS atomics.  Row words go out as [n][ps][W].
static constexpr int WIDE_W = 2;
static __device__ __forceinline__ uint64_t low_bits(int k) { return k <= 0 ? 0ull : (k >= 64 ? ~0ull : ((1ull << k) - 1ull)); }

template <bool SHAPE>
__global__ __launch_bounds__(256) void k_raster_shape_wide(int64_t n, const uint32_t *__restrict__ poly_off, const float *__restrict__ poly_xy,
                                                           int ps, uint64_t *__restrict__ rows_out, float *__restrict__ out, int F, int col0,
                                                           int *__restrict__ areas_out)
{
    __shared__ int s_tot[8];
    const int t = threadIdx.x;
    for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
        if (t < 8) s_tot[t] = 0;
        __syncthreads();
        const uint32_t v0 = poly_off[i];
        const int V = (int)(poly_off[i + 1] - v0);
        const double half = (double)ps / 2.0;
        uint64_t row[WIDE_W] = {0ull, 0ull};
        if (t < ps) {
            const double py = (double)t - half;
            for (int e = 0; e < V; ++e) {
                const float2 p1 = reinterpret_cast<const float2 *>(poly_xy)[v0 + e];
                const float2 p2 = reinterpret_cast<const float2 *>(poly_xy)[v0 + (e + 1 == V ? 0 : e + 1)];
                const double x1 = (double)p1.x, y1 = (double)p1.y, x2 = (double)p2.x, y2 = (double)p2.y;
                if ((y1 > py) != (y2 > py)) {
                    const int k = crossing_count(x1, y1, x2, y2, py, half, ps);
                    row[0] ^= low_bits(k);
                    row[1] ^= low_bits(k - 64);
                }
            }
            if (rows_out) { rows_out[(i * ps + t) * WIDE_W] = row[0]; rows_out[(i * ps + t) * WIDE_W + 1] = row[1]; }
            int p0, s0, q0, p1, s1, q1;
            word_moments(row[0], p0, s0, q0);
            word_moments(row[1], p1, s1, q1);
            const int kr = p0 + p1, sc = s0 + s1 + 64 * p1, scc = q0 + q1 + 128 * s1 + 4096 * p1;
            if (kr) { // sums stay below 2^31 for ps <= 128: 128^2 pixels x 127^2
                atomicAdd(&s_tot[0], kr); atomicAdd(&s_tot[1], t * kr); atomicAdd(&s_tot[2], sc);
                atomicAdd(&s_tot[3], t * t * kr); atomicAdd(&s_tot[4], t * sc); atomicAdd(&s_tot[5], scc);
            }
        }
        __syncthreads();
        const int K = s_tot[0];
        if (areas_out && t == 0) areas_out[i] = K;
        if (SHAPE) {
            const ShapeAxes A = shape_axes(K, s_tot[1], s_tot[2], s_tot[3], s_tot[4], s_tot[5], ps);
            if (t < ps) {
                const EllipseFrame E = ellipse_frame(A, ps);
                int lo, len;
                ellipse_row_span(E, ps, t, lo, len);
                const uint64_t e0 = low_bits(lo + len) & ~low_bits(lo), e1 = low_bits(lo + len - 64) & ~low_bits(lo - 64);
                const int x = __popcll(e0 ^ row[0]) + __popcll(e1 ^ row[1]);
                if (x) atomicAdd(&s_tot[6], x);
            }
            __syncthreads();
            if (t == 0) {
                float *o = out + i * (int64_t)F + col0;
                o[1] = A.major; o[2] = A.minor; o[3] = A.ecc; o[4] = A.angle;
                o[8] = (float)s_tot[6] / (float)K;
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_shape_from_rows_wide(int64_t n, const uint64_t *__restrict__ rows, int ps, float *__restrict__ out, int F, int col0)
{
    __shared__ int s_tot[8];
    const int t = threadIdx.x;
    for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
        if (t < 8) s_tot[t] = 0;
        __syncthreads();
        uint64_t row[WIDE_W] = {0ull, 0ull};
        if (t < ps) {
            row[0] = rows[(i * ps + t) * WIDE_W]; row[1] = rows[(i * ps + t) * WIDE_W + 1];
            int p0, s0, q0, p1, s1, q1;
            word_moments(row[0], p0, s0, q0);
            word_moments(row[1], p1, s1, q1);
            const int kr = p0 + p1, sc = s0 + s1 + 64 * p1, scc = q0 + q1 + 128 * s1 + 4096 * p1;
            if (kr) {
                atomicAdd(&s_tot[0], kr); atomicAdd(&s_tot[1], t * kr); atomicAdd(&s_tot[2], sc);
                atomicAdd(&s_tot[3], t * t * kr); atomicAdd(&s_tot[4], t * sc); atomicAdd(&s_tot[5], scc);
            }
        }
        __syncthreads();
        const int K = s_tot[0];
        const ShapeAxes A = shape_axes(K, s_tot[1], s_tot[2], s_tot[3], s_tot[4], s_tot[5], ps);
        if (t < ps) {
            const EllipseFrame E = ellipse_frame(A, ps);
            int lo, len;
            ellipse_row_span(E, ps, t, lo, len);
            const uint64_t e0 = low_bits(lo + len) & ~low_bits(lo), e1 = low_bits(lo + len - 64) & ~low_bits(lo - 64);
            const int x = __popcll(e0 ^ row[0]) + __popcll(e1 ^ row[1]);
            if (x) atomicAdd(&s_tot[6], x);
        }
        __syncthreads();
        if (t == 0) {
            float *o = out + i * (int64_t)F + col0;
            o[1] = A.major; o[2] = A.minor; o[3] = A.ecc; o[4] = A.angle;
            o[8] = (float)s_tot[6] / (float)K;
        }
        __syncthreads();
    }
}

// ---- polygon-derived columns (geometric_features::*, shape.rs:89-97) -----------------------------
// polygon_one is the serial program for ONE ring (shoelace area, ring perimeter, natural merge sort, Andrew's monotone
// chain in f64 over f32 vertices widened on the fly -- exact): the reference statement of the columns and the path of
// rings longer than PG_MAXV vertices.  Typical rings go through k_polygon's group-parallel form below.
struct pt { double x, y; };

static __device__ __forceinline__ bool pt_less(const pt &a, const pt &b)
{
    return (a.x < b.x) || (a.x == b.x && a.y < b.y);
}
static __device__ __forceinline__ double cross3(const pt &o, const pt &a, const pt &b)
{
    return (a.x - o.x) * (b.y - o.y) - (a.y - o.y) * (b.x - o.x);
}

static constexpr int PG_MAXV = 64;                 // ring length served from LDS
#ifndef NFX_PG_LPR
#define NFX_PG_LPR 8 // lanes per ring of the group-parallel kernel = partial sums of every edge total (both kernels)
#endif

// the butterfly of group_total_f64 below, on one lane's array of partials
static __device__ __forceinline__ double partial_tree(const double (&p)[NFX_PG_LPR])
{
    double q[NFX_PG_LPR];
#pragma unroll
    for (int j = 0; j < NFX_PG_LPR; ++j) q[j] = p[j];
#pragma unroll
    for (int w = 1; w < NFX_PG_LPR; w *= 2)
#pragma unroll
        for (int j = 0; j < NFX_PG_LPR; j += 2 * w) q[j] = q[j] + q[j + w];
    return q[0];
}

size_t polygon_scratch_bytes(int64_t n, int64_t total_vertices)
{
    // long rings only: sort permutation + hull stack as 16-bit... kept simple: one u32 per slot
    return sizeof(uint32_t) * (size_t)(3 * total_vertices + 4 * n + 16);
}

template <typename IDX, typename VP>
static __device__ __forceinline__ void polygon_one(VP v, int V, IDX *srt, IDX *hull, float *__restrict__ o, double *__restrict__ o64)
{
    auto P = [&](int k) { const float2 f = v[k]; return pt{(double)f.x, (double)f.y}; };
    const double PI = 3.14159265358979323846;
    // shoelace + perimeter.  SUMMATION ORDER (shared with the group-parallel k_polygon below, so that the same ring gives the same
    // bits whichever kernel a call happens to take): edge k goes to partial k mod NFX_PG_LPR, every partial is summed in ring
    // order, the partials are joined by the butterfly ((p0+p1)+(p2+p3))+((p4+p5)+(p6+p7)).
    double sh_p[NFX_PG_LPR], pe_p[NFX_PG_LPR];
#pragma unroll
    for (int j = 0; j < NFX_PG_LPR; ++j) { sh_p[j] = 0.0; pe_p[j] = 0.0; }
    if (V > 0) {
        const pt first = P(0);
        for (int k0 = 0; k0 < V; k0 += NFX_PG_LPR) {
#pragma unroll
            for (int j = 0; j < NFX_PG_LPR; ++j) {
                const int k = k0 + j;
                if (k < V) {
                    const pt a = P(k), b = (k + 1 == V) ? first : P(k + 1);
                    sh_p[j] += a.x * b.y - b.x * a.y;
                    const double dx = b.x - a.x, dy = b.y - a.y;
                    pe_p[j] += sqrt(dx * dx + dy * dy);
                }
            }
        }
    }
    const double sh = partial_tree(sh_p), perim = partial_tree(pe_p);
    const double area = fabs(sh) * 0.5;
    const double eq_perim = 2.0 * sqrt(PI * area);
    const double compacity = 4.0 * PI * area / (perim * perim);

    // NATURAL MERGE SORT of the indices (lexicographic x then y; ties keep ring order, as a stable sort of the points would).
    // A ring listed around its centre is two or three monotone runs in x, so the whole sort is one or two linear merges
    // (~2V dependent steps; the insertion sort it replaces paid one per inversion, ~V^2/4).
    // Phase 1 cuts the ring into maximal runs -- non-descending ones as they are, STRICTLY descending ones reversed, which
    // cannot reorder equal points -- and notes their ends; phase 2 merges neighbouring runs, left run first on ties, between
    // the index array and the (still unused) hull array until one run is left.
    {
        auto F2 = [&](int k) { return v[k]; };
        auto less = [](const float2 &a, const float2 &b) { return (a.x < b.x) || (a.x == b.x && a.y < b.y); }; // f32: widening is exact
        IDX *src = srt, *dst = hull, *ends = hull + V; // run r ends (exclusive) at ends[r]
        int nr = 0;
        for (int k = 0; k < V;) {
            int e = k + 1;
            float2 prev = F2(k);
            if (e < V && less(F2(e), prev)) {
                for (; e < V; ++e) { const float2 c = F2(e); if (!less(c, prev)) break; prev = c; }
                for (int t = 0; t < e - k; ++t) src[k + t] = (IDX)(e - 1 - t);
            } else {
                for (; e < V; ++e) { const float2 c = F2(e); if (less(c, prev)) break; prev = c; }
                for (int t = k; t < e; ++t) src[t] = (IDX)t;
            }
            ends[nr++] = (IDX)e;
            k = e;
        }
        while (nr > 1) {
            int lo = 0, nout = 0;
            for (int r = 0; r < nr; r += 2) {
                const int mid = (int)ends[r], hi = r + 1 < nr ? (int)ends[r + 1] : mid;
                int i = lo, j = mid, o = lo;
                if (i < mid && j < hi) {
                    IDX ia = src[i], ib = src[j];
                    float2 pa = F2((int)ia), pb = F2((int)ib);
                    for (;;) {
                        if (less(pb, pa)) {
                            dst[o++] = ib;
                            if (++j == hi) break;
                            ib = src[j]; pb = F2((int)ib);
                        } else {
                            dst[o++] = ia;
                            if (++i == mid) break;
                            ia = src[i]; pa = F2((int)ia);
                        }
                    }
                }
                for (; i < mid; ++i) dst[o++] = src[i];
                for (; j < hi; ++j) dst[o++] = src[j];
                // the merged run's end goes to the front of the list: slot nout <= r has been consumed already.  When the
                // destination is the hull array the ends live right behind its first V slots, which stay untouched
                ends[nout++] = (IDX)hi;
                lo = hi;
            }
            nr = nout;
            IDX *t = src; src = dst; dst = t;
        }
        if (src != srt)
            for (int k = 0; k < V; ++k) srt[k] = src[k];
    }
    int m = 0;
    for (int k = 0; k < V; ++k) {
        const pt c = P(srt[k]);
        while (m >= 2 && cross3(P(hull[m - 2]), P(hull[m - 1]), c) <= 0.0) --m;
        hull[m++] = srt[k];
    }
    const int lower = m + 1;
    for (int k = V - 2; k >= 0; --k) {
        const pt c = P(srt[k]);
        while (m >= lower && cross3(P(hull[m - 2]), P(hull[m - 1]), c) <= 0.0) --m;
        hull[m++] = srt[k];
    }
    if (m > 1) --m;
    // hull area and perimeter in the same order as k_polygon: the hull is the lower chain (hull[0 .. nlow-1], nlow-1 edges) followed by
    // the upper chain (its edges continue from hull[nlow-1] back to hull[0]); edge e of a CHAIN goes to partial e mod NFX_PG_LPR, lower
    // chain first.
    double hs_p[NFX_PG_LPR], hp_p[NFX_PG_LPR];
#pragma unroll
    for (int j = 0; j < NFX_PG_LPR; ++j) { hs_p[j] = 0.0; hp_p[j] = 0.0; }
    const int nlow = lower - 1;          // points of the lower chain
    if (m > 1 || (m == 1 && V > 1)) {
        for (int c0 = 0; c0 < 2; ++c0) {
            const int e_begin = c0 == 0 ? 0 : nlow - 1, e_end = c0 == 0 ? nlow - 1 : m; // hull edges k -> k + 1 (mod m) of this chain
            for (int k0 = e_begin; k0 < e_end; k0 += NFX_PG_LPR) {
#pragma unroll
                for (int j = 0; j < NFX_PG_LPR; ++j) {
                    const int k = k0 + j;
                    if (k < e_end) {
                        const pt a = P(hull[k]), b = P(hull[k + 1 == m ? 0 : k + 1]);
                        hs_p[j] += a.x * b.y - b.x * a.y;
                        const double dx = b.x - a.x, dy = b.y - a.y;
                        hp_p[j] += sqrt(dx * dx + dy * dy);
                    }
                }
            }
        }
    }
    const double hsh = partial_tree(hs_p), hperim = partial_tree(hp_p);
    const double harea = fabs(hsh) * 0.5;
    const double deviation = (harea - area) / harea;
    if (o) { // columns of the geometry block (shape.rs:116-127), rounded once
        o[0] = (float)area;
        o[5] = (float)perim;
        o[6] = (float)eq_perim;
        o[7] = (float)compacity;
        o[9] = (float)harea;
        o[10] = (float)deviation;
        o[11] = (float)hperim;
    }
    if (o64) { // nfx_polygon_features_f64: the same seven values at geometric_features' own width
        o64[0] = area; o64[1] = perim; o64[2] = eq_perim; o64[3] = compacity; o64[4] = harea; o64[5] = deviation; o64[6] = hperim;
    }
}

// ---- one lane per ring ----------------------------------------------------------------------------------------------
// The serial program, 64 rings per wave: about 40 wave-instructions per ring -- twenty times fewer issue slots than the
// group-parallel kernel below -- but a chain of dependent LDS reads on 64 diverging control paths, at 5 waves per CU: 92 %
// of its time is waiting.  That makes it the right kernel BESIDE other work (the side stream of a call that also runs
// pixel sets: it hides completely and takes almost nothing from the kernels it runs next to) and the wrong one when the
// polygon columns are on the critical path (geometry alone), where the group-parallel kernel is 1.6x faster.
static constexpr int PG_CAPV = 2432;               // vertices of one wave's 64 rings kept in LDS (38 per ring on average)
static constexpr int PG_SLICE = 3 * PG_MAXV + 8;   // sort permutation (V) + hull stack (2V+2), one byte each

__global__ __launch_bounds__(64) void k_polygon_serial(int64_t n, const uint32_t *__restrict__ poly_off,
                                                       const float *__restrict__ poly_xy, uint32_t *__restrict__ scratch,
                                                       float *__restrict__ out, int F, int col0, double *__restrict__ out64)
{
    // The rings of the wave's 64 nuclei are copied once into LDS, packed back to back (a wave scan of the ring lengths gives
    // every lane its slice of the vertex and of the index arrays), so the sort's and the hull walk's dependent reads are LDS
    // reads.  A wave whose rings do not all fit (more than PG_CAPV vertices in total, or a ring longer than PG_MAXV) reads
    // them in place, with per-lane index slices of the maximum size.
    __shared__ __attribute__((aligned(8))) uint8_t s_raw[PG_CAPV * 8 + 3 * PG_CAPV + 64 * 8];
    static_assert(sizeof(s_raw) >= 64 * PG_SLICE, "the in-place path's index slices fit too");
    float2 *s_xy = reinterpret_cast<float2 *>(s_raw);
    uint8_t *s_idx = s_raw + PG_CAPV * 8;
    const int lane = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * 64 + lane;
    const bool live = i < n;
    const uint32_t v0 = live ? poly_off[i] : 0u;
    const int V = live ? (int)(poly_off[i + 1] - v0) : 0;
    const int incl = wave_inclusive_scan_i32(V);
    const int first = incl - V;
    const float2 *gv = reinterpret_cast<const float2 *>(poly_xy) + v0;
    const bool packed = __all(V <= PG_MAXV && incl <= PG_CAPV); // wave-uniform
    float *o = out ? out + i * (int64_t)F + col0 : nullptr;
    double *o64 = out64 ? out64 + i * 7 : nullptr;
    if (packed) {
        for (int k = 0; k < V; ++k) s_xy[first + k] = gv[k];
        if (!live) return;
        uint8_t *srt = s_idx + 3 * first + 8 * lane; // V sort indices, then 2V + 2 hull slots
        polygon_one<uint8_t, const float2 *>(s_xy + first, V, srt, srt + V, o, o64);
    } else {
        if (!live) return;
        if (V <= PG_MAXV) {
            uint8_t *srt = s_raw + lane * PG_SLICE;
            polygon_one<uint8_t, const float2 *__restrict__>(gv, V, srt, srt + PG_MAXV, o, o64);
        } else {
            uint32_t *srt = scratch + 3 * (size_t)v0 + 4 * (size_t)i;
            polygon_one<uint32_t, const float2 *__restrict__>(gv, V, srt, srt + V, o, o64);
        }
    }
}

// ---- LPR lanes per ring --------------------------------------------------------------------------------------------
// For calls whose critical path the polygon columns are on.  A ring belongs to a GROUP of LPR lanes (64 / LPR rings per
// wave, four waves per block), so the chip holds 4-16x as many waves for the same rings and most of the work is lane-parallel:
//   * the vertices go to LDS once (ring stride padded so the groups' rings start on different banks) and stay in the
//     lanes' registers (lane `sub` holds vertices sub, sub + LPR, ...);
//   * shoelace and perimeter are per-edge terms summed over the group (f64; DPP butterflies inside the row of 16);
//   * the lexicographic sort is a RANK sort: vertex t is broadcast from LDS and every lane counts, for each of its own
//     vertices, whether t sorts before it (ties by ring index: the permutation a stable sort gives) -- V steps of
//     independent compares instead of a dependent merge chain; the permutation is scattered to LDS as byte indices;
//   * only Andrew's monotone chain stays serial: two lanes of the group walk its lower and its upper chain side by side,
//     with the top two hull points in registers (a push costs no LDS read, a pop one);
//   * hull area and perimeter are again per-edge terms over the group.
// Rings longer than PG_MAXV keep the serial program (polygon_one on global memory), run by their group's first lane.
template <int LPR>
static __device__ __forceinline__ double group_total_f64(double v)
{
    v += xchg<1>(v);
    v += xchg<2>(v);
    if constexpr (LPR >= 8) v += xchg<4>(v);
    if constexpr (LPR >= 16) v += xchg<8>(v);
    return v;
}

template <int LPR>
__global__ __launch_bounds__(256) void k_polygon(int64_t n, const uint32_t *__restrict__ poly_off,
                                                 const float *__restrict__ poly_xy, uint32_t *__restrict__ scratch,
                                                 float *__restrict__ out, int F, int col0, double *__restrict__ out64)
{
    static_assert(LPR == 4 || LPR == 8 || LPR == 16, "a group is an aligned part of a DPP row");
    constexpr int R = 64 / LPR;           // rings per wave
    constexpr int E = PG_MAXV / LPR;      // vertex slots per lane
    constexpr int RSTRIDE = PG_MAXV + 1;  // float2 per ring: 130 words, so ring g starts on bank 2g (8-byte reads: 64 banks)
    __shared__ float2 s_pts[4][R * RSTRIDE];
    __shared__ uint8_t s_ord[4][R][PG_MAXV];
    __shared__ uint8_t s_hull[4][R][2][PG_MAXV + 4]; // lower chain | upper chain
    __shared__ int s_m[4][R][2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int g = lane / LPR, sub = lane % LPR;
    const int64_t i = ((int64_t)blockIdx.x * 4 + wv) * R + g;
    const bool live = i < n;
    const uint32_t v0 = live ? poly_off[i] : 0u;
    const int Vraw = live ? (int)(poly_off[i + 1] - v0) : 0;
    const bool big = Vraw > PG_MAXV;
    const int V = big ? 0 : Vraw; // a long ring takes no part in the group work below
    const float2 *gv = reinterpret_cast<const float2 *>(poly_xy) + v0;
    float2 *pts = s_pts[wv] + g * RSTRIDE;
    uint8_t *ord = s_ord[wv][g];
    float *o = (live && out) ? out + i * (int64_t)F + col0 : nullptr;
    double *o64 = (live && out64) ? out64 + i * 7 : nullptr;
    const double PI = 3.14159265358979323846;

    // ---- vertices: registers and LDS --------------------------------------------------------------------------------
    float px[E], py[E];
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const int k = sub + LPR * j;
        float2 p = make_float2(0.0f, 0.0f);
        if (k < V) { p = gv[k]; pts[k] = p; }
        px[j] = p.x; py[j] = p.y;
        ord[k] = 0; // a ring with NaN coordinates has no total order: every slot of the permutation still holds a vertex index
    }
    wave_lds_fence();
    // Most rings are short: when every ring of the wave fits into half of the slots, the per-slot loops below run over that half
    const bool half = !__any(V > LPR * (E / 2)); // wave-uniform
    double sh = 0.0, perim = 0.0;
    int rank[E];
#pragma unroll
    for (int j = 0; j < E; ++j) rank[j] = 0;
    auto slots = [&](auto ee_tag) {
        constexpr int EE = decltype(ee_tag)::value;
        // ---- shoelace + perimeter: the edge that leaves each own vertex ------------------------------------------------
#pragma unroll
        for (int j = 0; j < EE; ++j) {
            const int k = sub + LPR * j;
            if (k < V) {
                const float2 nb = pts[k + 1 == V ? 0 : k + 1];
                const double ax = (double)px[j], ay = (double)py[j], bx = (double)nb.x, by = (double)nb.y;
                sh += ax * by - bx * ay;
                const double dx = bx - ax, dy = by - ay;
                perim += sqrt(dx * dx + dy * dy);
            }
        }
        // ---- rank sort: lexicographic (x, then y), ties by ring index --------------------------------------------------
#pragma unroll 1
        for (int t = 0; __any(t < V); ++t) {
            const bool valid = t < V;
            const float2 q = pts[valid ? t : 0];
#pragma unroll
            for (int j = 0; j < EE; ++j) {
                const int k = sub + LPR * j;
                const bool before = (q.x < px[j]) || (q.x == px[j] && ((q.y < py[j]) || (q.y == py[j] && t < k)));
                rank[j] += (valid && before) ? 1 : 0;
            }
        }
#pragma unroll
        for (int j = 0; j < EE; ++j) {
            const int k = sub + LPR * j;
            if (k < V) ord[rank[j] < V ? rank[j] : 0] = (uint8_t)k;
        }
    };
    if (half) slots(std::integral_constant<int, E / 2>{});
    else slots(std::integral_constant<int, E>{});
    sh = group_total_f64<LPR>(sh);
    perim = group_total_f64<LPR>(perim);
    wave_lds_fence();
    // ---- Andrew's monotone chain (collinear points dropped: cross <= 0 pops) ---------------------------------------------
    // Two lanes per ring: lane 0 walks the lower chain over the sorted points first -> last, lane 1 the upper chain last -> first.
    // The single-stack form (polygon_one) never pops its upper chain below the lower chain's last point, so the two chains are
    // independent stacks: the hull is lower + upper without the upper chain's two end points, and its edges are the consecutive
    // pairs of the two chains.  Top two hull points in registers (a push costs no LDS read, a pop one); the next sorted point is
    // requested before the current one is worked on.
    if (sub < 2 && V > 0) {
        uint8_t *hull = s_hull[wv][g][sub];
        auto P = [&](int idx) { const float2 f = pts[idx < V ? idx : 0]; return pt{(double)f.x, (double)f.y}; };
        const int first = sub == 0 ? 0 : V - 1, step = sub == 0 ? 1 : -1;
        int m = 0;
        pt h2 = {0.0, 0.0}, h1 = {0.0, 0.0}; // hull[m-2], hull[m-1]
        int idx = ord[first];
        pt c = P(idx);
        for (int k = 0; k < V; ++k) {
            const int idx_next = ord[k + 1 < V ? first + step * (k + 1) : first];
            const pt c_next = P(idx_next);
            while (m >= 2 && cross3(h2, h1, c) <= 0.0) {
                --m;
                h1 = h2;
                if (m >= 2) h2 = P(hull[m - 2]);
            }
            hull[m++] = (uint8_t)idx;
            h2 = h1;
            h1 = c;
            idx = idx_next;
            c = c_next;
        }
        s_m[wv][g][sub] = m;
    } else if (sub < 2) {
        s_m[wv][g][sub] = 0;
    }
    wave_lds_fence();
    // ---- hull area and perimeter: per-edge terms of both chains over the group ----------------------------------------------
    double hsh = 0.0, hperim = 0.0;
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
        const int m = s_m[wv][g][ch];
        const uint8_t *hull = s_hull[wv][g][ch];
        for (int e = sub; e + 1 < m; e += LPR) {
            const int ia = hull[e], ib = hull[e + 1];
            const float2 fa = pts[ia < V ? ia : 0], fb = pts[ib < V ? ib : 0];
            const double ax = (double)fa.x, ay = (double)fa.y, bx = (double)fb.x, by = (double)fb.y;
            hsh += ax * by - bx * ay;
            const double dx = bx - ax, dy = by - ay;
            hperim += sqrt(dx * dx + dy * dy);
        }
    }
    hsh = group_total_f64<LPR>(hsh);
    hperim = group_total_f64<LPR>(hperim);
    if (sub != 0 || !live) return;
    if (big) { // the serial program on global memory, index scratch of the ring's own
        uint32_t *srt = scratch + 3 * (size_t)v0 + 4 * (size_t)i;
        polygon_one<uint32_t, const float2 *__restrict__>(gv, Vraw, srt, srt + Vraw, o, o64);
        return;
    }
    const double area = fabs(sh) * 0.5;
    const double eq_perim = 2.0 * sqrt(PI * area);
    const double compacity = 4.0 * PI * area / (perim * perim);
    const double harea = fabs(hsh) * 0.5;
    const double deviation = (harea - area) / harea;
    if (o) { // columns of the geometry block (shape.rs:116-127), rounded once
        o[0] = (float)area;
        o[5] = (float)perim;
        o[6] = (float)eq_perim;
        o[7] = (float)compacity;
        o[9] = (float)harea;
        o[10] = (float)deviation;
        o[11] = (float)hperim;
    }
    if (o64) { // nfx_polygon_features_f64: the same seven values at geometric_features' own width
        o64[0] = area; o64[1] = perim; o64[2] = eq_perim; o64[3] = compacity; o64[4] = harea; o64[5] = deviation; o64[6] = hperim;
    }
}

// ---- patch sizes 129..512: the same workgroup-per-nucleus shape with rows of W = ceil(ps / 64) <= 8 words ------------------------
// Thread t rasterises rows t and t + 256 (every row tests every edge, as the wide kernel), keeps their words in registers for the
// ellipse step, and adds its rows' integer moments -- 64-bit now: sum r^2 over a 512 x 512 mask is 2^34 -- to the workgroup's
// totals through LDS atomics.  Same expression trees as the kernels above: a nucleus gives the same columns in any family.
static constexpr int HUGE_W = NFX_MAX_ROW_WORDS, HUGE_ROWS = NFX_MAX_PATCH_SIZE / 256;
static_assert(NFX_MAX_PATCH_SIZE % 256 == 0 && NFX_MAX_PATCH_SIZE % 64 == 0, "rows per thread, words per row");

// integer moments of one row from its words: count, sum of columns, sum of squared columns
static __device__ __forceinline__ void row_moments_huge(const uint64_t (&w)[HUGE_W], long long &kr, long long &sc, long long &scc)
{
    kr = 0; sc = 0; scc = 0;
#pragma unroll
    for (int k = 0; k < HUGE_W; ++k) {
        int p, s1, q;
        word_moments(w[k], p, s1, q);
        const long long b = 64ll * k; // (b + j)^2 = b^2 + 2 b j + j^2
        kr += p; sc += s1 + b * p; scc += q + 2 * b * s1 + b * b * p;
    }
}

// rows in registers -> moments into s_tot[0..5], then (SHAPE) the columns that come from the mask; shared by the two kernels below
template <bool SHAPE>
static __device__ __forceinline__ void shape_from_rows_huge(const uint64_t (&row)[HUGE_ROWS][HUGE_W], int ps, int t, unsigned long long *s_tot,
                                                            float *__restrict__ o, int *__restrict__ area_out)
{
    long long a[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < HUGE_ROWS; ++u) {
        const long long r = t + 256 * u;
        long long kr, sc, scc;
        row_moments_huge(row[u], kr, sc, scc); // (rows at or beyond ps are all zero)
        a[0] += kr; a[1] += r * kr; a[2] += sc; a[3] += r * r * kr; a[4] += r * sc; a[5] += scc;
    }
    if (a[0]) {
#pragma unroll
        for (int k = 0; k < 6; ++k) atomicAdd(&s_tot[k], (unsigned long long)a[k]);
    }
    __syncthreads();
    const long long K = (long long)s_tot[0];
    if (area_out && t == 0) *area_out = (int)K;
    if (SHAPE) {
        const ShapeAxes A = shape_axes<long long>(K, (long long)s_tot[1], (long long)s_tot[2], (long long)s_tot[3], (long long)s_tot[4], (long long)s_tot[5], ps);
        const EllipseFrame E = ellipse_frame(A, ps);
        int x = 0;
#pragma unroll
        for (int u = 0; u < HUGE_ROWS; ++u) {
            const int r = t + 256 * u;
            if (r < ps) {
                int lo, len;
                ellipse_row_span(E, ps, r, lo, len);
#pragma unroll
                for (int k = 0; k < HUGE_W; ++k) {
                    const uint64_t e = low_bits(lo + len - 64 * k) & ~low_bits(lo - 64 * k);
                    x += __popcll(e ^ row[u][k]);
                }
            }
        }
        if (x) atomicAdd(&s_tot[6], (unsigned long long)x);
        __syncthreads();
        if (t == 0) {
            o[1] = A.major; o[2] = A.minor; o[3] = A.ecc; o[4] = A.angle;
            o[8] = (float)(long long)s_tot[6] / (float)K;
        }
    }
}

template <bool SHAPE>
__global__ __launch_bounds__(256) void k_raster_shape_huge(int64_t n, const uint32_t *__restrict__ poly_off, const float *__restrict__ poly_xy,
                                                           int ps, uint64_t *__restrict__ rows_out, float *__restrict__ out, int F, int col0,
                                                           int *__restrict__ areas_out)
{
    __shared__ unsigned long long s_tot[8];
    const int t = threadIdx.x;
    const int W = (ps + 63) >> 6;
    for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
        if (t < 8) s_tot[t] = 0ull;
        __syncthreads();
        const uint32_t v0 = poly_off[i];
        const int V = (int)(poly_off[i + 1] - v0);
        const double half = (double)ps / 2.0;
        uint64_t row[HUGE_ROWS][HUGE_W];
#pragma unroll
        for (int u = 0; u < HUGE_ROWS; ++u) {
#pragma unroll
            for (int k = 0; k < HUGE_W; ++k) row[u][k] = 0ull;
            const int r = t + 256 * u;
            if (r < ps) {
                const double py = (double)r - half;
                for (int e = 0; e < V; ++e) {
                    const float2 p1 = reinterpret_cast<const float2 *>(poly_xy)[v0 + e];
                    const float2 p2 = reinterpret_cast<const float2 *>(poly_xy)[v0 + (e + 1 == V ? 0 : e + 1)];
                    const double x1 = (double)p1.x, y1 = (double)p1.y, x2 = (double)p2.x, y2 = (double)p2.y;
                    if ((y1 > py) != (y2 > py)) {
                        const int k = crossing_count(x1, y1, x2, y2, py, half, ps); // <= ps: words at or beyond W stay zero
#pragma unroll
                        for (int w = 0; w < HUGE_W; ++w) row[u][w] ^= low_bits(k - 64 * w);
                    }
                }
                if (rows_out) {
#pragma unroll
                    for (int w = 0; w < HUGE_W; ++w)
                        if (w < W) rows_out[((size_t)i * ps + r) * W + w] = row[u][w];
                }
            }
        }
        shape_from_rows_huge<SHAPE>(row, ps, t, s_tot, SHAPE ? out + i * (int64_t)F + col0 : nullptr, areas_out ? areas_out + i : nullptr);
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_shape_from_rows_huge(int64_t n, const uint64_t *__restrict__ rows, int ps, float *__restrict__ out, int F, int col0)
{
    __shared__ unsigned long long s_tot[8];
    const int t = threadIdx.x;
    const int W = (ps + 63) >> 6;
    for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
        if (t < 8) s_tot[t] = 0ull;
        __syncthreads();
        uint64_t row[HUGE_ROWS][HUGE_W];
#pragma unroll
        for (int u = 0; u < HUGE_ROWS; ++u) {
            const int r = t + 256 * u;
#pragma unroll
            for (int w = 0; w < HUGE_W; ++w) row[u][w] = (r < ps && w < W) ? rows[((size_t)i * ps + r) * W + w] : 0ull;
        }
        shape_from_rows_huge<true>(row, ps, t, s_tot, out + i * (int64_t)F + col0, nullptr);
        __syncthreads();
    }
}

// ---- launchers ------------------------------------------------------------------------------------
hipError_t launch_raster_shape(nfx_ctx *ctx, int64_t n, const uint32_t *poly_off, const float *poly_xy, int ps,
                               uint64_t *rows_out, float *out, int F, int col0, int *areas_out)
{
    if (n == 0) return hipSuccess;
    ScopedTiming t(ctx, KS_RASTER_SHAPE);
    if (ps > NFX_WIDE_PS) {
        const unsigned wgrid = (unsigned)(n < 256 * 8 ? n : 256 * 8);
        if (out) hipLaunchKernelGGL(k_raster_shape_huge<true>, dim3(wgrid), dim3(256), 0, ctx->stream, n, poly_off, poly_xy, ps, rows_out, out, F, col0, areas_out);
        else hipLaunchKernelGGL(k_raster_shape_huge<false>, dim3(wgrid), dim3(256), 0, ctx->stream, n, poly_off, poly_xy, ps, rows_out, out, F, col0, areas_out);
        return hipGetLastError();
    }
    if (ps > 64) {
        const unsigned wgrid = (unsigned)(n < 256 * 16 ? n : 256 * 16);
        if (out) hipLaunchKernelGGL(k_raster_shape_wide<true>, dim3(wgrid), dim3(256), 0, ctx->stream, n, poly_off, poly_xy, ps, rows_out, out, F, col0, areas_out);
        else hipLaunchKernelGGL(k_raster_shape_wide<false>, dim3(wgrid), dim3(256), 0, ctx->stream, n, poly_off, poly_xy, ps, rows_out, out, F, col0, areas_out);
        return hipGetLastError();
    }
    const unsigned grid = (unsigned)((n + 3) / 4);
    if (out)
        hipLaunchKernelGGL(k_raster_shape<true>, dim3(grid), dim3(256), 0, ctx->stream, n, poly_off, poly_xy, ps, rows_out, out, F, col0, areas_out);
    else
        hipLaunchKernelGGL(k_raster_shape<false>, dim3(grid), dim3(256), 0, ctx->stream, n, poly_off, poly_xy, ps, rows_out, out, F, col0, areas_out);
    return hipGetLastError();
}

hipError_t launch_shape_from_rows(nfx_ctx *ctx, int64_t n, const uint64_t *rows, int ps, float *out, int F, int col0)
{
    if (n == 0) return hipSuccess;
    ScopedTiming t(ctx, KS_RASTER_SHAPE);
    if (ps > NFX_WIDE_PS) {
        hipLaunchKernelGGL(k_shape_from_rows_huge, dim3((unsigned)(n < 256 * 8 ? n : 256 * 8)), dim3(256), 0, ctx->stream, n, rows, ps, out, F, col0);
        return hipGetLastError();
    }
    if (ps > 64) {
        hipLaunchKernelGGL(k_shape_from_rows_wide, dim3((unsigned)(n < 256 * 16 ? n : 256 * 16)), dim3(256), 0, ctx->stream, n, rows, ps, out, F, col0);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_shape_from_rows, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream, n, rows, ps, out, F, col0);
    return hipGetLastError();
}

hipError_t launch_polygon(nfx_ctx *ctx, int64_t n, const uint32_t *poly_off, const float *poly_xy, void *scratch,
                          float *out, int F, int col0, double *out64, bool beside_other_work)
{
    if (n == 0) return hipSuccess;
    ScopedTiming t(ctx, KS_POLYGON);
    if (beside_other_work) {
        hipLaunchKernelGGL(k_polygon_serial, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, n, poly_off, poly_xy,
                           reinterpret_cast<uint32_t *>(scratch), out, F, col0, out64);
        return hipGetLastError();
    }
    constexpr int rings_per_block = 4 * (64 / NFX_PG_LPR);
    hipLaunchKernelGGL(k_polygon<NFX_PG_LPR>, dim3((unsigned)((n + rings_per_block - 1) / rings_per_block)), dim3(256), 0, ctx->stream, n,
                       poly_off, poly_xy, reinterpret_cast<uint32_t *>(scratch), out, F, col0, out64);
    return hipGetLastError();
}

}  // namespace nfx
