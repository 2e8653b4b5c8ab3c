/*
 * nfx_oracle.c -- CPU restatement of the reference's per-nucleus feature path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (libnfx.so) never links,
 * loads or calls anything in this directory.
 *
 * Reference: oxabz/nuclei-feature-extraction (Rust, /root/reference).  Citations below are
 * file:line relative to /root/reference/.
 *
 * PARITY STATUS
 *   - In-tree arithmetic (src/features/shape.rs, color.rs, texture.rs orchestration, masked
 *     statistics, utils.rs:54-74 centroid) is restated op for op and is pinned by
 *     tests/test_torch_twin.py, which issues the same ATen op sequence the reference issues
 *     through `tch` on torch-CPU (same ATen CPU kernels, MKL LAPACK geev).
 *   - The reference has NO tests, golden vectors or fixtures, and the arithmetic of
 *     tch_utils::{shapes::polygon, shapes::ellipse, color::hsv_from_rgb, color::hed_from_rgb,
 *     glcm, glrlm, gabor} (tch-utils 0.1.1 @ d1c10c0, Cargo.lock:2591-2598) and
 *     geometric_features::* (0.1.0 @ 163ae81, Cargo.lock:932-935) is in un-vendored git
 *     dependencies that are absent here.  For those functions this file restates the published
 *     textbook algorithm named at each function, anchored on the reference's call sites.
 *     ==> PARITY UNPINNED for those functions (marked [DEP-UNPINNED] below).
 *
 * Build: gcc -O2 -fPIC -shared -ffp-contract=off -fno-fast-math -fopenmp (oracle/Makefile).
 * -ffp-contract=off matters: integer-valued outputs (mask bits, counts) are compared
 * bit-exactly with the HIP path, so no fused multiply-adds may be formed here.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_PI 3.14159265358979323846

/* ------------------------------------------------------------------------------------------
 * Column counts (shape.rs:113-128, color.rs:80-99, texture.rs:19-20,81-157,174-176,244-301,
 * 319-320,346-361)
 * ---------------------------------------------------------------------------------------- */
enum { ORC_F_GEOMETRY = 12, ORC_F_COLOR = 18, ORC_F_GLCM = 224, ORC_F_GLRLM = 68, ORC_F_GABOR = 96 };
static const int ORC_NCOLS[5] = { ORC_F_GEOMETRY, ORC_F_COLOR, ORC_F_GLCM, ORC_F_GLRLM, ORC_F_GABOR };

int orc_num_columns(int set) { return (set >= 0 && set < 5) ? ORC_NCOLS[set] : -1; }

/* ------------------------------------------------------------------------------------------
 * R1  preprocess_polygon  (utils.rs:54-74)
 *   centroid = sequential f32 sum of ALL listed ring vertices / V ; centred = v - centroid (f32)
 * ---------------------------------------------------------------------------------------- */
void orc_preprocess_polygon(const float *ring_xy, int V, float centroid[2], float *centred_xy)
{
    float ax = 0.0f, ay = 0.0f;
    for (int i = 0; i < V; ++i) {
        ax += ring_xy[2 * i + 0];
        ay += ring_xy[2 * i + 1];
    }
    float cx = ax / (float)V; /* V == 0 -> 0/0 = NaN, as the reference would produce */
    float cy = ay / (float)V;
    centroid[0] = cx;
    centroid[1] = cy;
    for (int i = 0; i < V; ++i) {
        centred_xy[2 * i + 0] = ring_xy[2 * i + 0] - cx;
        centred_xy[2 * i + 1] = ring_xy[2 * i + 1] - cy;
    }
}

/* ------------------------------------------------------------------------------------------
 * R2  tch_utils::shapes::polygon(ps, ps, &pts_f64, (Float,Cpu))   [DEP-UNPINNED]
 *   call sites utils.rs:90-95,152-157.  Restated as the even-odd crossing-number fill
 *   (Shimrat 1962 / Franklin PNPOLY) evaluated at integer pixel coordinates with the origin at
 *   the patch centre (the vertices are centroid-centred, utils.rs:66-73):
 *     pixel (r,c) is at (px,py) = (c - ps/2, r - ps/2); all arithmetic f64 (the call passes
 *     (f64,f64) points, utils.rs:39-42).
 *     edge i: (x1,y1)=v[i], (x2,y2)=v[(i+1)%V]; it toggles the pixel iff
 *         (y1 > py) != (y2 > py)  &&  px < (x2-x1)*(py-y1)/(y2-y1) + x1
 *   Output [ps*ps] f32 in {0,1}, row-major (index 0 = row = y).
 * ---------------------------------------------------------------------------------------- */
void orc_raster_polygon(const float *centred_xy, int V, int ps, float *mask)
{
    const double half = (double)ps / 2.0;
    for (int r = 0; r < ps; ++r) {
        const double py = (double)r - half;
        for (int c = 0; c < ps; ++c) {
            const double px = (double)c - half;
            int inside = 0;
            for (int i = 0; i < V; ++i) {
                int j = (i + 1 == V) ? 0 : i + 1;
                double x1 = (double)centred_xy[2 * i], y1 = (double)centred_xy[2 * i + 1];
                double x2 = (double)centred_xy[2 * j], y2 = (double)centred_xy[2 * j + 1];
                if ((y1 > py) != (y2 > py)) {
                    double xint = (x2 - x1) * (py - y1) / (y2 - y1) + x1;
                    if (px < xint) inside ^= 1;
                }
            }
            mask[r * ps + c] = inside ? 1.0f : 0.0f;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * S3 helper: LAPACK slanv2 / slapy2 in f32 -- what sgeev (torch.linalg.eig on CPU, shape.rs:186)
 * reaches for a 2x2 block through shseqr->slahqr.  Validated against torch-CPU in
 * tests/test_torch_twin.py.
 * ---------------------------------------------------------------------------------------- */
static float orc_sign1(float x) { return signbit(x) ? -1.0f : 1.0f; } /* Fortran SIGN(1,x) */

static float orc_slapy2(float x, float y)
{
    float xa = fabsf(x), ya = fabsf(y);
    float w = xa > ya ? xa : ya;
    float z = xa > ya ? ya : xa;
    if (z == 0.0f) return w;
    float t = z / w;
    return w * sqrtf(1.0f + t * t);
}

/* in: a b c d (row-major 2x2).  out: rt1r rt2r cs sn.  Only real-eigenvalue paths can occur for a
 * symmetric input, the complex path is kept for structural fidelity. */
static void orc_slanv2(float a, float b, float c, float d, float *rt1r, float *rt2r, float *cs_, float *sn_)
{
    const float eps = 1.1920929e-07f; /* slamch('P') = 2^-23 */
    const float multpl = 4.0f;
    float cs, sn;
    if (c == 0.0f) {
        cs = 1.0f; sn = 0.0f;
    } else if (b == 0.0f) {
        cs = 0.0f; sn = 1.0f;
        float t = d; d = a; a = t; b = -c; c = 0.0f;
    } else if ((a - d) == 0.0f && orc_sign1(b) != orc_sign1(c)) {
        cs = 1.0f; sn = 0.0f;
    } else {
        float temp = a - d;
        float p = 0.5f * temp;
        float bcmax = fmaxf(fabsf(b), fabsf(c));
        float bcmis = fminf(fabsf(b), fabsf(c)) * orc_sign1(b) * orc_sign1(c);
        float scale = fmaxf(fabsf(p), bcmax);
        float z = p / scale * p + bcmax / scale * bcmis;
        if (z >= multpl * eps) {
            float sq = sqrtf(scale) * sqrtf(z);
            z = p + (signbit(p) ? -fabsf(sq) : fabsf(sq));
            a = d + z;
            d = d - (bcmax / z) * bcmis;
            float tau = orc_slapy2(c, z);
            cs = z / tau;
            sn = c / tau;
            b = b - c;
            c = 0.0f;
        } else {
            float sigma = b + c;
            float tau = orc_slapy2(sigma, temp);
            cs = sqrtf(0.5f * (1.0f + fabsf(sigma) / tau));
            sn = -(p / (tau * cs)) * orc_sign1(sigma);
            float aa = a * cs + b * sn, bb = -a * sn + b * cs;
            float cc = c * cs + d * sn, dd = -c * sn + d * cs;
            a = aa * cs + cc * sn; b = bb * cs + dd * sn;
            c = -aa * sn + cc * cs; d = -bb * sn + dd * cs;
            temp = 0.5f * (a + d);
            a = temp; d = temp;
            if (c != 0.0f) {
                if (b != 0.0f) {
                    if (orc_sign1(b) == orc_sign1(c)) {
                        float sab = sqrtf(fabsf(b)), sac = sqrtf(fabsf(c));
                        float pp = signbit(c) ? -fabsf(sab * sac) : fabsf(sab * sac);
                        tau = 1.0f / sqrtf(fabsf(b + c));
                        a = temp + pp; d = temp - pp;
                        b = b - c; c = 0.0f;
                        float cs1 = sab * tau, sn1 = sac * tau;
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
    *rt1r = a; *rt2r = d; *cs_ = cs; *sn_ = sn;
}

/* ------------------------------------------------------------------------------------------
 * S7  geometric_features::{area,perimeter,equivalent_perimeter,compacity} and
 *     convex_hull::convex_hull_features  (call site shape.rs:89-97)        [DEP-UNPINNED]
 *   f64 on the centred vertices (shape.rs:68, utils.rs:39-42).  Restated as:
 *     area = |shoelace|/2 ; perimeter = closed-ring edge-length sum ;
 *     equivalent_perimeter = 2*sqrt(pi*area) ; compacity = 4*pi*area/perimeter^2 ;
 *     hull = Andrew monotone chain (collinear points dropped) ; hull area/perimeter as above ;
 *     deviation ("convex_deffect") = (hull_area - area)/hull_area.
 * ---------------------------------------------------------------------------------------- */
typedef struct { double x, y; } orc_pt;

static int orc_pt_less(const orc_pt *a, const orc_pt *b)
{
    return (a->x < b->x) || (a->x == b->x && a->y < b->y);
}

static double orc_cross(const orc_pt *o, const orc_pt *a, const orc_pt *b)
{
    return (a->x - o->x) * (b->y - o->y) - (a->y - o->y) * (b->x - o->x);
}

static double orc_ring_area(const orc_pt *p, int n)
{
    double s = 0.0;
    for (int i = 0; i < n; ++i) {
        int j = (i + 1 == n) ? 0 : i + 1;
        s += p[i].x * p[j].y - p[j].x * p[i].y;
    }
    return fabs(s) * 0.5;
}

static double orc_ring_perimeter(const orc_pt *p, int n)
{
    double s = 0.0;
    for (int i = 0; i < n; ++i) {
        int j = (i + 1 == n) ? 0 : i + 1;
        double dx = p[j].x - p[i].x, dy = p[j].y - p[i].y;
        s += sqrt(dx * dx + dy * dy);
    }
    return s;
}

static void orc_polygon_features(const float *centred_xy, int V, double out7[7])
{
    orc_pt *p = (orc_pt *)malloc(sizeof(orc_pt) * (size_t)(V > 0 ? V : 1));
    orc_pt *s = (orc_pt *)malloc(sizeof(orc_pt) * (size_t)(V > 0 ? V : 1));
    orc_pt *h = (orc_pt *)malloc(sizeof(orc_pt) * (size_t)(2 * V + 2));
    for (int i = 0; i < V; ++i) {
        p[i].x = (double)centred_xy[2 * i];
        p[i].y = (double)centred_xy[2 * i + 1];
        s[i] = p[i];
    }
    double area = orc_ring_area(p, V);
    double perim = orc_ring_perimeter(p, V);
    double eq_perim = 2.0 * sqrt(ORC_PI * area);
    double compacity = 4.0 * ORC_PI * area / (perim * perim);

    /* insertion sort: deterministic, V is small */
    for (int i = 1; i < V; ++i) {
        orc_pt k = s[i];
        int j = i - 1;
        while (j >= 0 && orc_pt_less(&k, &s[j])) { s[j + 1] = s[j]; --j; }
        s[j + 1] = k;
    }
    int m = 0;
    for (int i = 0; i < V; ++i) { /* lower hull */
        while (m >= 2 && orc_cross(&h[m - 2], &h[m - 1], &s[i]) <= 0.0) --m;
        h[m++] = s[i];
    }
    int lower = m + 1;
    for (int i = V - 2; i >= 0; --i) { /* upper hull */
        while (m >= lower && orc_cross(&h[m - 2], &h[m - 1], &s[i]) <= 0.0) --m;
        h[m++] = s[i];
    }
    if (m > 1) --m; /* last point == first point */
    double harea = orc_ring_area(h, m);
    double hperim = orc_ring_perimeter(h, m);
    double deviation = (harea - area) / harea;

    out7[0] = area; out7[1] = perim; out7[2] = eq_perim; out7[3] = compacity;
    out7[4] = harea; out7[5] = deviation; out7[6] = hperim;
    free(p); free(s); free(h);
}

/* ------------------------------------------------------------------------------------------
 * S5  tch_utils::shapes::ellipse(ps, ps, (x_c,y_c), (a,b), angle, (Float,dev))  [DEP-UNPINNED]
 *   call site shape.rs:80-87.  Restated: pixel (r,c) is inside iff, in f64,
 *     x = (c - ps/2) - x_c ; y = (r - ps/2) - y_c
 *     xr = x*cos(t) + y*sin(t) ; yr = -x*sin(t) + y*cos(t)
 *     (xr/a)*(xr/a) + (yr/b)*(yr/b) <= 1
 * ---------------------------------------------------------------------------------------- */
static int orc_ellipse_test(int r, int c, double half, double xc, double yc, double a, double b, double ct, double st)
{
    double x = ((double)c - half) - xc;
    double y = ((double)r - half) - yc;
    double xr = x * ct + y * st;
    double yr = y * ct - x * st;
    double u = xr / a, v = yr / b;
    return (u * u + v * v) <= 1.0;
}

void orc_raster_ellipse(int ps, float xc, float yc, float a, float b, float angle, float *out)
{
    double ct = cos((double)angle), st = sin((double)angle);
    for (int r = 0; r < ps; ++r)
        for (int c = 0; c < ps; ++c)
            out[r * ps + c] = orc_ellipse_test(r, c, (double)ps / 2.0, xc, yc, a, b, ct, st) ? 1.0f : 0.0f;
}

/* ------------------------------------------------------------------------------------------
 * S2/S3/S4/S6  (shape.rs:137-226), per nucleus.  mask: [ps*ps] f32, nonzero == inside.
 * aux (optional, 8 doubles): K, sum_r, sum_c, xor_count, p, q, r, -  for exact-integer checks.
 * ---------------------------------------------------------------------------------------- */
/* the seven polygon columns at geometric_features' own width (shape.rs:89-97): area, perimeter, equivalent_perimeter,
   compacity, convex_hull_area, convex_deffect, convex_perimeter */
void orc_polygon_features_f64(const float *centred_xy, int V, double out7[7]) { orc_polygon_features(centred_xy, V, out7); }

void orc_shape_one(const float *centred_xy, int V, const float *mask, int ps, float out[12], double *aux)
{
    /* center_of_mass, shape.rs:219-226: nonzero() -> mean(dim0, Float).  Sums of integer
       coordinates < 2^24 are exact in f32 in any order, so mean = fl32(sum)/fl32(K). */
    int64_t K = 0, sr = 0, sc = 0, srr = 0, src = 0, scc = 0;
    for (int r = 0; r < ps; ++r)
        for (int c = 0; c < ps; ++c)
            if (mask[r * ps + c] != 0.0f) {
                K += 1; sr += r; sc += c; srr += (int64_t)r * r; src += (int64_t)r * c; scc += (int64_t)c * c;
            }
    float mu_r = (float)sr / (float)K; /* K == 0 -> NaN */
    float mu_c = (float)sc / (float)K;
    float halfps = (float)ps / 2.0f;              /* shape.rs:72-73 */
    float cen0 = mu_r - halfps, cen1 = mu_c - halfps;

    /* major_minor_axes_w_angle, shape.rs:141-203.  cov = P^T P / K with P = nz - mean(f32).
       The reference forms it with an f32 mm whose summation order is BLAS-internal; restated as
       the exactly-rounded value: central moments about the f32 means in f64, rounded to f32,
       then the f32 division by K the reference performs. */
    float major, minor, angle;
    float p = NAN, q = NAN, rr = NAN;
    if (K == 0) {
        major = minor = angle = NAN; /* mean of empty is NaN -> isnan branch, shape.rs:159-166 */
    } else {
        double mr = (double)mu_r, mc = (double)mu_c, kd = (double)K;
        double Crr = (double)srr - 2.0 * mr * (double)sr + kd * mr * mr;
        double Crc = (double)src - mr * (double)sc - mc * (double)sr + kd * mr * mc;
        double Ccc = (double)scc - 2.0 * mc * (double)sc + kd * mc * mc;
        p = (float)Crr / (float)K;
        q = (float)Crc / (float)K;
        rr = (float)Ccc / (float)K;
        float w0, w1, cs, sn;
        if (q == 0.0f) { /* sgebal isolates both eigenvalues: w = diag, V = I */
            w0 = p; w1 = rr; cs = 1.0f; sn = 0.0f;
        } else {
            orc_slanv2(p, q, q, rr, &w0, &w1, &cs, &sn);
        }
        /* V = [[cs,-sn],[sn,cs]]; shape.rs:191-195 takes a ROW of V */
        float mx, my, ea, eb;
        if (w0 > w1) { ea = sqrtf(w0); eb = sqrtf(w1); mx = cs; my = -sn; }
        else         { ea = sqrtf(w1); eb = sqrtf(w0); mx = sn; my = cs; }
        /* atan2 evaluated in f64 and rounded: removes libm-vs-device f32 atan2 differences */
        angle = (float)atan2((double)mx, (double)my); /* shape.rs:200: x.atan2(y) */
        major = ea * 2.0f;
        minor = eb * 2.0f;
    }
    /* eccentricity, shape.rs:205-207 */
    float hM = major * 0.5f, hm = minor * 0.5f;
    float ecc = sqrtf(hM * hM - hm * hm) / hM;

    /* ellipse + eliptic_deviation, shape.rs:80-87,209-217: centre (x_c,y_c)=(centroid[1],centroid[0]) */
    int64_t xorcnt = 0;
    {
        double ct = cos((double)angle), st = sin((double)angle);
        double half = (double)ps / 2.0;
        for (int r = 0; r < ps; ++r)
            for (int c = 0; c < ps; ++c) {
                int e = orc_ellipse_test(r, c, half, (double)cen1, (double)cen0, (double)major, (double)minor, ct, st);
                int m = mask[r * ps + c] != 0.0f;
                xorcnt += (e != m);
            }
    }
    float edev = (float)xorcnt / (float)K;

    double pf[7];
    orc_polygon_features(centred_xy, V, pf);

    /* column order shape.rs:113-128 */
    out[0] = (float)pf[0];   /* area */
    out[1] = major;          /* major_axis */
    out[2] = minor;          /* minor_axis */
    out[3] = ecc;            /* eccentricity */
    out[4] = angle;          /* orientation */
    out[5] = (float)pf[1];   /* perimeter */
    out[6] = (float)pf[2];   /* equivalent_perimeter */
    out[7] = (float)pf[3];   /* compacity */
    out[8] = edev;           /* eliptic_deviation */
    out[9] = (float)pf[4];   /* convex_hull_area */
    out[10] = (float)pf[5];  /* convex_deffect */
    out[11] = (float)pf[6];  /* convex_perimeter */
    if (aux) {
        aux[0] = (double)K; aux[1] = (double)sr; aux[2] = (double)sc; aux[3] = (double)xorcnt;
        aux[4] = p; aux[5] = q; aux[6] = rr; aux[7] = 0.0;
    }
}

/* ------------------------------------------------------------------------------------------
 * C4  tch_utils::color::hsv_from_rgb / hed_from_rgb (call site color.rs:45-46) [DEP-UNPINNED]
 *   HSV: Smith 1978 hexcone, H in degrees [0,360] (color.rs:147,154 treat hue as degrees),
 *        S = delta/max (0 when max==0), V = max; f32 arithmetic.
 *   HED: Ruifrok & Johnston 2001 colour deconvolution as in scikit-image rgb2hed:
 *        od = log(max(rgb,1e-6))/log(1e-6) ; stains = od @ inv(rgb_from_hed) ; max(stains,0).
 *        The log is evaluated in f64 and rounded to f32; the 3x3 product is f32, c=0,1,2 order.
 * ---------------------------------------------------------------------------------------- */
static void orc_hsv(float r, float g, float b, float *h, float *s, float *v)
{
    float mx = fmaxf(r, fmaxf(g, b));
    float mn = fminf(r, fminf(g, b));
    float d = mx - mn;
    float hh;
    if (d == 0.0f) hh = 0.0f;
    else if (mx == r) { hh = fmodf((g - b) / d, 6.0f); if (hh < 0.0f) hh += 6.0f; }
    else if (mx == g) hh = (b - r) / d + 2.0f;
    else hh = (r - g) / d + 4.0f;
    *h = hh * 60.0f;
    *s = (mx == 0.0f) ? 0.0f : d / mx;
    *v = mx;
}

static const float ORC_HED_FROM_RGB[3][3] = {
    { 1.87798273563385f, -1.0076786279678345f, -0.5561158061027527f },
    { -0.06590805947780609f, 1.134730339050293f, -0.135521799325943f },
    { -0.6019073724746704f, -0.48041418194770813f, 1.5735880136489868f },
};

static float orc_od(float x)
{
    float xc = x > 1e-6f ? x : 1e-6f;
    return (float)(log((double)xc) / log(1e-6));
}

static void orc_hed(float r, float g, float b, float out[3])
{
    float o0 = orc_od(r), o1 = orc_od(g), o2 = orc_od(b);
    for (int k = 0; k < 3; ++k) {
        float s = o0 * ORC_HED_FROM_RGB[0][k];
        s = s + o1 * ORC_HED_FROM_RGB[1][k];
        s = s + o2 * ORC_HED_FROM_RGB[2][k];
        out[k] = s > 0.0f ? s : 0.0f;
    }
}

/* C2 mean_std (color.rs:117-134) for one nucleus / one channel plane.
   mean = sum(x*m)/K ; std = sqrt(sum((x-mean)^2*m)/K).  Sums accumulate in f64 over f32 terms and
   are rounded to f32 before the f32 division the reference performs. */
static void orc_mean_std_plane(const float *x, const float *m, int n, float area, float *mean, float *std)
{
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += (double)(x[i] * m[i]);
    float mu = (float)s / area;
    double s2 = 0.0;
    for (int i = 0; i < n; ++i) {
        float dv = x[i] - mu;
        float sq = dv * dv;
        s2 += (double)(sq * m[i]);
    }
    *mean = mu;
    *std = sqrtf((float)s2 / area);
}

/* ------------------------------------------------------------------------------------------
 * C1  ColorFeatureSet::compute_features_batched (color.rs:11-102) for ONE reference batch.
 *   patches [n,3,ps,ps] f32, masks [n,ps,ps] f32 {0,1}.  out [n,18], column order color.rs:80-99.
 *   C3 circular_mean (color.rs:144-155) is batch-coupled exactly as written: hue [n,ps,ps] times
 *   mask [n,1,ps,ps] broadcasts to [n,n,ps,ps]; nucleus i sums mask_i * sin(h_j) over every patch j
 *   of the batch.
 * ---------------------------------------------------------------------------------------- */
/* single-pixel entry points of the two colour transforms, for the independent cross-checks in tests/ */
void orc_hsv_pixel(float r, float g, float b, float out[3]) { orc_hsv(r, g, b, &out[0], &out[1], &out[2]); }
void orc_hed_pixel(float r, float g, float b, float out[3]) { orc_hed(r, g, b, out); }

void orc_color_batch(int n, const float *patches, const float *masks, int ps, float *out)
{
    const int P = ps * ps;
    float *hsv = (float *)malloc(sizeof(float) * (size_t)n * 3 * P);
    float *hed = (float *)malloc(sizeof(float) * (size_t)n * 3 * P);
    double *S = (double *)calloc((size_t)P, sizeof(double));
    double *C = (double *)calloc((size_t)P, sizeof(double));
    const float deg2rad = (float)(ORC_PI / 180.0);
    const float rad2deg = (float)(180.0 / ORC_PI);
    for (int j = 0; j < n; ++j) {
        const float *pr = patches + (size_t)j * 3 * P, *pg = pr + P, *pb = pg + P;
        float *hj = hsv + (size_t)j * 3 * P;
        float *ej = hed + (size_t)j * 3 * P;
        for (int i = 0; i < P; ++i) {
            float h, s, v, e[3];
            orc_hsv(pr[i], pg[i], pb[i], &h, &s, &v);
            hj[i] = h; hj[P + i] = s; hj[2 * P + i] = v;
            orc_hed(pr[i], pg[i], pb[i], e);
            ej[i] = e[0]; ej[P + i] = e[1]; ej[2 * P + i] = e[2];
            float hr = h * deg2rad;          /* color.rs:147 */
            S[i] += (double)sinf(hr);        /* color.rs:149 */
            C[i] += (double)cosf(hr);        /* color.rs:148 */
        }
    }
    float *htmp = (float *)malloc(sizeof(float) * (size_t)P);
    for (int i = 0; i < n; ++i) {
        const float *m = masks + (size_t)i * P;
        const float *pp = patches + (size_t)i * 3 * P;
        float *hi = hsv + (size_t)i * 3 * P;
        float *ei = hed + (size_t)i * 3 * P;
        float *o = out + (size_t)i * ORC_F_COLOR;
        double ad = 0.0;
        for (int k = 0; k < P; ++k) ad += (double)m[k];
        float area = (float)ad;
        /* circular mean */
        double ss = 0.0, cc = 0.0;
        for (int k = 0; k < P; ++k) { ss += (double)m[k] * S[k]; cc += (double)m[k] * C[k]; }
        float sinm = (float)ss / area, cosm = (float)cc / area;
        float mh = (float)atan2((double)sinm, (double)cosm);
        mh = mh * rad2deg;
        mh = mh + 360.0f;
        mh = fmodf(mh, 360.0f);
        /* h -= mean_h (color.rs:52) then mean_std(hsv) (color.rs:53) */
        for (int k = 0; k < P; ++k) htmp[k] = hi[k] - mh;
        float mean, std;
        for (int ch = 0; ch < 3; ++ch) { /* rgb: cols 0..2 mean, 3..5 std */
            orc_mean_std_plane(pp + ch * P, m, P, area, &mean, &std);
            o[ch] = mean; o[3 + ch] = std;
        }
        o[6] = mh;                                               /* mean_h */
        orc_mean_std_plane(htmp, m, P, area, &mean, &std); o[9] = std;                  /* std_h */
        orc_mean_std_plane(hi + P, m, P, area, &mean, &std); o[7] = mean; o[10] = std;  /* s */
        orc_mean_std_plane(hi + 2 * P, m, P, area, &mean, &std); o[8] = mean; o[11] = std; /* v */
        for (int ch = 0; ch < 3; ++ch) {
            orc_mean_std_plane(ei + ch * P, m, P, area, &mean, &std);
            o[12 + ch] = mean; o[15 + ch] = std;
        }
    }
    free(htmp); free(hsv); free(hed); free(S); free(C);
}

/* ------------------------------------------------------------------------------------------
 * gray = patchs.mean_dim(-3, keepdim, Float)  (texture.rs:36,189,332): f32 (r+g)+b then /3.
 * Quantisation to L levels [DEP-UNPINNED]: q = min(floor(gray*L), L-1), uniform bins on [0,1].
 * ---------------------------------------------------------------------------------------- */
static float orc_gray(float r, float g, float b) { float s = r + g; s = s + b; return s / 3.0f; }
static int orc_quant(float gray, int L)
{
    float f = floorf(gray * (float)L);
    int q = (int)f;
    if (q < 0) q = 0;
    if (q > L - 1) q = L - 1;
    return q;
}

/* ------------------------------------------------------------------------------------------
 * G2  tch_utils::glcm::glcm(gray,(dy,dx),L,Some(mask),true) + glcm_features  [DEP-UNPINNED]
 *   call site texture.rs:38-63.  Restated (Haralick, Shanmugam, Dinstein 1973; 0-based levels;
 *   natural log; 0*log(0)=0):
 *     count ordered pairs (p, p+(dy,dx)) with both pixels inside the patch and inside the mask,
 *     symmetric => G = C + C^T, P = G / sum(G).  With px = row marginal, mu = sum i px(i),
 *     var = sum (i-mu)^2 px(i), pxy(k) = sum_{i+j=k} P, pd(k) = sum_{|i-j|=k} P:
 *       correlation = (sum ij P - mu^2)/var      contrast = sum k^2 pd(k)
 *       dissimilarity = sum k pd(k)               entropy = -sum P ln P
 *       angular_second_moment = sum P^2           sum_average = sum k pxy(k)
 *       sum_variance = sum (k-sum_average)^2 pxy  sum_entropy = -sum pxy ln pxy
 *       sum_of_squares = var                      inverse_difference_moment = sum P/(1+(i-j)^2)
 *       difference_average = sum k pd(k)          difference_variance = sum (k-DA)^2 pd(k)
 *       imc1 = (HXY-HXY1)/max(HX,HY)              imc2 = sqrt(max(0,1-exp(-2(HXY2-HXY))))
 *     HXY1 = -sum P ln(px(i)px(j)), HXY2 = -sum px(i)px(j) ln(px(i)px(j)).
 *   No pairs => 0/0 => every feature NaN.
 * ---------------------------------------------------------------------------------------- */
static const int ORC_GLCM_LEVELS[4] = { 32, 64, 128, 254 };                      /* texture.rs:19 */
static const int ORC_GLCM_OFFSETS[4][2] = { { 0, 1 }, { 1, 1 }, { 1, 0 }, { 1, -1 } }; /* texture.rs:20 (dy,dx) */

static void orc_glcm_features_dense(const int64_t *G, int L, float out[14])
{
    int64_t T = 0;
    for (int i = 0; i < L * L; ++i) T += G[i];
    if (T == 0) { for (int k = 0; k < 14; ++k) out[k] = NAN; return; }
    double *px = (double *)calloc((size_t)L, sizeof(double));
    double *pxy = (double *)calloc((size_t)(2 * L), sizeof(double));
    double *pd = (double *)calloc((size_t)L, sizeof(double));
    double t = (double)T;
    double sij = 0, ent = 0, asm_ = 0, idm = 0;
    for (int i = 0; i < L; ++i)
        for (int j = 0; j < L; ++j) {
            int64_t g = G[i * L + j];
            if (!g) continue;
            double pv = (double)g / t;
            px[i] += pv; pxy[i + j] += pv; pd[abs(i - j)] += pv;
            sij += (double)i * (double)j * pv;
            ent -= pv * log(pv);
            asm_ += pv * pv;
            idm += pv / (1.0 + (double)(i - j) * (double)(i - j));
        }
    double mu = 0, var = 0, hx = 0;
    for (int i = 0; i < L; ++i) mu += (double)i * px[i];
    for (int i = 0; i < L; ++i) { var += ((double)i - mu) * ((double)i - mu) * px[i]; if (px[i] > 0) hx -= px[i] * log(px[i]); }
    double sa = 0, sv = 0, se = 0;
    for (int k = 0; k < 2 * L - 1; ++k) sa += (double)k * pxy[k];
    for (int k = 0; k < 2 * L - 1; ++k) { sv += ((double)k - sa) * ((double)k - sa) * pxy[k]; if (pxy[k] > 0) se -= pxy[k] * log(pxy[k]); }
    double con = 0, dis = 0, dv = 0;
    for (int k = 0; k < L; ++k) { con += (double)k * k * pd[k]; dis += (double)k * pd[k]; }
    for (int k = 0; k < L; ++k) dv += ((double)k - dis) * ((double)k - dis) * pd[k];
    double hxy1 = 0, hxy2 = 0;
    for (int i = 0; i < L; ++i)
        for (int j = 0; j < L; ++j) {
            double pp = px[i] * px[j];
            if (pp > 0) {
                double lg = log(pp);
                hxy1 -= ((double)G[i * L + j] / t) * lg;
                hxy2 -= pp * lg;
            }
        }
    out[0] = (float)((sij - mu * mu) / var);
    out[1] = (float)con;
    out[2] = (float)dis;
    out[3] = (float)ent;
    out[4] = (float)asm_;
    out[5] = (float)sa;
    out[6] = (float)sv;
    out[7] = (float)se;
    out[8] = (float)var;
    out[9] = (float)idm;
    out[10] = (float)dis;
    out[11] = (float)dv;
    out[12] = (float)((ent - hxy1) / hx);
    { double a2 = 1.0 - exp(-2.0 * (hxy2 - ent)); out[13] = (float)sqrt(a2 > 0.0 ? a2 : 0.0); } /* clamp: HXY2-HXY is >= 0 up to rounding */
    free(px); free(pxy); free(pd);
}

/* counts for one (level, offset); G is L*L int64, zeroed here */
void orc_glcm_counts(const float *patch, const float *mask, int ps, int L, int dy, int dx, int64_t *G)
{
    const int P = ps * ps;
    memset(G, 0, sizeof(int64_t) * (size_t)L * L);
    for (int r = 0; r < ps; ++r)
        for (int c = 0; c < ps; ++c) {
            int r2 = r + dy, c2 = c + dx;
            if (r2 < 0 || r2 >= ps || c2 < 0 || c2 >= ps) continue;
            int a = r * ps + c, b = r2 * ps + c2;
            if (mask[a] == 0.0f || mask[b] == 0.0f) continue;
            int qa = orc_quant(orc_gray(patch[a], patch[P + a], patch[2 * P + a]), L);
            int qb = orc_quant(orc_gray(patch[b], patch[P + b], patch[2 * P + b]), L);
            G[qa * L + qb] += 1;
            G[qb * L + qa] += 1;
        }
}

/* G1 (texture.rs:25-167): out [224] for one nucleus; column order level-major, offset, 14 features */
void orc_glcm_one(const float *patch, const float *mask, int ps, float *out)
{
    int64_t *G = (int64_t *)malloc(sizeof(int64_t) * 254 * 254);
    int col = 0;
    for (int li = 0; li < 4; ++li)
        for (int oi = 0; oi < 4; ++oi) {
            int L = ORC_GLCM_LEVELS[li];
            orc_glcm_counts(patch, mask, ps, L, ORC_GLCM_OFFSETS[oi][0], ORC_GLCM_OFFSETS[oi][1], G);
            orc_glcm_features_dense(G, L, out + col);
            col += 14;
        }
    free(G);
}

/* ------------------------------------------------------------------------------------------
 * L2  tch_utils::glrlm::glrlm(gs,24,16,(dx,dy),Some(mask)) + glrlm_features(&glrlm,Some(&pixel_count))
 *   call site texture.rs:191-215.                                             [DEP-UNPINNED]
 *   Restated (Galloway 1975; Chu 1990; Dasarathy & Holder 1991): 24-level quantisation as above;
 *   a run is a maximal sequence of in-mask pixels of equal level along (dx,dy); runs longer than 16
 *   are counted in the last bin; R[g][l], g=1..24 (1-based in the weights), l=1..16.  Nr = sum R,
 *   Np = mask area.  "mid"/"extreme" gray-level emphases are not in the literature; restated with
 *   c=(L+1)/2, e(g)=(g-c)^2: extreme weight = e(g), mid weight = 1/(1+e(g)).
 *   Column order texture.rs:244-301.
 * ---------------------------------------------------------------------------------------- */
enum { ORC_RL_LEVELS = 24, ORC_RL_MAXLEN = 16 };
static const int ORC_RL_DIRS[4][2] = { { 1, 0 }, { 1, 1 }, { 0, 1 }, { -1, 1 } }; /* (dx,dy) texture.rs:176 */

void orc_glrlm_counts(const float *patch, const float *mask, int ps, int dx, int dy, int64_t *R /*[24*16]*/)
{
    const int P = ps * ps;
    memset(R, 0, sizeof(int64_t) * ORC_RL_LEVELS * ORC_RL_MAXLEN);
    int *q = (int *)malloc(sizeof(int) * (size_t)P);
    for (int a = 0; a < P; ++a)
        q[a] = (mask[a] != 0.0f) ? orc_quant(orc_gray(patch[a], patch[P + a], patch[2 * P + a]), ORC_RL_LEVELS) : -1;
    for (int r = 0; r < ps; ++r)
        for (int c = 0; c < ps; ++c) {
            int a = r * ps + c;
            if (q[a] < 0) continue;
            int rp = r - dy, cp = c - dx;
            if (rp >= 0 && rp < ps && cp >= 0 && cp < ps && q[rp * ps + cp] == q[a]) continue; /* not a run start */
            int len = 1, rn = r + dy, cn = c + dx;
            while (rn >= 0 && rn < ps && cn >= 0 && cn < ps && q[rn * ps + cn] == q[a]) { ++len; rn += dy; cn += dx; }
            int bin = (len > ORC_RL_MAXLEN ? ORC_RL_MAXLEN : len) - 1;
            R[q[a] * ORC_RL_MAXLEN + bin] += 1;
        }
    free(q);
}

static void orc_glrlm_features(const int64_t *R, double Np, float out[17])
{
    const int L = ORC_RL_LEVELS, M = ORC_RL_MAXLEN;
    double Nr = 0;
    for (int i = 0; i < L * M; ++i) Nr += (double)R[i];
    double cmid = ((double)L + 1.0) / 2.0;
    double sre = 0, lre = 0, gln = 0, rln = 0, lgre = 0, hgre = 0, srlge = 0, srhge = 0, lrlge = 0, lrhge = 0;
    double srmge = 0, lrmge = 0, srege = 0, lrege = 0, rlm = 0, rlv = 0;
    for (int g = 0; g < L; ++g) {
        double rowsum = 0;
        double gg = (double)(g + 1) * (double)(g + 1);
        double e = ((double)(g + 1) - cmid) * ((double)(g + 1) - cmid);
        double wmid = 1.0 / (1.0 + e), wext = e;
        for (int l = 0; l < M; ++l) {
            double v = (double)R[g * M + l];
            double ll = (double)(l + 1) * (double)(l + 1);
            rowsum += v;
            sre += v / ll; lre += v * ll; lgre += v / gg; hgre += v * gg;
            srlge += v / (ll * gg); srhge += v * gg / ll; lrlge += v * ll / gg; lrhge += v * ll * gg;
            srmge += v * wmid / ll; lrmge += v * wmid * ll; srege += v * wext / ll; lrege += v * wext * ll;
            rlm += v * (double)(l + 1);
        }
        gln += rowsum * rowsum;
    }
    for (int l = 0; l < M; ++l) {
        double colsum = 0;
        for (int g = 0; g < L; ++g) colsum += (double)R[g * M + l];
        rln += colsum * colsum;
    }
    rlm /= Nr;
    for (int g = 0; g < L; ++g)
        for (int l = 0; l < M; ++l) rlv += (double)R[g * M + l] * ((double)(l + 1) - rlm) * ((double)(l + 1) - rlm);
    rlv /= Nr;
    out[0] = (float)(sre / Nr);   out[1] = (float)(lre / Nr);
    out[2] = (float)(gln / Nr);   out[3] = (float)(rln / Nr);
    out[4] = (float)(lgre / Nr);  out[5] = (float)(hgre / Nr);
    out[6] = (float)(srlge / Nr); out[7] = (float)(srhge / Nr);
    out[8] = (float)(lrlge / Nr); out[9] = (float)(lrhge / Nr);
    out[10] = (float)(srmge / Nr); out[11] = (float)(lrmge / Nr);
    out[12] = (float)(srege / Nr); out[13] = (float)(lrege / Nr);
    out[14] = (float)(Nr / Np);   out[15] = (float)rlm; out[16] = (float)rlv;
}

void orc_glrlm_one(const float *patch, const float *mask, int ps, float *out /*[68]*/)
{
    int64_t R[ORC_RL_LEVELS * ORC_RL_MAXLEN];
    double Np = 0;
    for (int a = 0; a < ps * ps; ++a) Np += (double)mask[a];
    for (int d = 0; d < 4; ++d) {
        orc_glrlm_counts(patch, mask, ps, ORC_RL_DIRS[d][0], ORC_RL_DIRS[d][1], R);
        orc_glrlm_features(R, Np, out + 17 * d);
    }
}

/* ------------------------------------------------------------------------------------------
 * B2  tch_utils::gabor::apply_gabor_filter(gs, 8, 30, &[.5,1,2,4,6,8], 0.45)   [DEP-UNPINNED]
 *   call site texture.rs:333-334.  Restated (Gabor 1946 / Daugman 1985 real part, isotropic
 *   envelope): kernel size 30 sampled on t_k = -1 + 2k/29; for angle a (theta = a*pi/4, the
 *   column names use a*45 degrees, texture.rs:349) and frequency f:
 *       w[i][j] = exp(-(x^2+y^2)/(2*0.45^2)) * cos(2*pi*f*(x*cos(theta) + y*sin(theta))),
 *       x = t_j, y = t_i, rounded to f32.
 *   conv2d cross-correlation, zero padding 15 before / 14 after so the output keeps [ps,ps]
 *   (texture.rs:340 multiplies it by the mask): out[r][c] = sum w[i][j]*gray[r+i-15][c+j-15].
 *   channel = angle_idx*6 + freq_idx (texture.rs:346-350).
 * B1  masked mean / variance (texture.rs:339-344).  Evaluated in f64 at masked pixels only.
 * ---------------------------------------------------------------------------------------- */
enum { ORC_GB_ANGLES = 8, ORC_GB_FREQS = 6, ORC_GB_SIZE = 30 };
static const double ORC_GB_FREQ[6] = { 0.5, 1.0, 2.0, 4.0, 6.0, 8.0 }; /* texture.rs:320 */
static const double ORC_GB_SIGMA = 0.45;

void orc_gabor_bank(float *w /*[48][30][30]*/)
{
    for (int a = 0; a < ORC_GB_ANGLES; ++a)
        for (int f = 0; f < ORC_GB_FREQS; ++f) {
            double th = (double)a * ORC_PI / 4.0;
            double ct = cos(th), st = sin(th);
            float *wk = w + (size_t)(a * ORC_GB_FREQS + f) * ORC_GB_SIZE * ORC_GB_SIZE;
            for (int i = 0; i < ORC_GB_SIZE; ++i)
                for (int j = 0; j < ORC_GB_SIZE; ++j) {
                    double x = -1.0 + 2.0 * (double)j / (double)(ORC_GB_SIZE - 1);
                    double y = -1.0 + 2.0 * (double)i / (double)(ORC_GB_SIZE - 1);
                    double env = exp(-(x * x + y * y) / (2.0 * ORC_GB_SIGMA * ORC_GB_SIGMA));
                    wk[i * ORC_GB_SIZE + j] = (float)(env * cos(2.0 * ORC_PI * ORC_GB_FREQ[f] * (x * ct + y * st)));
                }
        }
}

void orc_gabor_one(const float *patch, const float *mask, int ps, const float *bank, float *out /*[96]*/)
{
    const int P = ps * ps, SZ = ORC_GB_SIZE, PAD = 15;
    float *gray = (float *)malloc(sizeof(float) * (size_t)P);
    for (int a = 0; a < P; ++a) gray[a] = orc_gray(patch[a], patch[P + a], patch[2 * P + a]);
    double area = 0;
    for (int a = 0; a < P; ++a) area += (double)mask[a];
    double *resp = (double *)malloc(sizeof(double) * (size_t)P);
    for (int ch = 0; ch < ORC_GB_ANGLES * ORC_GB_FREQS; ++ch) {
        const float *wk = bank + (size_t)ch * SZ * SZ;
        double s = 0;
        for (int r = 0; r < ps; ++r)
            for (int c = 0; c < ps; ++c) {
                if (mask[r * ps + c] == 0.0f) continue;
                double acc = 0;
                for (int i = 0; i < SZ; ++i) {
                    int rr = r + i - PAD;
                    if (rr < 0 || rr >= ps) continue;
                    for (int j = 0; j < SZ; ++j) {
                        int cc = c + j - PAD;
                        if (cc < 0 || cc >= ps) continue;
                        acc += (double)wk[i * SZ + j] * (double)gray[rr * ps + cc];
                    }
                }
                resp[r * ps + c] = acc;
                s += acc;
            }
        double mean = s / area, v = 0;
        for (int a = 0; a < P; ++a)
            if (mask[a] != 0.0f) v += (resp[a] - mean) * (resp[a] - mean);
        out[2 * ch] = (float)mean;
        out[2 * ch + 1] = (float)(v / area);
    }
    free(resp); free(gray);
}

/* ------------------------------------------------------------------------------------------
 * One reference batch, one feature set == FeatureSet::compute_features_batched
 * (features/mod.rs:12-28).  polygons are centred; patches [n,3,ps,ps] f32; masks [n,ps,ps] f32.
 * out [n, ncols(set)] row-major.
 * ---------------------------------------------------------------------------------------- */
int orc_compute_features_batched(int set, int n, const uint32_t *poly_offsets, const float *poly_xy,
                                 const float *patches, const float *masks, int ps, float *out)
{
    const size_t P = (size_t)ps * ps;
    switch (set) {
    case 0:
        for (int i = 0; i < n; ++i)
            orc_shape_one(poly_xy + 2 * (size_t)poly_offsets[i], (int)(poly_offsets[i + 1] - poly_offsets[i]),
                          masks + i * P, ps, out + (size_t)i * ORC_F_GEOMETRY, NULL);
        return 0;
    case 1:
        orc_color_batch(n, patches, masks, ps, out);
        return 0;
    case 2:
        for (int i = 0; i < n; ++i) orc_glcm_one(patches + i * 3 * P, masks + i * P, ps, out + (size_t)i * ORC_F_GLCM);
        return 0;
    case 3:
        for (int i = 0; i < n; ++i) orc_glrlm_one(patches + i * 3 * P, masks + i * P, ps, out + (size_t)i * ORC_F_GLRLM);
        return 0;
    case 4: {
        float *bank = (float *)malloc(sizeof(float) * 48 * 30 * 30);
        orc_gabor_bank(bank);
        for (int i = 0; i < n; ++i) orc_gabor_one(patches + i * 3 * P, masks + i * P, ps, bank, out + (size_t)i * ORC_F_GABOR);
        free(bank);
        return 0;
    }
    default:
        return -1;
    }
}

/* ------------------------------------------------------------------------------------------
 * The pipeline body of main.rs:146-151 for packed patches: par_chunks(batch_size) ->
 * preprocess_polygon + polygon raster (utils.rs:148-157) -> u8/255 (utils.rs:172) ->
 * every selected set in order -> column concat (main.rs:76-89).  Rows stay in input order.
 *   rings: raw GeoJSON outer rings in slide pixels; patches_u8 [n,3,ps,ps] planar.
 *   out_rows [n, sum ncols]; out_centroids [n,2] (x,y) for the K1 key.
 * OpenMP over batches mirrors rayon's one-batch-per-thread model (main.rs:146-151).
 * ---------------------------------------------------------------------------------------- */
int orc_extract(const int *sets, int nsets, int64_t n, const uint32_t *ring_offsets, const float *ring_xy,
                const uint8_t *patches_u8, int ps, int batch_size, float *out_rows, float *out_centroids,
                int num_threads)
{
    int F = 0;
    for (int s = 0; s < nsets; ++s) {
        if (sets[s] < 0 || sets[s] > 4) return -1;
        F += ORC_NCOLS[sets[s]];
    }
    const size_t P = (size_t)ps * ps;
    const int64_t nb = (n + batch_size - 1) / batch_size;
    (void)num_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(num_threads > 0 ? num_threads : 1)
#endif
    for (int64_t b = 0; b < nb; ++b) {
        int64_t i0 = b * batch_size;
        int m = (int)((n - i0) < batch_size ? (n - i0) : batch_size);
        uint32_t *off = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(m + 1));
        uint32_t base = ring_offsets[i0];
        for (int i = 0; i <= m; ++i) off[i] = ring_offsets[i0 + i] - base;
        float *cxy = (float *)malloc(sizeof(float) * 2 * (size_t)(off[m] > 0 ? off[m] : 1));
        float *masks = (float *)malloc(sizeof(float) * (size_t)m * P);
        float *patches = (float *)malloc(sizeof(float) * (size_t)m * 3 * P);
        for (int i = 0; i < m; ++i) {
            int V = (int)(off[i + 1] - off[i]);
            orc_preprocess_polygon(ring_xy + 2 * (size_t)ring_offsets[i0 + i], V, out_centroids + 2 * (i0 + i), cxy + 2 * (size_t)off[i]);
            orc_raster_polygon(cxy + 2 * (size_t)off[i], V, ps, masks + i * P);
        }
        const uint8_t *pu = patches_u8 + (size_t)i0 * 3 * P;
        for (size_t k = 0; k < (size_t)m * 3 * P; ++k) patches[k] = (float)pu[k] / 255.0f; /* utils.rs:172 */
        int col = 0;
        for (int s = 0; s < nsets; ++s) {
            int nc = ORC_NCOLS[sets[s]];
            float *tmp = (float *)malloc(sizeof(float) * (size_t)m * nc);
            orc_compute_features_batched(sets[s], m, off, cxy, patches, masks, ps, tmp);
            for (int i = 0; i < m; ++i) memcpy(out_rows + (size_t)(i0 + i) * F + col, tmp + (size_t)i * nc, sizeof(float) * nc);
            free(tmp);
            col += nc;
        }
        free(off); free(cxy); free(masks); free(patches);
    }
    return 0;
}

/* §8f-1, the slide variant, load_slide_dataset (utils.rs:97-122): region address `(c - ps/2) as u32` (Rust's float->int
   cast truncates, sends negatives and NaN to 0 and saturates), a ps x ps region from there, zero where the slide ends. */
static int64_t orc_as_u32(float v)
{
    if (!(v > 0.0f)) return 0;
    if (v >= 4294967296.0f) return 4294967295ll;
    return (int64_t)v;
}

void orc_crop_patch_slide(const uint8_t *image, int H, int W, float cx, float cy, int ps, uint8_t *patch)
{
    float half = (float)ps / 2.0f;
    int64_t x0 = orc_as_u32(cx - half), y0 = orc_as_u32(cy - half);       /* utils.rs:104-105 */
    memset(patch, 0, (size_t)3 * ps * ps);                                /* utils.rs:113-122: pad bottom/right */
    for (int ch = 0; ch < 3; ++ch)
        for (int pr = 0; pr < ps && y0 + pr < H; ++pr)
            for (int pc = 0; pc < ps && x0 + pc < W; ++pc)
                patch[((size_t)ch * ps + pr) * ps + pc] = image[((size_t)ch * H + (y0 + pr)) * W + (x0 + pc)];
}

/* Rust's `f32 as i64` (utils.rs:159-162): truncates, saturates, NaN -> 0; clamped further to +-2^40 (no image is that
   large), so the window arithmetic cannot overflow */
static int64_t orc_as_i64(float v)
{
    if (v != v) return 0;
    if (v >= 1099511627776.0f) return (int64_t)1 << 40;
    if (v <= -1099511627776.0f) return -((int64_t)1 << 40);
    return (int64_t)v;
}

/* §8f-1 patch addressing, load_image_dataset (utils.rs:141-206): crop a [3,H,W] u8 planar image at the
   truncated box around the centroid, zero-padded, into [3,ps,ps] u8. */
void orc_crop_patch(const uint8_t *image, int H, int W, float cx, float cy, int ps, uint8_t *patch)
{
    float half = (float)ps / 2.0f;
    int64_t top = orc_as_i64(cy - half), left = orc_as_i64(cx - half);       /* utils.rs:159-160 (as i64: trunc, saturating) */
    int64_t bottom = orc_as_i64(cy + half), right = orc_as_i64(cx + half);   /* utils.rs:161-162 */
    int64_t t0 = top > 0 ? top : 0, l0 = left > 0 ? left : 0;
    int64_t b0 = bottom < H ? bottom : H, r0 = right < W ? right : W;
    int64_t oy = -(top < 0 ? top : 0), ox = -(left < 0 ? left : 0);         /* utils.rs:180-181 */
    memset(patch, 0, (size_t)3 * ps * ps);
    for (int ch = 0; ch < 3; ++ch)
        for (int64_t r = t0; r < b0; ++r)
            for (int64_t c = l0; c < r0; ++c) {
                int64_t pr = oy + (r - t0), pc = ox + (c - l0);
                if (pr < ps && pc < ps)
                    patch[((size_t)ch * ps + pr) * ps + pc] = image[((size_t)ch * H + r) * W + c];
            }
}
