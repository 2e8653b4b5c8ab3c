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
//         90 degrees (A = 0): (ra, rb) = (G, G) once, then three column passes with (ca, cb) = (G cos B0 y, G cos B1 y);
//         0 degrees (B = 0): the same group shape on the TRANSPOSED nucleus (see gabor_nucleus).
//   8 row passes and 12 column passes serve the whole bank (25 + 24 unpaired ones at first).
//
// One 256-thread workgroup per nucleus (persistent): gray patch with a zero apron and the plane pair (zero aprons,
// odd row stride) live in LDS (76 KB, two workgroups per CU).  Both passes are REGISTER-TILED, and the four waves are
// AUTONOMOUS: wave w owns column strip w of the mask's bounding box for BOTH passes of every group -- its row pass fills
// those columns of the plane pair, its column passes read nothing else -- so there is no workgroup barrier between the
// groups (one between the two views; see gabor_nucleus).  A lane loads a (TS+29)-value window once and produces TS
// outputs; the 30 tap pairs of a pass are wave-uniform and arrive through the SCALAR cache as scalar operands of the
// packed FMAs (no LDS traffic, no staging).  Row passes cover only the mask's column range and the rows within 15 of
// the mask; the envelope row passes pair two output positions per packed FMA; per-strip sums are f32 about a
// per-filter shift (DC gain x mean masked gray), reduced with one DPP butterfly per pass and accumulated in f64.
// The next nucleus's patch and mask rows are requested while the current nucleus is processed.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>
#include <vector>

#include "exact_div.h"
#include "nfx_internal.h"
#include "wave_ops.h"

namespace nfx {

static constexpr int GB_SIZE = 30, GB_PAD = 15, GB_THREADS = 256, GB_WAVES = GB_THREADS / 64;
static_assert(GB_WAVES == 4, "four strips of TSR <= 16 columns tile the plane's 64 columns; the transposed-mask step hands 16 rows to a wave");
#ifndef NFX_GB_REDO_RATIO
#define NFX_GB_REDO_RATIO 8.0 // (s1/K)^2 / var beyond which a nucleus is done again with the measured mean as shift (k_gabor)
#endif
// The envelope groups' three column passes share their folded window (gabor_nucleus, column_passes_env); 0 = the generic passes
#ifndef NFX_GB_ENV_FUSED
#define NFX_GB_ENV_FUSED 1
#endif
#ifndef NFX_GB_ENV_MAX_CT
#define NFX_GB_ENV_MAX_CT 16 // strips of more rows than this take the generic passes (register budget: 6 CT accumulators beside the window)
#endif
static constexpr int GB_GW = NFX_ROW_WORD_PS + GB_SIZE - 1; // gray row with apron: 93 floats
static constexpr int GB_UW = NFX_ROW_WORD_PS + 1;           // U/W row stride (odd: conflict-free both ways)
static constexpr int GB_UH = NFX_ROW_WORD_PS + GB_SIZE - 1; // U/W rows incl. zero aprons: 93
static constexpr int GB_MAX_GROUPS = 16, GB_MAX_FILTERS = 24, GB_MAX_PASSES = 3;
static constexpr int GB_WSLOT = 64;                            // one pass: 30 weight pairs padded to 32 float2
static constexpr int GB_WBUF = GB_WSLOT * (1 + GB_MAX_PASSES); // row pass + up to three column passes

static constexpr int GB_META = 60;                             // spare floats 60..63 of a slot hold the group's / pass's metadata

typedef float f2 __attribute__((ext_vector_type(2)));

struct GaborGroup {
    int n_pass, pad0, pad1, pad2;
    int combine[4];            // per column pass: 0 -> responses (accA, accB); 1 -> (accA - accB, accA + accB)
    int filt[4][2];            // per column pass: the two filter slots its responses belong to (-1: none)
    float image[GB_WBUF];      // what the kernel reads with scalar loads: [row (ra,rb) x 30, n_pass, envelope flag | pass0 (ca,cb) x 30, dc0, dc1, meta | pass1 | pass2]
};
struct GaborFilter {
    int ch0, ch1;
    float dc; // sum of the 2-D kernel: response to a constant image of 1
    int pad;
};
struct GaborTables {
    int n_groups, n_filters, n_plain, pad1; // groups [n_plain, n_groups) run on the transposed nucleus
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

// TS output pairs of a 30-tap correlation from a (TS+29)-value register window.  The 30 tap pairs of a pass are the same for
// every lane and every nucleus: they come through the SCALAR cache (the bank is 8 KB of read-only global memory; `w` is a
// wave-uniform address) and enter the packed FMAs as scalar operands -- no LDS traffic, no vector registers, no staging and
// therefore no workgroup barrier for them.  Tap-major order, pinned: the TS packed FMAs of one tap hit TS different
// accumulators (independent, back-to-back issue); left alone, the scheduler serialises all 30 taps of one accumulator into
// a dependent chain.
// PAIRWIN = false: row pass, one gray window, two kernels: (a, b)[o] = sum_j (ra, rb)[j] * x[o + j]
// PAIRWIN = true:  column pass, a window of plane pairs:   (accA, accB)[o] = sum_j (ca, cb)[j] * (a, b)[o + j]
template <int TS, typename X, int XN>
static __device__ __forceinline__ void taps2(const X (&x)[XN], const float *__restrict__ w, f2 init, f2 (&acc)[TS])
{
    static_assert(XN >= TS + GB_SIZE - 1, "window");
    f2 wr[GB_SIZE];
#pragma unroll
    for (int j = 0; j < GB_SIZE; ++j) wr[j] = (f2){w[2 * j], w[2 * j + 1]};
#pragma unroll
    for (int o = 0; o < TS; ++o) acc[o] = init;
#pragma unroll
    for (int j = 0; j < GB_SIZE; ++j) {
#pragma unroll
        for (int o = 0; o < TS; ++o) {
            f2 xv;
            if constexpr (sizeof(X) == sizeof(float)) xv = (f2){x[o + j], x[o + j]};
            else xv = x[o + j];
            acc[o] = __builtin_elementwise_fma(wr[j], xv, acc[o]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

struct GaborShared {
    float *g;
    f2 *UW;          // plane pair, interleaved
};

// the mask as one orientation of the nucleus sees it (kept in LDS, so that nothing of the second view occupies
// registers while the first one runs): mask row l for lane l; bounding box; rows the row pass covers
struct GaborView {
    uint64_t wrow[64];
    int rmin, rmax, cmin, cmax, rlo, nrow_rp, pad0, pad1;
};

// All groups of one nucleus, WAVE-AUTONOMOUS: wave w owns column strip w of the bounding box (TSR columns; the four strips are
// disjoint and lie inside the plane's row) for BOTH passes of every group -- its row pass fills those columns of the plane pair,
// its column passes read nothing else -- so no wave ever waits for another one between the start of the first group and the
// end of a view's last: no workgroup barrier per group (sixteen per nucleus before), only wave-level LDS fences and one
// barrier between the two views.
//
// ROW PASS: lanes run along the rows within 15 of the mask, the strip along the columns.
// COLUMN PASS: the strip's TSR columns times S = 64 / TSR row strips of CT = ceil(nrow / S) rows are laid over the wave's
// lanes (item L = row strip * TSR + column): a box 40 x 40 keeps 60 lanes busy with 7 outputs each.
// No lane-divergent branches: lanes without work duplicate a neighbour's (row pass: same values to the same address) or carry
// an all-zero mask (column pass), which keeps every FMA block in one basic block.
//
// TWO VIEWS.  The 0-degree filters (B = 0) are the 90-degree ones of the TRANSPOSED nucleus, and masked mean / variance do
// not care about a transposition.  So they run as a second 90-degree-style group (one envelope row pass, three column
// passes for six filters -- four passes instead of the six that three row-then-envelope groups would take) on the
// transposed view: swapped bounding box, transposed mask, and a row-pass window that walks DOWN the gray image (lanes along
// its columns: still conflict-free; the rows above and below the patch are the zero apron a transposed image would have).
template <int N> using gb_int = std::integral_constant<int, N>;
// strip sizes a workgroup of GB_WAVES waves can meet: TSR = ceil(ncol / GB_WAVES) <= GB_MAX_TSR, CT = ceil(nrow / (64 / TSR)) <= GB_MAX_CT
static constexpr int GB_MAX_TSR = (NFX_ROW_WORD_PS + GB_WAVES - 1) / GB_WAVES;
static constexpr int GB_MAX_CT = (NFX_ROW_WORD_PS + (64 / GB_MAX_TSR) - 1) / (64 / GB_MAX_TSR);
#define NFX_GB_CASE(N, MAXN, F) case N: if constexpr (N <= MAXN) F(gb_int<N>{}); break;

template <bool REFINED>
static __device__ __forceinline__ void gabor_nucleus(const GaborShared &S, const GaborTables *__restrict__ tab, int n_groups, int n_plain, int ps,
                                                     const GaborView *s_view, float mu_gray, const float *s_shift, int t,
                                                     double (*s_part)[GB_MAX_FILTERS][2], int dbg)
{
    const int lane = t & 63, wv = t >> 6;
#pragma unroll 1
  for (int view = 0; view < 2; ++view) {
    const bool tr = view == 1; // wave-uniform
    const int g_begin = tr ? n_plain : 0, g_end = tr ? n_groups : n_plain;
    // the strips of the second view cut the plane's columns differently: every wave has to be done with the first view's
    // last column passes before any wave writes the second view's first row pass (the one barrier between the groups)
    if (tr) __syncthreads();
    if (g_begin >= g_end) continue;
    const GaborView &V = s_view[view];
    const int cmin = __builtin_amdgcn_readfirstlane(V.cmin), cmax = __builtin_amdgcn_readfirstlane(V.cmax);
    const int rlo = __builtin_amdgcn_readfirstlane(V.rlo), nrow_rp = __builtin_amdgcn_readfirstlane(V.nrow_rp);
    const int ncol = cmax - cmin + 1;
    // the wave's strip: TSR columns from c0; the strips of the four waves tile [cs, cs + 4 TSR) inside the plane's 64 columns
    int tsr = (ncol + GB_WAVES - 1) / GB_WAVES;
    tsr = tsr < 3 ? 3 : tsr; // 3 .. 16
    const int cs = cmin < NFX_ROW_WORD_PS - GB_WAVES * tsr ? cmin : NFX_ROW_WORD_PS - GB_WAVES * tsr;
    const int c0 = cs + wv * tsr;
    if (c0 > cmax || (dbg & 4)) continue; // wave-uniform: no column of the box in this wave's strip (its partial sums stay zero)
    // row pass geometry: lanes run along rows (LDS stride GB_GW, odd)
    const int rr = rlo + (lane < nrow_rp ? lane : nrow_rp - 1);
    const float *src = tr ? &S.g[rr + GB_PAD] : &S.g[rr * GB_GW + c0]; // gray column c-15 sits at apron index c; transposed: column rr
    f2 *du = &S.UW[(rr + GB_PAD) * GB_UW + c0];
    // column pass geometry: item = (row strip, column of the wave's strip).  The strips are cut from the rows that hold a mask pixel in
    // THIS WAVE'S columns, not from the bounding box: a round nucleus's outer strips are shorter than its box, and CT -- the rows per
    // lane, i.e. the wave's cost per pass -- follows the wave's own extent (0.84 of the box-cut strips' rows per lane by a count over the
    // synthetic nuclei; -1.7 % time).  The row pass keeps the box's rows.  (Cutting every QUARTER of the wave's columns from its own
    // extent -- 0.70 by the same count -- was built and measured: +4 % time: the items' start rows then differ inside a wave, the window
    // loads collide in the LDS banks, and the geometry no longer fits the registers.)
    int rmin_w, nrow_w;
    {
        const uint64_t stripmask = ((tsr >= 64 ? ~0ull : ((1ull << tsr) - 1ull)) << c0);
        const uint64_t rows_w = __ballot((V.wrow[lane] & stripmask) != 0ull);
        if (rows_w == 0ull) continue; // wave-uniform: no mask pixel in the strip
        rmin_w = __ffsll((long long)rows_w) - 1;
        nrow_w = 64 - __clzll((long long)rows_w) - rmin_w;
    }
    const int nstrip_max = 64 / tsr;                             // 4 .. 21
    int ct = (nrow_w + nstrip_max - 1) / nstrip_max;             // 1 .. 16
    ct = ct < 2 ? 2 : ct;
    const int nstrip = (nrow_w + ct - 1) / ct;
    const int items = nstrip * tsr;                              // <= 64
    const bool item = lane < items;
    const int li = item ? lane : items - 1;
    const int strip = li / tsr, cj = li - strip * tsr;
    const int r0 = rmin_w + strip * ct;
    const int c = c0 + cj;                                       // <= 63
    int r0c = r0 < ps - ct ? r0 : ps - ct; // clamped strips overlap: rows below r0 are masked out
    r0c = r0c > 0 ? r0c : 0;
    const f2 *su = &S.UW[r0c * GB_UW + c]; // row (r0c - 15) + 15
    // bit q: pixel (r0c + q, c) is in the mask and belongs to this strip.  Column c of this view's mask is row c of
    // the other view's: one LDS word instead of CT lane reads
    unsigned mbits;
    {
        const uint64_t column = s_view[view ^ 1].wrow[c];
        const int skip = r0 - r0c < 31 ? r0 - r0c : 31; // clamped strips overlap: rows below r0 belong to the strip before
        mbits = (unsigned)(column >> r0c) & ((1u << ct) - 1u) & ~((1u << skip) - 1u);
        if (!item) mbits = 0u;
    }

#pragma unroll 1
    for (int g = g_begin; g < g_end; ++g) {
        const float *__restrict__ img = tab->groups[g].image; // wave-uniform: the taps arrive through the scalar cache
        // ---- row pass ---------------------------------------------------------------------------------------------
        const int env_flag = __float_as_int(img[GB_META + 1]);       // 0: a genuine kernel pair; 1: envelope group whose column taps are mirror-symmetric; 2: envelope group, taps not bit-symmetric
        const bool envelope = env_flag != 0; // both row kernels are the bare envelope (90 / 0 degree groups)
        const int n_pass = __float_as_int(img[GB_META]);
        const bool fused = NFX_GB_ENV_FUSED && env_flag == 1 && n_pass == GB_MAX_PASSES && ct <= NFX_GB_ENV_MAX_CT; // wave-uniform
        auto load_window = [&](auto win_tag, float (&x)[decltype(win_tag)::value]) {
            constexpr int WIN = decltype(win_tag)::value;
            if (tr) { // window element j is gray row c0 + j - 15 (wave-uniform): outside the patch it is the apron's zero
                // (the first row is made opaque here so that the 40-odd range tests are evaluated on the spot: hoisted out
                // of the group loop they would sit in registers the FMA blocks need)
                int row0 = __builtin_amdgcn_readfirstlane(c0 - GB_PAD);
                asm volatile("" : "+s"(row0));
                const float *col = src + row0 * GB_GW;
#pragma unroll
                for (int j = 0; j < WIN; ++j) x[j] = (unsigned)(row0 + j) < (unsigned)ps ? col[j * GB_GW] : 0.0f;
            } else {
#pragma unroll
                for (int j = 0; j < WIN; ++j) x[j] = src[j];
            }
        };
        auto row_pass = [&](auto ts_tag) {
            constexpr int TS = decltype(ts_tag)::value, WIN = TS + GB_SIZE - 1;
            f2 acc[TS];
            float x[WIN];
            load_window(gb_int<WIN>{}, x);
            taps2<TS, float>(x, img, (f2){0.0f, 0.0f}, acc);
#pragma unroll
            for (int q = 0; q < TS; ++q) du[q] = acc[q];
        };
        // (ra, rb) = (G, G): a packed FMA would compute the same number twice.  Here it carries two OUTPUT POSITIONS instead --
        // acc[p] = (E[2p], E[2p+1]), fed with the window pair (x[2p+j], x[2p+j+1]) -- half the FMAs for the same chain per
        // output (tap order and roundings unchanged: the plane is bit-identical).  For an odd TS the last pair's second half
        // is computed from one element past the window (inside s_g, or the range-tested zero) and not stored.
        auto row_pass_envelope = [&](auto ts_tag) {
            constexpr int TS = decltype(ts_tag)::value, TP = (TS + 1) / 2, WIN = 2 * TP + GB_SIZE - 1;
            float x[WIN];
            load_window(gb_int<WIN>{}, x);
            f2 wr[GB_SIZE];
#pragma unroll
            for (int j = 0; j < GB_SIZE; ++j) wr[j] = (f2){img[2 * j], img[2 * j + 1]};
            f2 acc[TP];
#pragma unroll
            for (int q = 0; q < TP; ++q) acc[q] = (f2){0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < GB_SIZE; ++j) {
#pragma unroll
                for (int q = 0; q < TP; ++q) acc[q] = __builtin_elementwise_fma(wr[j], (f2){x[2 * q + j], x[2 * q + j + 1]}, acc[q]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (fused) { // ONE plane, as floats in the first half of the wave's own columns of the plane pair (column_passes_env reads it back)
                float *duf = reinterpret_cast<float *>(S.UW) + (rr + GB_PAD) * (2 * GB_UW) + 2 * c0;
#pragma unroll
                for (int q = 0; q < TP; ++q) {
                    duf[2 * q] = acc[q].x;
                    if (2 * q + 1 < TS) duf[2 * q + 1] = acc[q].y;
                }
            } else {
#pragma unroll
                for (int q = 0; q < TP; ++q) {
                    du[2 * q] = (f2){acc[q].x, acc[q].x};
                    if (2 * q + 1 < TS) du[2 * q + 1] = (f2){acc[q].y, acc[q].y};
                }
            }
        };
        if (!(dbg & 1)) {
            if (envelope) {
                switch (tsr) {
                NFX_GB_CASE(3, GB_MAX_TSR, row_pass_envelope)
                NFX_GB_CASE(4, GB_MAX_TSR, row_pass_envelope)
                NFX_GB_CASE(5, GB_MAX_TSR, row_pass_envelope)
                NFX_GB_CASE(6, GB_MAX_TSR, row_pass_envelope)
                NFX_GB_CASE(7, GB_MAX_TSR, row_pass_envelope)
                NFX_GB_CASE(8, GB_MAX_TSR, row_pass_envelope)
                NFX_GB_CASE(9, GB_MAX_TSR, row_pass_envelope)
                NFX_GB_CASE(10, GB_MAX_TSR, row_pass_envelope)
                NFX_GB_CASE(11, GB_MAX_TSR, row_pass_envelope)
                NFX_GB_CASE(12, GB_MAX_TSR, row_pass_envelope)
                NFX_GB_CASE(13, GB_MAX_TSR, row_pass_envelope)
                NFX_GB_CASE(14, GB_MAX_TSR, row_pass_envelope)
                NFX_GB_CASE(15, GB_MAX_TSR, row_pass_envelope)
                NFX_GB_CASE(16, GB_MAX_TSR, row_pass_envelope)
                default: break;
                }
            } else {
                switch (tsr) {
                NFX_GB_CASE(3, GB_MAX_TSR, row_pass)
                NFX_GB_CASE(4, GB_MAX_TSR, row_pass)
                NFX_GB_CASE(5, GB_MAX_TSR, row_pass)
                NFX_GB_CASE(6, GB_MAX_TSR, row_pass)
                NFX_GB_CASE(7, GB_MAX_TSR, row_pass)
                NFX_GB_CASE(8, GB_MAX_TSR, row_pass)
                NFX_GB_CASE(9, GB_MAX_TSR, row_pass)
                NFX_GB_CASE(10, GB_MAX_TSR, row_pass)
                NFX_GB_CASE(11, GB_MAX_TSR, row_pass)
                NFX_GB_CASE(12, GB_MAX_TSR, row_pass)
                NFX_GB_CASE(13, GB_MAX_TSR, row_pass)
                NFX_GB_CASE(14, GB_MAX_TSR, row_pass)
                NFX_GB_CASE(15, GB_MAX_TSR, row_pass)
                NFX_GB_CASE(16, GB_MAX_TSR, row_pass)
                default: break;
                }
            }
        }
        wave_lds_fence(); // the wave's own plane columns: written above, read below
        // ---- column passes ------------------------------------------------------------------------------------------
        auto column_passes = [&](auto ct_tag) {
            constexpr int CT = decltype(ct_tag)::value, CWIN = CT + GB_SIZE - 1;
            f2 x[CWIN];
#pragma unroll
            for (int j = 0; j < CWIN; ++j) x[j] = su[j * GB_UW];
#pragma unroll 1
            for (int p = 0; p < n_pass; ++p) {
                const float *__restrict__ wp = img + (1 + p) * GB_WSLOT;
                const int meta = __float_as_int(wp[GB_META + 2]); // combine | slot0 << 8 | slot1 << 16 (255: none)
                f2 a = {0.0f, 0.0f}, b = {0.0f, 0.0f};
                f2 acc[CT];
                // per-filter shift (k_gabor): DC gain x mean masked gray -- two scalar operands; in the second launch, over the few nuclei
                // whose first result was far off that guess, the mean the first launch measured (LDS)
                float sh0, sh1;
                if constexpr (REFINED) {
                    const int fs0 = (meta >> 8) & 255, fs1 = (meta >> 16) & 255;
                    sh0 = s_shift[fs0 != 255 ? fs0 : 0];
                    sh1 = fs1 != 255 ? s_shift[fs1] : 0.0f;
                } else {
                    sh0 = wp[GB_META] * mu_gray;
                    sh1 = wp[GB_META + 1] * mu_gray;
                }
                // the accumulators start at minus the shift (for the mixed pair: A - B starts at -sh0, A + B at -sh1), so
                // the responses come out already shifted; masked partial sums of the strip's responses follow (f32
                // inside the strip and the wave, f64 across waves)
                if (meta & 1) {
                    taps2<CT, f2>(x, wp, (f2){-0.5f * (sh0 + sh1), 0.5f * (sh0 - sh1)}, acc);
#pragma unroll
                    for (int q = 0; q < CT; ++q) {
                        const float mq = ((mbits >> q) & 1u) ? 1.0f : 0.0f; // the mask as a 0/1 factor: packed multiply
                        // (A - B, A + B) in ONE packed add: both halves take A from the low word, B from the high word, the low half
                        // negates it (the compiler builds the pair with two adds and a move)
                        f2 pm;
                        asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,1]" : "=v"(pm) : "v"(acc[q]));
                        const f2 d = pm * (f2){mq, mq};
                        a += d;
                        b = __builtin_elementwise_fma(d, d, b);
                    }
                } else {
                    taps2<CT, f2>(x, wp, (f2){-sh0, -sh1}, acc);
#pragma unroll
                    for (int q = 0; q < CT; ++q) {
                        const float mq = ((mbits >> q) & 1u) ? 1.0f : 0.0f;
                        const f2 d = acc[q] * (f2){mq, mq};
                        a += d;
                        b = __builtin_elementwise_fma(d, d, b);
                    }
                }
                // each lane holds shifted sums (small numbers), so an f32 wave reduction is exact enough
                float tot[4] = {a.x, b.x, a.y, b.y}; // four wave totals in one butterfly; lane j < 4 ends up with total scatter_index<4>(j)
                const float mine = wave_reduce_scatter(tot, lane);
                if (lane < 4) {
                    const int which = scatter_index<4>(lane);           // 0: a.x, 1: b.x, 2: a.y, 3: b.y
                    const int f = (meta >> (8 + 8 * (which >> 1))) & 255; // filter slot of the response
                    if (f != 255) s_part[wv][f][which & 1] = (double)mine;
                }
            }
        };
        // ENVELOPE GROUPS (90 degrees; 0 degrees on the transposed view): the plane is ONE plane (both row kernels are the bare envelope) and
        // all six column kernels G(y) cos(B y) are EVEN about the middle of the 30-tap grid (t_{29-j} = -t_j), so taps j and 29 - j share a
        // multiplication: per output 15 additions x[o+j] + x[o+29-j], shared by the three passes, and 3 x 15 packed FMAs instead of 3 x 30.
        // The window is a plain float window (CT + 29 registers instead of twice that), which is what lets the three accumulator
        // sets (6 CT registers) sit beside it.  Tap-major order pinned as in taps2; the tap pairs of tap j are three scalar pairs.
        auto column_passes_env = [&](auto ct_tag) {
            constexpr int CT = decltype(ct_tag)::value, CWIN = CT + GB_SIZE - 1, HALF = GB_SIZE / 2;
            float x[CWIN];
            {
                const float *suf = reinterpret_cast<const float *>(S.UW) + r0c * (2 * GB_UW) + 2 * c0 + cj;
#pragma unroll
                for (int j = 0; j < CWIN; ++j) x[j] = suf[j * (2 * GB_UW)];
            }
            const float *__restrict__ wp0 = img + GB_WSLOT, *__restrict__ wp1 = img + 2 * GB_WSLOT, *__restrict__ wp2 = img + 3 * GB_WSLOT;
            int meta[GB_MAX_PASSES];
            f2 acc[GB_MAX_PASSES][CT];
#pragma unroll
            for (int p = 0; p < GB_MAX_PASSES; ++p) {
                const float *__restrict__ wp = img + (1 + p) * GB_WSLOT;
                meta[p] = __float_as_int(wp[GB_META + 2]);
                float sh0, sh1;
                if constexpr (REFINED) {
                    const int fs0 = (meta[p] >> 8) & 255, fs1 = (meta[p] >> 16) & 255;
                    sh0 = s_shift[fs0 != 255 ? fs0 : 0];
                    sh1 = fs1 != 255 ? s_shift[fs1] : 0.0f;
                } else {
                    sh0 = wp[GB_META] * mu_gray;
                    sh1 = wp[GB_META + 1] * mu_gray;
                }
#pragma unroll
                for (int o = 0; o < CT; ++o) acc[p][o] = (f2){-sh0, -sh1};
            }
#pragma unroll
            for (int j = 0; j < HALF; ++j) {
                const f2 w0 = {wp0[2 * j], wp0[2 * j + 1]}, w1 = {wp1[2 * j], wp1[2 * j + 1]}, w2 = {wp2[2 * j], wp2[2 * j + 1]};
#pragma unroll
                for (int o = 0; o < CT; ++o) {
                    const float tf = x[o + j] + x[o + GB_SIZE - 1 - j];
                    const f2 tt = {tf, tf};
                    acc[0][o] = __builtin_elementwise_fma(w0, tt, acc[0][o]);
                    acc[1][o] = __builtin_elementwise_fma(w1, tt, acc[1][o]);
                    acc[2][o] = __builtin_elementwise_fma(w2, tt, acc[2][o]);
                }
                // the accumulators are pinned at the end of every tap: left alone, instruction selection lays the block out pass-major
                // (all of pass 0's FMAs, the 15 CT folded sums kept alive -- in scratch memory -- for passes 1 and 2)
#pragma unroll
                for (int o = 0; o < CT; ++o) {
                    asm volatile("" : "+v"(acc[0][o]));
                    asm volatile("" : "+v"(acc[1][o]));
                    asm volatile("" : "+v"(acc[2][o]));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int p = 0; p < GB_MAX_PASSES; ++p) {
                f2 a = {0.0f, 0.0f}, b = {0.0f, 0.0f};
#pragma unroll
                for (int q = 0; q < CT; ++q) {
                    const float mq = ((mbits >> q) & 1u) ? 1.0f : 0.0f;
                    const f2 d = acc[p][q] * (f2){mq, mq};
                    a += d;
                    b = __builtin_elementwise_fma(d, d, b);
                }
                float tot[4] = {a.x, b.x, a.y, b.y};
                const float mine = wave_reduce_scatter(tot, lane);
                if (lane < 4) {
                    const int which = scatter_index<4>(lane);
                    const int f = (meta[p] >> (8 + 8 * (which >> 1))) & 255;
                    if (f != 255) s_part[wv][f][which & 1] = (double)mine;
                }
            }
        };
        if (fused && !(dbg & 2)) {
            switch (ct) {
            NFX_GB_CASE(2, (GB_MAX_CT < NFX_GB_ENV_MAX_CT ? GB_MAX_CT : NFX_GB_ENV_MAX_CT), column_passes_env)
            NFX_GB_CASE(3, (GB_MAX_CT < NFX_GB_ENV_MAX_CT ? GB_MAX_CT : NFX_GB_ENV_MAX_CT), column_passes_env)
            NFX_GB_CASE(4, (GB_MAX_CT < NFX_GB_ENV_MAX_CT ? GB_MAX_CT : NFX_GB_ENV_MAX_CT), column_passes_env)
            NFX_GB_CASE(5, (GB_MAX_CT < NFX_GB_ENV_MAX_CT ? GB_MAX_CT : NFX_GB_ENV_MAX_CT), column_passes_env)
            NFX_GB_CASE(6, (GB_MAX_CT < NFX_GB_ENV_MAX_CT ? GB_MAX_CT : NFX_GB_ENV_MAX_CT), column_passes_env)
            NFX_GB_CASE(7, (GB_MAX_CT < NFX_GB_ENV_MAX_CT ? GB_MAX_CT : NFX_GB_ENV_MAX_CT), column_passes_env)
            NFX_GB_CASE(8, (GB_MAX_CT < NFX_GB_ENV_MAX_CT ? GB_MAX_CT : NFX_GB_ENV_MAX_CT), column_passes_env)
            NFX_GB_CASE(9, (GB_MAX_CT < NFX_GB_ENV_MAX_CT ? GB_MAX_CT : NFX_GB_ENV_MAX_CT), column_passes_env)
            NFX_GB_CASE(10, (GB_MAX_CT < NFX_GB_ENV_MAX_CT ? GB_MAX_CT : NFX_GB_ENV_MAX_CT), column_passes_env)
            NFX_GB_CASE(11, (GB_MAX_CT < NFX_GB_ENV_MAX_CT ? GB_MAX_CT : NFX_GB_ENV_MAX_CT), column_passes_env)
            NFX_GB_CASE(12, (GB_MAX_CT < NFX_GB_ENV_MAX_CT ? GB_MAX_CT : NFX_GB_ENV_MAX_CT), column_passes_env)
            NFX_GB_CASE(13, (GB_MAX_CT < NFX_GB_ENV_MAX_CT ? GB_MAX_CT : NFX_GB_ENV_MAX_CT), column_passes_env)
            NFX_GB_CASE(14, (GB_MAX_CT < NFX_GB_ENV_MAX_CT ? GB_MAX_CT : NFX_GB_ENV_MAX_CT), column_passes_env)
            NFX_GB_CASE(15, (GB_MAX_CT < NFX_GB_ENV_MAX_CT ? GB_MAX_CT : NFX_GB_ENV_MAX_CT), column_passes_env)
            NFX_GB_CASE(16, (GB_MAX_CT < NFX_GB_ENV_MAX_CT ? GB_MAX_CT : NFX_GB_ENV_MAX_CT), column_passes_env)
            default: break;
            }
        } else if (!(dbg & 2)) {
            switch (ct) {
            NFX_GB_CASE(2, GB_MAX_CT, column_passes)
            NFX_GB_CASE(3, GB_MAX_CT, column_passes)
            NFX_GB_CASE(4, GB_MAX_CT, column_passes)
            NFX_GB_CASE(5, GB_MAX_CT, column_passes)
            NFX_GB_CASE(6, GB_MAX_CT, column_passes)
            NFX_GB_CASE(7, GB_MAX_CT, column_passes)
            NFX_GB_CASE(8, GB_MAX_CT, column_passes)
            NFX_GB_CASE(9, GB_MAX_CT, column_passes)
            NFX_GB_CASE(10, GB_MAX_CT, column_passes)
            NFX_GB_CASE(11, GB_MAX_CT, column_passes)
            NFX_GB_CASE(12, GB_MAX_CT, column_passes)
            NFX_GB_CASE(13, GB_MAX_CT, column_passes)
            NFX_GB_CASE(14, GB_MAX_CT, column_passes)
            NFX_GB_CASE(15, GB_MAX_CT, column_passes)
            NFX_GB_CASE(16, GB_MAX_CT, column_passes)
            default: break;
            }
        }
        wave_lds_fence(); // the next group's row pass overwrites the columns read above
    }
  }
}

// ---- patch sizes 65..128: the same kernel on TILES ------------------------------------------------------------------------------
// The planes of a 128 x 128 patch do not fit the LDS, but a response only depends on the 30 x 30 pixels around it: a 64 x 64 WINDOW of
// the patch (zeros outside the patch, as the reference's padding has them) gives the exact responses of its central 35 x 35 pixels
// (15 taps before, 14 after).  k_gabor_plan cuts a wide nucleus's mask into such central squares -- up to 16 tiles at 128 x 128 (225 at 512 x 512), only those that hold
// mask pixels -- k_gabor<T, GB_TILES> runs every tile as a nucleus of its own (window as the patch, central mask as the mask, the
// WHOLE mask's shift) and writes the tile's masked sums, and k_gabor_finish adds a nucleus's tiles in a fixed order and forms the
// 96 columns.
static constexpr int GB_MAIN = 0, GB_REFINED = 1, GB_TILES = 2;
static constexpr int GB_CENTRAL = NFX_ROW_WORD_PS - (GB_SIZE - 1);                 // 35: rows / columns [15, 50) of a window
static __host__ __device__ constexpr int gb_max_tiles(int ps) { return ((ps + GB_CENTRAL - 1) / GB_CENTRAL) * ((ps + GB_CENTRAL - 1) / GB_CENTRAL); } // per nucleus: 16 at 128, 225 at 512
static_assert(gb_max_tiles(NFX_MAX_PATCH_SIZE) <= 256, "k_gabor_plan: one thread per candidate tile");
struct GaborTile { int nucleus, wr, wc, pad; };                                    // window origin in patch coordinates (may be negative)
struct GaborTileArgs {
    const GaborTile *tiles; const int *tile_count; const float *mu_gray; double *tile_sums; // [tile][GB_MAX_FILTERS][2]
};

// bits [off, off + 64) of a mask row of W words, zeros outside the row
static __device__ __forceinline__ uint64_t window_bits(const uint64_t *row, int W, int off)
{
    if (off <= -64 || off >= 64 * W) return 0ull;
    if (off < 0) return row[0] << (-off);
    const int wi = off >> 6, sh = off & 63;
    uint64_t v = row[wi] >> sh;
    if (sh && wi + 1 < W) v |= row[wi + 1] << (64 - sh);
    return v;
}

// MODE = GB_MAIN: the launch over all n nuclei; a nucleus one of whose variances came out of a badly centred sum is appended to
// redo_list.  GB_REFINED: the launch over redo_list (count read on the device), shifts = the means already in the rows.
// GB_TILES: the launch over the tiles of wide nuclei (above).
template <typename T, int MODE>
__global__ __launch_bounds__(GB_THREADS, GB_THREADS / 128) void k_gabor(int64_t n, int ps, const T *__restrict__ patches,
                                                         const uint64_t *__restrict__ mask_rows,
                                                         const GaborTables *__restrict__ tab, float *__restrict__ out,
                                                         int F, int col0, int dbg, int *__restrict__ redo_list, int *__restrict__ redo_count,
                                                         GaborTileArgs ta)
{
    constexpr bool REFINED = MODE == GB_REFINED, TILES = MODE == GB_TILES;
    __shared__ float s_g[NFX_ROW_WORD_PS * GB_GW + 16];
    __shared__ f2 s_UW[GB_UH * GB_UW]; // plane pair; row r lives at index r + GB_PAD; aprons stay zero
    __shared__ uint64_t s_rows[64 + 16];
    __shared__ double s_part[GB_WAVES][GB_MAX_FILTERS][2];
    __shared__ float s_shift[GB_MAX_FILTERS];
    __shared__ int s_redo;
    __shared__ float s_gsum[GB_WAVES];
    __shared__ GaborView s_view[2];

    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int P = ps * ps;
    for (int k = t; k < NFX_ROW_WORD_PS * GB_GW + 16; k += GB_THREADS) s_g[k] = 0.0f; // apron stays zero
    for (int k = t; k < GB_UH * GB_UW; k += GB_THREADS) s_UW[k] = (f2){0.0f, 0.0f};
    if (t < 64 + 16) s_rows[t] = 0ull;
    const int n_groups = tab->n_groups, n_filters = tab->n_filters, n_plain = tab->n_plain;
    __syncthreads();
    GaborShared S;
    S.g = s_g; S.UW = s_UW;
    // u8 patches whose rows are whole 16-byte pieces come in with one 16-byte load per channel and thread
    const bool vec = !TILES && sizeof(T) == 1 && (ps & 15) == 0 && (reinterpret_cast<uintptr_t>(patches) & 15) == 0;
    const int pse = TILES ? NFX_ROW_WORD_PS : ps; // the patch the nucleus code below sees: a tile's window is 64 x 64

    // the next nucleus's mask row and (16-byte-piece) patch travel in registers while the current nucleus is processed: their
    // HBM latency hides behind a whole nucleus of work instead of stalling the workgroup at the top of every iteration
    uint64_t row_next = 0ull;
    uint4 r4n = make_uint4(0, 0, 0, 0), g4n = r4n, b4n = r4n;
    auto request = [&](int64_t i2) {
        if (TILES) return; // (a tile's window rows are read where they are staged)
        if (t < 64) row_next = t < ps ? mask_rows[i2 * ps + t] : 0ull;
        if (vec && t < P / 16) {
            const uint4 *src = reinterpret_cast<const uint4 *>(patches + (size_t)i2 * 3 * P) + t;
            r4n = src[0]; g4n = src[P / 16]; b4n = src[2 * (P / 16)];
        }
    };
    if (REFINED) n = *redo_count;            // the list's length (this launch follows the one that filled it, on the same stream)
    if (TILES) n = *ta.tile_count;
    auto nucleus_at = [&](int64_t j) -> int64_t { return REFINED ? (int64_t)redo_list[j] : (TILES ? (int64_t)ta.tiles[j].nucleus : j); };
    if (t == 0) s_redo = 0;
    if ((int64_t)blockIdx.x < n) request(nucleus_at(blockIdx.x));
    for (int64_t j = blockIdx.x; j < n; j += gridDim.x) {
        const int64_t i = nucleus_at(j);
        int wr = 0, wc = 0;
        if (TILES) { // the window's mask rows: the wide mask's bits [wc, wc + 64) of row wr + t, central square only
            wr = ta.tiles[j].wr; wc = ta.tiles[j].wc;
            if (t < 64) {
                const int r = wr + t;
                uint64_t bits = 0ull;
                if (t >= GB_PAD && t < GB_PAD + GB_CENTRAL && r >= 0 && r < ps)
                    bits = window_bits(mask_rows + ((size_t)i * ps + r) * ((ps + 63) >> 6), (ps + 63) >> 6, wc) & (((1ull << GB_CENTRAL) - 1ull) << GB_PAD);
                s_rows[t] = bits;
            }
        } else if (t < 64) s_rows[t] = row_next;
        __syncthreads();
        const T *pp = patches + (size_t)i * 3 * P;
        float gsum = 0.0f; // masked gray sum -> the shift that keeps the f32 partial sums well conditioned
        if (TILES) { // the 64 x 64 window of the patch, zeros outside it
            for (int p = t; p < NFX_ROW_WORD_PS * NFX_ROW_WORD_PS; p += GB_THREADS) {
                const int rr = p >> 6, cc = p & 63, pr = wr + rr, pc = wc + cc;
                float gs = 0.0f;
                if (pr >= 0 && pr < ps && pc >= 0 && pc < ps) {
                    const int q = pr * ps + pc;
                    gs = BPx<T>::ld(pp, q) + BPx<T>::ld(pp, P + q);
                    gs = gs + BPx<T>::ld(pp, 2 * P + q);
                    gs = sizeof(T) == 1 ? third_of_u8_sum(gs) : gs / 3.0f; // texture.rs:332
                }
                s_g[rr * GB_GW + cc + GB_PAD] = gs;
            }
        } else if (vec) {
            for (int k = t; k < P / 16; k += GB_THREADS) {
                const uint4 r4 = r4n, g4 = g4n, b4 = b4n; // P / 16 <= 256: one piece per thread and channel
                const unsigned rr[4] = {r4.x, r4.y, r4.z, r4.w}, gg[4] = {g4.x, g4.y, g4.z, g4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
                const int p0 = 16 * k, r = p0 / ps, c0 = p0 - r * ps;
                const unsigned mrow = (unsigned)(s_rows[r] >> c0) & 0xFFFFu;
                float *dst = &s_g[r * GB_GW + c0 + GB_PAD];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int sh = 8 * (q & 3);
                    float gs = u8_unit((rr[q >> 2] >> sh) & 255u) + u8_unit((gg[q >> 2] >> sh) & 255u); // utils.rs:172
                    gs = gs + u8_unit((bb[q >> 2] >> sh) & 255u);
                    gs = third_of_u8_sum(gs); // texture.rs:332
                    dst[q] = gs;
                    gsum += ((mrow >> q) & 1u) ? gs : 0.0f;
                }
            }
        } else {
            int r = t / ps, c = t - r * ps;
            const int dr = GB_THREADS / ps, dc = GB_THREADS - dr * ps;
            for (int p = t; p < P; p += GB_THREADS) {
                float gs = BPx<T>::ld(pp, p) + BPx<T>::ld(pp, P + p);
                gs = gs + BPx<T>::ld(pp, 2 * P + p);
                gs = sizeof(T) == 1 ? third_of_u8_sum(gs) : gs / 3.0f; // texture.rs:332
                s_g[r * GB_GW + c + GB_PAD] = gs;
                if ((s_rows[r] >> c) & 1ull) gsum += gs;
                c += dc; r += dr;
                if (c >= ps) { c -= ps; r += 1; }
            }
        }
        if (j + (int64_t)gridDim.x < n) request(nucleus_at(j + gridDim.x));
        gsum = wave_total_f32(gsum);
        // mask bounding box and area: lane r looks at row r, the wave votes
        const uint64_t wrow = s_rows[lane];
        const uint64_t rowsany = __ballot(wrow != 0ull);
        const int area = wave_total_i32(__popcll(wrow));
        int olo = (int)(unsigned)wrow, ohi = (int)(unsigned)(wrow >> 32);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { olo |= __shfl_xor(olo, o, 64); ohi |= __shfl_xor(ohi, o, 64); }
        const uint64_t orr = ((uint64_t)(unsigned)ohi << 32) | (uint64_t)(unsigned)olo;
        const int rmin = rowsany ? __ffsll((long long)rowsany) - 1 : pse, rmax = rowsany ? 63 - __clzll((long long)rowsany) : -1;
        float *o = out + i * (int64_t)F + col0;
        if (TILES && area == 0) { __syncthreads(); continue; } // (the plan emits no empty tile)
        if (area == 0) { // 0/0 in texture.rs:340-344
            if (t < 96) o[t] = __builtin_nanf("");
            __syncthreads();
            continue;
        }
        const int cmin = __ffsll((long long)orr) - 1, cmax = 63 - __clzll((long long)orr);
        auto put_view = [&](GaborView &V, bool with_rows, uint64_t rows_of_lane, int r_lo, int r_hi, int c_lo, int c_hi) {
            if (with_rows) V.wrow[lane] = rows_of_lane;
            if (lane == 0) {
                V.rmin = r_lo; V.rmax = r_hi; V.cmin = c_lo; V.cmax = c_hi;
                V.rlo = r_lo - GB_PAD > 0 ? r_lo - GB_PAD : 0;
                const int rhi = r_hi + (GB_SIZE - 1 - GB_PAD) < pse - 1 ? r_hi + (GB_SIZE - 1 - GB_PAD) : pse - 1;
                V.nrow_rp = rhi - V.rlo + 1;
            }
        };
        // transposed mask: lane c collects column c, one bit per mask row (three instructions a row); wave w does rows
        // 16w .. 16w+15 and puts its 16 bits down as one quarter of the lane's word
        {
            unsigned acc = 0u;
            const unsigned wlo = (unsigned)wrow, whi = (unsigned)(wrow >> 32);
            const int Ra = rmin > 16 * wv ? rmin : 16 * wv, Rb = rmax < 16 * wv + 15 ? rmax : 16 * wv + 15;
            for (int R = Ra; R <= Rb; ++R) {
                const unsigned rl = (unsigned)__builtin_amdgcn_readlane((int)wlo, R), rh = (unsigned)__builtin_amdgcn_readlane((int)whi, R);
                acc |= (((lane < 32 ? rl : rh) >> (lane & 31)) & 1u) << (R & 15);
            }
            reinterpret_cast<unsigned short *>(s_view[1].wrow)[4 * lane + wv] = (unsigned short)acc;
        }
        if (wv == 0) { // visible to the other waves after the barrier below
            put_view(s_view[0], true, wrow, rmin, rmax, cmin, cmax);
            put_view(s_view[1], false, 0ull, cmin, cmax, rmin, rmax);
        }
        if (lane == 0) s_gsum[wv] = gsum;
        if (lane < GB_MAX_FILTERS) { s_part[wv][lane][0] = 0.0; s_part[wv][lane][1] = 0.0; } // a wave whose strip lies outside the box adds nothing
        // The masked sums are f32 about a per-filter SHIFT (the accumulators start at minus the shift), so that sum d^2 carries no
        // cancellation: var = s2/K - (s1/K)^2 carries the f32 sums' relative error (3..6e-7: <= 16 terms per lane, a six-level
        // butterfly) times 1 + (s1/K)^2 / var.  First launch: shift = DC gain x mean masked gray -- within a few standard
        // deviations of the mean response whenever the 30-tap kernel sees mostly the patch.  Where (s1/K)^2 > 8 var all the same --
        // patches smaller than the kernel, whose zero padding moves every response off that guess ((s1/K)^2 up to 2700 var at
        // 17 x 17: 4e-4 relative on the frequency-0.5 variances), flat patches -- the nucleus goes on a list, and a SECOND launch
        // does the listed nuclei again with the means just measured as shifts (this launch's own code stays what it was: the
        // same loop with the second attempt inside it cost 2.3 % on every nucleus).
        if (REFINED && t < n_filters) { const GaborFilter &Fl = tab->filters[t]; s_shift[t] = o[2 * (Fl.ch0 >= 0 ? Fl.ch0 : Fl.ch1)]; }
        __syncthreads();
        float gtot = 0.0f;
#pragma unroll
        for (int w = 0; w < GB_WAVES; ++w) gtot += s_gsum[w];
        const float mu_gray = TILES ? ta.mu_gray[i] : gtot / (float)area; // a tile's shift is its NUCLEUS's: the tiles' sums then simply add up
        gabor_nucleus<REFINED>(S, tab, n_groups, n_plain, pse, s_view, mu_gray, s_shift, t, s_part, dbg);
        __syncthreads();
        if (TILES) { // the tile's masked sums; k_gabor_finish adds a nucleus's tiles and forms the columns
            if (t < n_filters) {
                double a = 0.0, b = 0.0;
#pragma unroll
                for (int w = 0; w < GB_WAVES; ++w) { a += s_part[w][t][0]; b += s_part[w][t][1]; }
                ta.tile_sums[((size_t)j * GB_MAX_FILTERS + t) * 2] = a;
                ta.tile_sums[((size_t)j * GB_MAX_FILTERS + t) * 2 + 1] = b;
            }
            __syncthreads();
            continue;
        }
        // ---- mean / variance (texture.rs:339-344) from the shifted sums: mean = c + s1/K, var = s2/K - (s1/K)^2 -----
        if (t < n_filters) {
            const GaborFilter &Fl = tab->filters[t];
            const double K = (double)area;
            const double shift = REFINED ? (double)s_shift[t] : (double)(Fl.dc * mu_gray);
            double a = 0.0, b = 0.0;
#pragma unroll
            for (int w = 0; w < GB_WAVES; ++w) { a += s_part[w][t][0]; b += s_part[w][t][1]; }
            const double m1 = a / K;
            double var = b / K - m1 * m1;
            var = var > 0.0 ? var : 0.0;
            const double mean = shift + m1;
            if (MODE == GB_MAIN && m1 * m1 > NFX_GB_REDO_RATIO * var) s_redo = 1; // (several threads may store the same 1)
            if (Fl.ch0 >= 0) { o[2 * Fl.ch0] = (float)mean; o[2 * Fl.ch0 + 1] = (float)var; }
            if (Fl.ch1 >= 0) { o[2 * Fl.ch1] = (float)mean; o[2 * Fl.ch1 + 1] = (float)var; }
        }
        __syncthreads();
        if (MODE == GB_MAIN && t == 0 && s_redo) { // after the barrier: every filter's verdict is in
            redo_list[atomicAdd(redo_count, 1)] = (int)i;
            s_redo = 0;
        }
    }
}

// One workgroup per wide nucleus: mask area and masked gray sum (the shift), bounding box, and the tiles -- central squares of
// GB_CENTRAL pixels laid over the bounding box from its top-left corner, those without a mask pixel left out -- written as ONE run
// of the tile list (tile_begin / tile_n per nucleus), in row-major order: the order k_gabor_finish adds them in.
template <typename T, int MAXPS>
__global__ __launch_bounds__(256) void k_gabor_plan(int64_t n, int ps, const T *__restrict__ patches, const uint64_t *__restrict__ mask_rows,
                                                    GaborTile *__restrict__ tiles, int *__restrict__ tile_count, int *__restrict__ tile_begin,
                                                    int *__restrict__ tile_n, float *__restrict__ mu_gray, int *__restrict__ area_out)
{
    constexpr int MAXW = MAXPS / 64;
    __shared__ uint64_t s_rows[MAXPS * MAXW];
    __shared__ float s_g[4];
    __shared__ int s_box[4], s_k[4], s_votes[4], s_base;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int P = ps * ps, W = (ps + 63) >> 6;
    for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
        for (int k = t; k < W * ps; k += 256) s_rows[k] = mask_rows[(size_t)i * ps * W + k];
        __syncthreads();
        const T *pp = patches + (size_t)i * 3 * P;
        float gsum = 0.0f;
        int cnt = 0;
        for (int p = t; p < P; p += 256) {
            const int r = p / ps, c = p - r * ps;
            if ((s_rows[r * W + (c >> 6)] >> (c & 63)) & 1ull) {
                float gs = BPx<T>::ld(pp, p) + BPx<T>::ld(pp, P + p);
                gs = gs + BPx<T>::ld(pp, 2 * P + p);
                gsum += sizeof(T) == 1 ? third_of_u8_sum(gs) : gs / 3.0f;
                ++cnt;
            }
        }
        gsum = wave_total_f32(gsum);
        cnt = wave_total_i32(cnt);
        if (lane == 0) { s_g[wv] = gsum; s_k[wv] = cnt; }
        if (t == 0) { // bounding box (a serial look at the row words: nothing beside the pixels above)
            int rmin = ps, rmax = -1, cmin = ps, cmax = -1;
            for (int r = 0; r < ps; ++r) {
                bool any = false;
                for (int w = 0; w < W; ++w) {
                    const uint64_t a = s_rows[r * W + w];
                    if (!a) continue;
                    any = true;
                    const int lo = 64 * w + __ffsll((long long)a) - 1, hi = 64 * w + 63 - __clzll((long long)a);
                    cmin = lo < cmin ? lo : cmin; cmax = hi > cmax ? hi : cmax;
                }
                if (any) { rmin = r < rmin ? r : rmin; rmax = r; }
            }
            s_box[0] = rmin; s_box[1] = rmax; s_box[2] = cmin; s_box[3] = cmax;
        }
        __syncthreads();
        const int area = (s_k[0] + s_k[1]) + (s_k[2] + s_k[3]);
        const int rmin = s_box[0], rmax = s_box[1], cmin = s_box[2], cmax = s_box[3];
        const int tr = area ? (rmax - rmin) / GB_CENTRAL + 1 : 0, tc = area ? (cmax - cmin) / GB_CENTRAL + 1 : 0;
        // thread a * tc + b looks at tile (a, b): does its central square hold a mask pixel?  (tr * tc <= gb_max_tiles(MAXPS) <= 256)
        bool has = false;
        if (t < tr * tc) {
            const int a = t / tc, b = t - a * tc, r0 = rmin + a * GB_CENTRAL, c0 = cmin + b * GB_CENTRAL;
            for (int r = r0; r < r0 + GB_CENTRAL && r < ps && !has; ++r)
                has = (window_bits(&s_rows[r * W], W, c0) & ((1ull << GB_CENTRAL) - 1ull)) != 0ull;
        }
        const uint64_t votes = __ballot(has);
        if (lane == 0) s_votes[wv] = __popcll(votes);
        __syncthreads();
        if (t == 0) {
            const int m = (s_votes[0] + s_votes[1]) + (s_votes[2] + s_votes[3]);
            s_base = m ? atomicAdd(tile_count, m) : 0;
            tile_begin[i] = s_base; tile_n[i] = m;
            mu_gray[i] = ((s_g[0] + s_g[1]) + (s_g[2] + s_g[3])) / (float)area;
            area_out[i] = area;
        }
        __syncthreads();
        if (has) { // row-major order of the tiles: the waves before this one, then the lanes before this one
            const int a = t / tc, b = t - a * tc;
            int before = 0;
            for (int w = 0; w < wv; ++w) before += s_votes[w];
            const int slot = s_base + before + __popcll(votes & ((1ull << lane) - 1ull));
            tiles[slot] = GaborTile{(int)i, rmin + a * GB_CENTRAL - GB_PAD, cmin + b * GB_CENTRAL - GB_PAD, 0};
        }
        __syncthreads();
    }
}

// One thread per (nucleus, filter): the tiles' sums in list order, then mean / variance exactly as k_gabor forms them.
__global__ __launch_bounds__(256) void k_gabor_finish(int64_t n, const GaborTables *__restrict__ tab, const int *__restrict__ tile_begin,
                                                      const int *__restrict__ tile_n, const float *__restrict__ mu_gray, const int *__restrict__ area,
                                                      const double *__restrict__ tile_sums, float *__restrict__ out, int F, int col0)
{
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = g / GB_MAX_FILTERS;
    const int f = (int)(g - i * GB_MAX_FILTERS);
    if (i >= n || f >= tab->n_filters) return;
    const GaborFilter &Fl = tab->filters[f];
    float *o = out + i * (int64_t)F + col0;
    const int K = area[i];
    float mean_f = __builtin_nanf(""), var_f = __builtin_nanf(""); // empty mask: 0/0 in texture.rs:340-344
    if (K > 0) {
        double a = 0.0, b = 0.0;
        const int t0 = tile_begin[i], m = tile_n[i];
        for (int k = 0; k < m; ++k) {
            a += tile_sums[((size_t)(t0 + k) * GB_MAX_FILTERS + f) * 2];
            b += tile_sums[((size_t)(t0 + k) * GB_MAX_FILTERS + f) * 2 + 1];
        }
        const double shift = (double)(Fl.dc * mu_gray[i]);
        const double m1 = a / (double)K;
        double var = b / (double)K - m1 * m1;
        var = var > 0.0 ? var : 0.0;
        mean_f = (float)(shift + m1);
        var_f = (float)var;
    }
    if (Fl.ch0 >= 0) { o[2 * Fl.ch0] = mean_f; o[2 * Fl.ch0 + 1] = var_f; }
    if (Fl.ch1 >= 0) { o[2 * Fl.ch1] = mean_f; o[2 * Fl.ch1 + 1] = var_f; }
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
    // (2) A = 0 (90 degrees), then B = 0 (0 degrees) on the transposed nucleus, where x and y swap roles: one row pass with
    // the bare envelope (in both halves), two frequencies per column pass
    for (int view = 0; view < 2 && ok; ++view) {
        if (view == 1) T.n_plain = ng;
        std::vector<Uniq *> col;
        for (auto &u : uniq)
            if (!u.placed && (view == 0 ? u.A == 0.0 : u.B == 0.0)) col.push_back(&u);
        auto wave_number = [&](const Uniq *u) { return view == 0 ? u->B : u->A; };
        for (size_t i = 0; i < col.size() && ok; i += 2 * GB_MAX_PASSES) {
            GaborGroup *G = new_group();
            if (!G) { ok = false; break; }
            for (int j = 0; j < GB_SIZE; ++j) set_row(*G, j, env(tk(j)), env(tk(j)));
            for (int p = 0; p < GB_MAX_PASSES && i + 2 * p < col.size(); ++p) {
                Uniq *u = col[i + 2 * p], *v = i + 2 * p + 1 < col.size() ? col[i + 2 * p + 1] : nullptr;
                for (int j = 0; j < GB_SIZE; ++j)
                    set_col(*G, p, j, env(tk(j)) * cos(wave_number(u) * tk(j)), v ? env(tk(j)) * cos(wave_number(v) * tk(j)) : 0.0);
                G->filt[p][0] = u->slot; G->filt[p][1] = v ? v->slot : -1;
                u->placed = true;
                if (v) v->placed = true;
                G->n_pass = p + 1;
            }
            // the row pass pairs output positions (flag != 0); the column passes may share a FOLDED window (column_passes_env multiplies taps j and
            // 29 - j by ONE weight) only if the f32 taps really are mirror images of each other -- tk(j) and tk(29 - j) are computed independently in
            // f64 and rounded, so this is checked, not assumed: flag 1 = symmetric bit for bit, 2 = not (the generic passes run)
            int flag = 1;
            for (int p = 0; p < G->n_pass; ++p)
                for (int j = 0; j < GB_SIZE / 2; ++j)
                    for (int h = 0; h < 2; ++h)
                        if (memcmp(&G->image[(1 + p) * GB_WSLOT + 2 * j + h], &G->image[(1 + p) * GB_WSLOT + 2 * (GB_SIZE - 1 - j) + h], sizeof(float)) != 0) flag = 2;
            memcpy(&G->image[GB_META + 1], &flag, sizeof(int));
        }
    }
    for (auto &u : uniq) ok = ok && u.placed;
    for (int g = 0; g < ng; ++g) { // metadata the kernel reads from the LDS image
        GaborGroup &G = T.groups[g];
        memcpy(&G.image[GB_META], &G.n_pass, sizeof(int));
        for (int p = 0; p < G.n_pass; ++p) {
            float *slot = &G.image[(1 + p) * GB_WSLOT + GB_META];
            const int f0 = G.filt[p][0], f1 = G.filt[p][1];
            slot[0] = f0 >= 0 ? T.filters[f0].dc : 0.0f;
            slot[1] = f1 >= 0 ? T.filters[f1].dc : 0.0f;
            const int meta = (G.combine[p] ? 1 : 0) | ((f0 >= 0 ? f0 : 255) << 8) | ((f1 >= 0 ? f1 : 255) << 16);
            memcpy(&slot[2], &meta, sizeof(int));
        }
    }
    T.n_groups = ok ? ng : GB_MAX_GROUPS + 1;
}

// The bank is the same for every context: built once per process (contexts are created and first used concurrently,
// one per host thread -- main.rs:146-151), immutable afterwards.
static const GaborTables &gabor_tables()
{
    static GaborTables T;
    static std::once_flag once;
    std::call_once(once, [] { build_tables(T); });
    return T;
}

hipError_t prepare_gabor(nfx_ctx *ctx)
{
    if (ctx->gabor_tab) return hipSuccess;
    if (tables_locked(ctx)) return hipErrorNotReady; // device-memory calls on a reserved context and captures never allocate: nfx_ctx_reserve builds the tables
    const GaborTables &T = gabor_tables();
    if (T.n_groups < 1 || T.n_groups > GB_MAX_GROUPS || T.n_filters < 1 || T.n_filters > GB_MAX_FILTERS) return hipErrorInvalidValue;
    float *tab = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&tab), sizeof(GaborTables));
    if (e != hipSuccess) return e;
    e = hipMemcpy(tab, &T, sizeof(GaborTables), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(tab); return e; }
    ctx->gabor_tab = tab; // published only once it is complete
    return hipSuccess;
}

// wide patches are planned, run and finished in chunks of nuclei: the tile list and the tiles' sums of one chunk (100 MB of sums:
// 16 384 nuclei of up to 16 tiles at 128 x 128, 1 165 nuclei of up to 225 tiles at 512 x 512)
static int64_t gbt_chunk(int ps) { return ps <= NFX_WIDE_PS ? 16384 : (16384 * 16) / gb_max_tiles(ps); }
struct WideScratch { int *tile_count, *tile_begin, *tile_n, *area; float *mu; GaborTile *tiles; double *sums; size_t bytes; };
static WideScratch wide_scratch(void *scratch, int64_t nc, int ps)
{
    char *const base = reinterpret_cast<char *>(scratch);
    char *p = base;
    auto take = [&](size_t bytes) { char *q = p; p += (bytes + 255) & ~(size_t)255; return q; };
    const size_t mt = (size_t)gb_max_tiles(ps);
    WideScratch w;
    w.tile_count = reinterpret_cast<int *>(take(64));
    w.tile_begin = reinterpret_cast<int *>(take(4 * (size_t)nc));
    w.tile_n = reinterpret_cast<int *>(take(4 * (size_t)nc));
    w.area = reinterpret_cast<int *>(take(4 * (size_t)nc));
    w.mu = reinterpret_cast<float *>(take(4 * (size_t)nc));
    w.tiles = reinterpret_cast<GaborTile *>(take(sizeof(GaborTile) * (size_t)nc * mt));
    w.sums = reinterpret_cast<double *>(take(sizeof(double) * 2 * GB_MAX_FILTERS * (size_t)nc * mt));
    w.bytes = (size_t)(p - base);
    return w;
}
size_t gabor_scratch_bytes(const Job &job)
{
    const size_t n = (size_t)(job.n > 0 ? job.n : 1);
    if (job.ps <= NFX_ROW_WORD_PS) return sizeof(int) * (n + 16); // redo list + its length
    const size_t chunk = (size_t)gbt_chunk(job.ps);
    return wide_scratch(nullptr, (int64_t)(n < chunk ? n : chunk), job.ps).bytes; // (pointer arithmetic only)
}

template <typename T>
static hipError_t launch_gabor_wide(nfx_ctx *ctx, const Job &job, int col0, void *scratch, const GaborTables *tab)
{
    const size_t P3 = 3 * (size_t)job.ps * job.ps;
    const int W = mask_words_per_row(job.ps);
    const int64_t chunk = gbt_chunk(job.ps);
    const int64_t mt = gb_max_tiles(job.ps);
    for (int64_t i0 = 0; i0 < job.n; i0 += chunk) {
        const int64_t nc = job.n - i0 < chunk ? job.n - i0 : chunk;
        const WideScratch w = wide_scratch(scratch, job.n < chunk ? job.n : chunk, job.ps);
        const T *patches = reinterpret_cast<const T *>(job.patches) + (size_t)i0 * P3;
        const uint64_t *rows = job.mask_rows + (size_t)i0 * job.ps * W;
        float *out = job.out + (size_t)i0 * job.F;
        hipError_t e = hipMemsetAsync(w.tile_count, 0, sizeof(int), ctx->stream);
        if (e != hipSuccess) return e;
        const dim3 pgrid((unsigned)(nc < 256 * 8 ? nc : 256 * 8));
        if (job.ps <= NFX_WIDE_PS)
            hipLaunchKernelGGL((k_gabor_plan<T, NFX_WIDE_PS>), pgrid, dim3(256), 0, ctx->stream, nc, job.ps, patches, rows, w.tiles, w.tile_count, w.tile_begin, w.tile_n, w.mu, w.area);
        else
            hipLaunchKernelGGL((k_gabor_plan<T, NFX_MAX_PATCH_SIZE>), pgrid, dim3(256), 0, ctx->stream, nc, job.ps, patches, rows, w.tiles, w.tile_count, w.tile_begin, w.tile_n, w.mu, w.area);
        // the tile launch walks a list whose length only the device knows: sixteen rounds of the resident workgroups at most
        const int64_t max_tiles = nc * mt;
        const unsigned grid = (unsigned)(max_tiles < 256 * 2 * 16 ? max_tiles : 256 * 2 * 16);
        GaborTileArgs ta{w.tiles, w.tile_count, w.mu, w.sums};
        hipLaunchKernelGGL((k_gabor<T, GB_TILES>), dim3(grid), dim3(GB_THREADS), 0, ctx->stream, (int64_t)0, job.ps, patches, rows, tab, out, job.F, col0, 0,
                           (int *)nullptr, (int *)nullptr, ta);
        hipLaunchKernelGGL(k_gabor_finish, dim3((unsigned)((nc * GB_MAX_FILTERS + 255) / 256)), dim3(256), 0, ctx->stream, nc, tab, w.tile_begin, w.tile_n, w.mu, w.area,
                           w.sums, out, job.F, col0);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_gabor(nfx_ctx *ctx, const Job &job, int col0, void *scratch)
{
    if (job.n == 0) return hipSuccess;
    if (job.n > 0x7FFFFFFFll) return hipErrorInvalidValue; // the redo list holds 32-bit nucleus indices
    hipError_t e = prepare_gabor(ctx);
    if (e != hipSuccess) return e;
    ScopedTiming t(ctx, KS_GABOR);
    const GaborTables *tab = reinterpret_cast<const GaborTables *>(ctx->gabor_tab);
    if (job.ps > NFX_ROW_WORD_PS)
        return job.patches_f32 ? launch_gabor_wide<float>(ctx, job, col0, scratch, tab) : launch_gabor_wide<uint8_t>(ctx, job, col0, scratch, tab);
    // 64 rounds of the resident workgroups: nuclei differ several-fold in work, and with few large blocks the launch ends when
    // the unluckiest block does (2 rounds: +5 %); a block still amortises its LDS set-up over some forty nuclei of a large call
    const int64_t maxb = 256 * 2 * 64;
    const unsigned grid = (unsigned)(job.n < maxb ? job.n : maxb);
#ifdef NFX_EXPERIMENTS // phase-skipping builds for timing experiments only (wrong results): 1 skip row pass, 2 skip column pass, 4 skip all groups
    const char *dbg_env = getenv("NFX_GABOR_DBG");
    const int dbg = dbg_env ? atoi(dbg_env) : 0;
#else
    const int dbg = 0;
#endif
    int *redo_count = reinterpret_cast<int *>(scratch), *redo_list = redo_count + 16;
    e = hipMemsetAsync(redo_count, 0, sizeof(int), ctx->stream);
    if (e != hipSuccess) return e;
    // the second launch walks the list the first one filled: two workgroups per CU (its length is only known on the device; on
    // the BASELINE patch sizes it holds a percent of the nuclei or less, on patches smaller than the filter nearly all of them)
    const unsigned grid2 = (unsigned)(job.n < 256 * 2 * 4 ? job.n : 256 * 2 * 4);
    const GaborTileArgs none{nullptr, nullptr, nullptr, nullptr};
#ifdef NFX_EXPERIMENTS // NFX_GABOR_PAD = bytes of dynamic LDS nobody uses: 8192 leaves room for ONE workgroup per CU (co-residency A/Bs, profiles/EXPERIMENTS.md)
    static const unsigned pad = [] { const char *e = getenv("NFX_GABOR_PAD"); const long v = e ? atol(e) : 0; return (unsigned)(v < 0 ? 0 : (v > 65536 ? 65536 : v)); }();
#else
    constexpr unsigned pad = 0;
#endif
    if (job.patches_f32) {
        const float *pf = reinterpret_cast<const float *>(job.patches);
        hipLaunchKernelGGL((k_gabor<float, GB_MAIN>), dim3(grid), dim3(GB_THREADS), pad, ctx->stream, job.n, job.ps, pf, job.mask_rows, tab, job.out, job.F, col0, dbg, redo_list, redo_count, none);
        hipLaunchKernelGGL((k_gabor<float, GB_REFINED>), dim3(grid2), dim3(GB_THREADS), pad, ctx->stream, job.n, job.ps, pf, job.mask_rows, tab, job.out, job.F, col0, dbg, redo_list, redo_count, none);
    } else {
        const uint8_t *pb = reinterpret_cast<const uint8_t *>(job.patches);
        hipLaunchKernelGGL((k_gabor<uint8_t, GB_MAIN>), dim3(grid), dim3(GB_THREADS), pad, ctx->stream, job.n, job.ps, pb, job.mask_rows, tab, job.out, job.F, col0, dbg, redo_list, redo_count, none);
        hipLaunchKernelGGL((k_gabor<uint8_t, GB_REFINED>), dim3(grid2), dim3(GB_THREADS), pad, ctx->stream, job.n, job.ps, pb, job.mask_rows, tab, job.out, job.F, col0, dbg, redo_list, redo_count, none);
    }
    return hipGetLastError();
}

}  // namespace nfx
