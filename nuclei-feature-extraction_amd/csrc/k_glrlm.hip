// k_glrlm.hip -- the "glrlm" feature set: 4 directions x 17 run-length features.
//
// Replaces (reference file:line):
//   GLRLMFeatureSet::compute_features_batched    texture.rs:181-310 (GLRLM_LEVELS=24,
//                                                GLRLM_MAX_LENGTH=16, DIRECTIONS, 174-176)
//   tch_utils::glrlm::{glrlm, glrlm_features}    call site texture.rs:193-215
//
// ONE WAVEFRONT PER NUCLEUS (persistent, grid-stride; four independent waves per workgroup, each with a
// private LDS slice and no s_barrier -- a wave's LDS ops execute in order).  Runs are found BIT-PARALLEL:
//   * staging: u8 patches arrive as 16-byte pieces (lane L owns the 16 pixels of piece L + 64k in all three
//     channels; the NEXT nucleus's pieces are requested before the current ones are used, so HBM latency hides
//     behind a whole nucleus), are quantised to 24 levels in registers and written to an LDS byte map with 0xFF
//     outside the mask; then (lane = column) for each of the four directions a ballot builds the row's
//     "continues its predecessor" word C_d[r] (pixel and predecessor in the mask with equal level), filed
//     under lane r;
//   * per direction: run starts are S = M & ~C (lane = row).  S and C are re-filed so that a lane holds one LINE of the direction
//     -- row, column or diagonal (modulo 64) -- with the run direction along its bits: a per-row rotation turns the diagonals
//     into columns, a 64 x 64 bit transpose across the wave (six butterfly stages) turns columns into rows.  The length of
//     the run that starts at bit p is then 1 + the number of set C bits right above p: a funnel shift and a find-first-clear,
//     with no counting loop over the steps of the run;
//   * each lane walks its line's starts (ffs) and bumps one of the 384 run-length bins per RUN.
// The 17 features are weighted sums over the bins: with bin = lane + 64k a lane always sees the same run
// length l = lane % 16 and six fixed gray levels, so every weight is a register computed once per kernel;
// row/column sums for the non-uniformity features are 16-lane / 4-lane butterfly reductions.
#include "exact_div.h"
#include "nfx_internal.h"
#include "wave_ops.h"

namespace nfx {

static constexpr int RL_LEVELS = 24, RL_MAXLEN = 16, RL_BINS = RL_LEVELS * RL_MAXLEN; // 384 = 6 * 64
__constant__ int kRlDirs[4][2] = {{1, 0}, {1, 1}, {0, 1}, {-1, 1}};                  // (dx,dy) texture.rs:176

template <typename T> struct RPx;
template <> struct RPx<uint8_t> {
    static __device__ __forceinline__ float ld(const uint8_t *p, size_t i) { return u8_unit(p[i]); }
};
template <> struct RPx<float> {
    static __device__ __forceinline__ float ld(const float *p, size_t i) { return p[i]; }
};

static __device__ __forceinline__ double rsum(double v) { return wave_total_f64(v); }

// bit i of the result: byte i of a equals byte i of b (exact zero-byte test of a ^ b, then the four flag bits gathered)
static __device__ __forceinline__ unsigned eq4(unsigned a, unsigned b)
{
    const unsigned x = a ^ b;
    const unsigned y = ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x; // bit 7 of a byte set iff the byte is non-zero
    const unsigned t = (~y & 0x80808080u) >> 7;                // bits 0, 8, 16, 24
    return (t | (t >> 7) | (t >> 14) | (t >> 21)) & 0xFu;
}
static __device__ __forceinline__ unsigned eq16(const unsigned (&a)[4], unsigned b0, unsigned b1, unsigned b2, unsigned b3)
{
    return eq4(a[0], b0) | (eq4(a[1], b1) << 4) | (eq4(a[2], b2) << 8) | (eq4(a[3], b3) << 12);
}

static __device__ __forceinline__ uint64_t rotr64(uint64_t x, unsigned s) // s in 0..63
{
    const unsigned lo = (unsigned)x, hi = (unsigned)(x >> 32);
    const bool sw = s & 32u;
    const unsigned a = sw ? hi : lo, b = sw ? lo : hi; // rotating by 32 swaps the halves; alignbit takes the amount modulo 32
    return ((uint64_t)__builtin_amdgcn_alignbit(a, b, s) << 32) | __builtin_amdgcn_alignbit(b, a, s);
}

// 64 x 64 bit transpose of two matrices held one 64-bit row per lane: bit c of lane r moves to bit r of lane c.  Six butterfly
// stages; stage j swaps index bit j of the lane with index bit j of the bit position (j x j blocks change places between lanes
// l and l ^ j), and the stages below 32 treat the two 32-bit halves alike.
static __device__ __forceinline__ void wave_bit_transpose2(uint64_t &a, uint64_t &b, int lane)
{
    unsigned w[4] = {(unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32)};
    {
        const bool up = lane & 32; // lanes below 32 give their high half away and file the partner's low half there; the others mirror it
        const unsigned ga = up ? w[0] : w[1], gb = up ? w[2] : w[3];
        const unsigned ra = (unsigned)__shfl_xor((int)ga, 32, 64), rb = (unsigned)__shfl_xor((int)gb, 32, 64);
        w[0] = up ? ra : w[0]; w[1] = up ? w[1] : ra;
        w[2] = up ? rb : w[2]; w[3] = up ? w[3] : rb;
    }
#pragma unroll
    for (int st = 0; st < 5; ++st) {
        const int j = 16 >> st;
        const unsigned m = j == 16 ? 0x0000FFFFu : j == 8 ? 0x00FF00FFu : j == 4 ? 0x0F0F0F0Fu : j == 2 ? 0x33333333u : 0x55555555u; // positions with bit j clear
        const bool up = lane & j;
        const unsigned rot = up ? (unsigned)j : 32u - (unsigned)j; // rotate right by j: bit p + j comes to p; by 32 - j: bit p - j comes to p
        const unsigned keep = up ? ~m : m;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned t = (unsigned)__shfl_xor((int)w[k], j, 64);
            const unsigned r = __builtin_amdgcn_alignbit(t, t, rot);
            w[k] = (w[k] & keep) | (r & ~keep);
        }
    }
    a = ((uint64_t)w[1] << 32) | w[0];
    b = ((uint64_t)w[3] << 32) | w[2];
}

// per-lane constant weights of the 17 features: with bin = lane + 64 k a lane always sees the same run length
// l = lane % 16 + 1 and six fixed gray levels g_k = lane / 16 + 4 k + 1
struct RlWeights {
    double l, l2, il2, g2[6], ig2[6], wmid[6], wext[6];
    __device__ __forceinline__ void init(int lane)
    {
        l = (double)((lane & 15) + 1);
        l2 = l * l; il2 = 1.0 / l2;
        const double cmid = ((double)RL_LEVELS + 1.0) / 2.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double g = (double)((lane >> 4) + 4 * k + 1);
            g2[k] = g * g;
            ig2[k] = 1.0 / g2[k];
            const double e = (g - cmid) * (g - cmid);
            wext[k] = e;
            wmid[k] = 1.0 / (1.0 + e);
        }
    }
};

// The 17 features of one direction (texture.rs:244-301) from the 384 bins in LDS, by one wave; clears the bins.
// o: the direction's 17 columns; dbg_bins: test hook (nfx_debug_texture_counts), the direction's [level][length] counts
static __device__ __forceinline__ void rl_features(const RlWeights &W, unsigned *s_R, int lane, double Np, float *__restrict__ o,
                                                   uint32_t *__restrict__ dbg_bins)
{
    const double l = W.l, l2 = W.l2, il2 = W.il2;
    const double (&g2)[6] = W.g2, (&ig2)[6] = W.ig2, (&wmid)[6] = W.wmid, (&wext)[6] = W.wext;
    // weighted sums over the bins this lane owns
    double nr = 0, sre = 0, lre = 0, lgre = 0, hgre = 0, srlge = 0, srhge = 0, lrlge = 0, lrhge = 0;
    double srmge = 0, lrmge = 0, srege = 0, lrege = 0, rlm = 0, gln = 0, colsum = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const unsigned bin = s_R[lane + 64 * k];
        const double v = (double)bin;
        s_R[lane + 64 * k] = 0u;
        if (dbg_bins) dbg_bins[lane + 64 * k] = bin; // test hook (nfx_debug_texture_counts): [level][length]
        nr += v;
        const double vs = v * il2, vl = v * l2;
        sre += vs; lre += vl;
        lgre += v * ig2[k]; hgre += v * g2[k];
        srlge += vs * ig2[k]; srhge += vs * g2[k];
        lrlge += vl * ig2[k]; lrhge += vl * g2[k];
        srmge += vs * wmid[k]; lrmge += vl * wmid[k];
        srege += vs * wext[k]; lrege += vl * wext[k];
        rlm += v * l;
        colsum += v;
        // row sum of gray level g_k over the 16 lanes that share lane/16
        const double rs = row16_total_f64(v);
        if ((lane & 15) == 0) gln += rs * rs;
    }
    // column sum of run length l over the 4 lanes that share lane%16
    colsum += __shfl_xor(colsum, 16, 64);
    colsum += __shfl_xor(colsum, 32, 64);
    const double rln = lane < 16 ? colsum * colsum : 0.0;
    // sixteen totals at once (wave_ops.h: reduce-scatter butterfly), in column order (texture.rs:244-301): lane l then
    // holds the total of V[scatter_index<16>(l & 15)], and sixteen lanes do the sixteen divisions side by side
    double V[16] = {sre, lre, gln, rln, lgre, hgre, srlge, srhge, lrlge, lrhge, srmge, lrmge, srege, lrege, nr, rlm};
    const double tot = wave_reduce_scatter(V, lane);
    // scatter_index<16> reverses the four bits: V[14] (nr) sits in lanes 7 + 16k, V[15] (rlm) in lanes 15 + 16k
    const double nr_t = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(tot), 7), __builtin_amdgcn_readlane(__double2loint(tot), 7));
    const double rlm_t = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(tot), 15), __builtin_amdgcn_readlane(__double2loint(tot), 15));
    if (lane < 16) {
                        const int idx = scatter_index<16>(lane);
        const double mean = rlm_t / nr_t;
        const double q = idx == 14 ? nr_t / Np : tot / nr_t; // run_percentage = Nr / Np; everything else over Nr
        o[idx] = (float)q;                                    // idx 15: run_length_mean
        if (idx == 1) o[16] = (float)(q - mean * mean);       // run_length_variance = sum R (l-mean)^2 / Nr = lre/nr - mean^2
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_glrlm(int64_t n, int ps, const T *__restrict__ patches,
                                              const uint64_t *__restrict__ mask_rows, float *__restrict__ out, int F,
                                              int col0, uint32_t *__restrict__ dbg_bins)
{
    constexpr int QS = 68; // byte-map row stride: lane = row reads of one column spread over the banks
    __shared__ uint8_t s_q_all[4][NFX_ROW_WORD_PS * QS];
    __shared__ unsigned s_R_all[4][RL_BINS];
    __shared__ uint64_t s_rows_all[4][64];
    __shared__ __attribute__((aligned(16))) uint64_t s_C_all[4][4 * 64]; // vector path: C_d[r] assembled from 16-bit pieces
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint8_t *s_q = s_q_all[wv];
    unsigned *s_R = s_R_all[wv];
    uint64_t *s_rows = s_rows_all[wv];
    uint64_t *s_C = s_C_all[wv];
    const int P = ps * ps;
    // vector staging: rows are whole 16-byte pieces, every patch starts on a 16-byte boundary
    constexpr int NPC = NFX_ROW_WORD_PS * NFX_ROW_WORD_PS / 16 / 64; // pieces per lane and channel (4)
    const bool vec = sizeof(T) == 1 && (ps & 15) == 0 && (reinterpret_cast<uintptr_t>(patches) & 15) == 0;
    const int npieces = P / 16;
    uint4 pre[3][NPC];
    uint64_t row_next = 0ull;
    auto request = [&](int64_t i) {
        row_next = lane < ps ? mask_rows[i * ps + lane] : 0ull;
        if (vec) {
            const uint4 *src = reinterpret_cast<const uint4 *>(patches + (size_t)i * 3 * P);
#pragma unroll
            for (int k = 0; k < NPC; ++k)
                if (lane + 64 * k < npieces) {
                    pre[0][k] = src[lane + 64 * k];
                    pre[1][k] = src[npieces + lane + 64 * k];
                    pre[2][k] = src[2 * npieces + lane + 64 * k];
                }
        }
    };

    RlWeights Wt;
    Wt.init(lane);
#pragma unroll
    for (int k = 0; k < 6; ++k) s_R[lane + 64 * k] = 0u;
    wave_lds_fence();

    const int64_t istep = (int64_t)gridDim.x * 4;
    if ((int64_t)blockIdx.x * 4 + wv < n) request((int64_t)blockIdx.x * 4 + wv);
#pragma unroll 1
    for (int64_t i = (int64_t)blockIdx.x * 4 + wv; i < n; i += istep) {
        const uint64_t myrow = row_next;
        const uint64_t anyrow = __ballot(myrow != 0ull);
        unsigned lvl[NPC][4]; // vector path: the sixteen levels of each of this lane's pieces
        if (vec) {
            // quantise this lane's pieces and put the levels down (0xFF outside the mask), then ask for the next nucleus
            s_rows[lane] = myrow;
            reinterpret_cast<uint4 *>(s_C)[2 * lane] = make_uint4(0u, 0u, 0u, 0u);
            reinterpret_cast<uint4 *>(s_C)[2 * lane + 1] = make_uint4(0u, 0u, 0u, 0u);
            wave_lds_fence();
#pragma unroll
            for (int k = 0; k < NPC; ++k) {
                const int piece = lane + 64 * k;
                if (piece < npieces) {
                    const int p0 = 16 * piece, r = p0 / ps, c0 = p0 - r * ps;
                    const unsigned mrow = (unsigned)(s_rows[r] >> c0) & 0xFFFFu;
                    const unsigned rr[4] = {pre[0][k].x, pre[0][k].y, pre[0][k].z, pre[0][k].w};
                    const unsigned gg[4] = {pre[1][k].x, pre[1][k].y, pre[1][k].z, pre[1][k].w};
                    const unsigned bb[4] = {pre[2][k].x, pre[2][k].y, pre[2][k].z, pre[2][k].w};
                    unsigned (&packed)[4] = lvl[k];
                    // a band of sixteen rows above or below the nucleus holds no mask pixel in any lane's piece: nothing to quantise
                    if (__ballot(mrow != 0u) == 0ull) {
                        packed[0] = packed[1] = packed[2] = packed[3] = 0xFFFFFFFFu;
                        unsigned *dst0 = reinterpret_cast<unsigned *>(s_q + r * QS + c0);
                        dst0[0] = dst0[1] = dst0[2] = dst0[3] = 0xFFFFFFFFu;
                        continue;
                    }
#pragma unroll
                    for (int w4 = 0; w4 < 4; ++w4) {
                        unsigned word = 0;
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            const int sh = 8 * b;
                            float gs = u8_unit((rr[w4] >> sh) & 255u) + u8_unit((gg[w4] >> sh) & 255u); // utils.rs:172
                            gs = gs + u8_unit((bb[w4] >> sh) & 255u);
                            gs = third_of_u8_sum(gs); // texture.rs:189
                            int qi = (int)floorf(gs * (float)RL_LEVELS);
                            qi = qi < 0 ? 0 : (qi > RL_LEVELS - 1 ? RL_LEVELS - 1 : qi);
                            const unsigned lvl = ((mrow >> (4 * w4 + b)) & 1u) ? (unsigned)qi : 0xFFu;
                            word |= lvl << sh;
                        }
                        packed[w4] = word;
                    }
                    unsigned *dst = reinterpret_cast<unsigned *>(s_q + r * QS + c0); // QS and c0 are multiples of 4
                    dst[0] = packed[0]; dst[1] = packed[1]; dst[2] = packed[2]; dst[3] = packed[3];
                }
            }
            wave_lds_fence();
            // "continues its predecessor" bits of all sixteen pixels of a piece at once (SWAR byte compares against the
            // neighbouring words of the level map), per direction (dx,dy) of texture.rs:176: predecessor left, up-left, up,
            // up-right.  The 16-bit results are filed into C_d[r] through LDS; lane r picks its four words up below.
#pragma unroll
            for (int k = 0; k < NPC; ++k) {
                const int piece = lane + 64 * k;
                if (piece < npieces) {
                    const int p0 = 16 * piece, r = p0 / ps, c0 = p0 - r * ps;
                    const unsigned mrow = (unsigned)(s_rows[r] >> c0) & 0xFFFFu;
                    const unsigned *own = reinterpret_cast<const unsigned *>(s_q + r * QS + c0);
                    const unsigned *up = reinterpret_cast<const unsigned *>(s_q + (r > 0 ? r - 1 : 0) * QS + c0);
                    const bool has_up = r > 0, has_l = c0 > 0, has_r = c0 + 16 < ps;
                    const unsigned none = 0xFFFFFFFFu;
                    const unsigned lw = has_l ? own[-1] : none;
                    const unsigned um = (has_up && has_l) ? up[-1] : none, up4 = (has_up && has_r) ? up[4] : none;
                    const unsigned u0 = has_up ? up[0] : none, u1 = has_up ? up[1] : none, u2 = has_up ? up[2] : none, u3 = has_up ? up[3] : none;
                    const unsigned (&o4)[4] = lvl[k];
                    const unsigned c_l = eq16(o4, __builtin_amdgcn_alignbyte(o4[0], lw, 3), __builtin_amdgcn_alignbyte(o4[1], o4[0], 3),
                                              __builtin_amdgcn_alignbyte(o4[2], o4[1], 3), __builtin_amdgcn_alignbyte(o4[3], o4[2], 3));
                    const unsigned c_ul = eq16(o4, __builtin_amdgcn_alignbyte(u0, um, 3), __builtin_amdgcn_alignbyte(u1, u0, 3),
                                               __builtin_amdgcn_alignbyte(u2, u1, 3), __builtin_amdgcn_alignbyte(u3, u2, 3));
                    const unsigned c_u = eq16(o4, u0, u1, u2, u3);
                    const unsigned c_ur = eq16(o4, __builtin_amdgcn_alignbyte(u1, u0, 1), __builtin_amdgcn_alignbyte(u2, u1, 1),
                                               __builtin_amdgcn_alignbyte(u3, u2, 1), __builtin_amdgcn_alignbyte(up4, u3, 1));
                    unsigned short *cw = reinterpret_cast<unsigned short *>(s_C) + r * 4 + (c0 >> 4);
                    cw[0 * 256] = (unsigned short)(c_l & mrow);  // only masked pixels continue anything (0xFF == 0xFF does not count)
                    cw[1 * 256] = (unsigned short)(c_ul & mrow);
                    cw[2 * 256] = (unsigned short)(c_u & mrow);
                    cw[3 * 256] = (unsigned short)(c_ur & mrow);
                }
            }
            wave_lds_fence();
        }
        if (i + istep < n) request(i + istep);
        const double Np = (double)wave_total_i32(__popcll(myrow)); // pixel_count, texture.rs:196
        const T *pp = patches + (size_t)i * 3 * P;

        // ---- staging: level map + the four "continues" words of every row ---------------------------------
        int clo[4] = {0, 0, 0, 0}, chi[4] = {0, 0, 0, 0}; // C_d[lane] as two 32-bit halves
        if (vec) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const uint64_t cd = s_C[d * 64 + lane];
                clo[d] = (int)(unsigned)cd;
                chi[d] = (int)(unsigned)(cd >> 32);
            }
        }
        unsigned qprev = 0xFFu;                             // the previous row's level in this lane's column
        for (int r = 0; r < (vec ? 0 : ps); ++r) {
            const bool rowhas = (anyrow >> r) & 1ull;
            unsigned q = 0xFFu;
            if (rowhas) {
                const uint64_t w = ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(myrow >> 32), r) << 32) |
                                   (unsigned)__builtin_amdgcn_readlane((int)(unsigned)myrow, r); // r is wave-uniform
                if (vec) {
                    if (lane < ps) q = s_q[r * QS + lane];
                } else if (lane < ps && ((w >> lane) & 1ull)) {
                    const int p = r * ps + lane;
                    float gs = RPx<T>::ld(pp, p) + RPx<T>::ld(pp, P + p);
                    gs = gs + RPx<T>::ld(pp, 2 * P + p);
                    gs = sizeof(T) == 1 ? third_of_u8_sum(gs) : gs / 3.0f;
                    int qi = (int)floorf(gs * (float)RL_LEVELS);
                    qi = qi < 0 ? 0 : (qi > RL_LEVELS - 1 ? RL_LEVELS - 1 : qi);
                    q = (unsigned)qi;
                    s_q[r * QS + lane] = (uint8_t)qi;
                }
                // predecessors: (1,0) left; (1,1) up-left; (0,1) up; (-1,1) up-right      (dx,dy) texture.rs:176
                // neighbours through whole-wave DPP shifts (one VALU op each, no LDS round trip); the lane that has no
                // neighbour keeps the 'outside the mask' level
                const unsigned ql = (unsigned)__builtin_amdgcn_update_dpp(0xFF, (int)q, 0x138, 0xF, 0xF, false);      // wave_shr:1
                const unsigned pul = (unsigned)__builtin_amdgcn_update_dpp(0xFF, (int)qprev, 0x138, 0xF, 0xF, false);
                const unsigned pur = (unsigned)__builtin_amdgcn_update_dpp(0xFF, (int)qprev, 0x130, 0xF, 0xF, false);  // wave_shl:1
                const bool on = q != 0xFFu;
                const uint64_t c0 = __ballot(on && ql == q), c1 = __ballot(on && pul == q);
                const uint64_t c2 = __ballot(on && qprev == q), c3 = __ballot(on && pur == q);
                const bool mine = lane == r; // the ballots are wave-uniform: lane r files them (a v_cndmask per half)
                clo[0] = mine ? (int)(unsigned)c0 : clo[0]; chi[0] = mine ? (int)(unsigned)(c0 >> 32) : chi[0];
                clo[1] = mine ? (int)(unsigned)c1 : clo[1]; chi[1] = mine ? (int)(unsigned)(c1 >> 32) : chi[1];
                clo[2] = mine ? (int)(unsigned)c2 : clo[2]; chi[2] = mine ? (int)(unsigned)(c2 >> 32) : chi[2];
                clo[3] = mine ? (int)(unsigned)c3 : clo[3]; chi[3] = mine ? (int)(unsigned)(c3 >> 32) : chi[3];
            }
            qprev = q;
        }
        wave_lds_fence();

        for (int d = 0; d < 4; ++d) {
            // Lines: the words are re-filed so that a lane holds one LINE of the direction (a row, a column, a diagonal modulo 64) with
            // the run direction along its bits -- a skew (row r rotated by -r / +r bits: the two diagonals become columns; where two
            // diagonals share a column the second one's first pixel has no predecessor, so its C bit is clear and no run crosses)
            // and a 64 x 64 bit transpose.  Pixel of (line l, bit p): d = 0: (row l, column p); else row p, column (l + sgn p) mod 64.
            const int sgn = d == 1 ? 1 : (d == 3 ? -1 : 0);
            int ln = lane;
            asm volatile("" : "+v"(ln)); // opaque per direction: the butterflies' per-lane masks, amounts and partner addresses are cheap to
                                         // make and would otherwise be hoisted out of both loops into twenty registers the kernel does not have
            uint64_t C = ((uint64_t)(unsigned)chi[d] << 32) | (uint64_t)(unsigned)clo[d];
            uint64_t S = myrow & ~C;                // run starts of this lane's row
            if (d != 0) {
                if (sgn) {
                    const unsigned rot = (unsigned)(sgn > 0 ? ln : 64 - ln) & 63u; // rotate right by rot: bit c -> bit (c - rot) mod 64
                    C = rotr64(C, rot);
                    S = rotr64(S, rot);
                }
                wave_bit_transpose2(C, S, ln);
            }
            // a start at bit p has the length 1 + (number of set C bits right above p), at most 16; one bin update per run, the
            // line's starts walked as two 32-bit words (single-instruction find-first-set, funnel shift, find-first-clear)
            const uint64_t C1 = C >> 1;
            const unsigned c1lo = (unsigned)C1, c1hi = (unsigned)(C1 >> 32);
            const int pa = d == 0 ? 1 : QS, la = d == 0 ? QS : 1; // byte-map strides of the bit position and of the line index
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                unsigned rem = half ? (unsigned)(S >> 32) : (unsigned)S;
                const unsigned whi = half ? 0u : c1hi, wlo = half ? c1hi : c1lo;
                while (rem) {
                    const int p = __ffs((int)rem) - 1;
                    rem &= rem - 1u;
                    const unsigned above = __builtin_amdgcn_alignbit(whi, wlo, (unsigned)p); // C bits p + 1, p + 2, ...
                    unsigned len1 = (unsigned)(__ffs((int)~above) - 1);                       // all set: 0xFFFFFFFF -> clamped below
                    len1 = len1 < (unsigned)RL_MAXLEN - 1u ? len1 : (unsigned)RL_MAXLEN - 1u;
                    const int pos = 32 * half + p;
                    const unsigned q = s_q[pos * pa + ((ln + sgn * pos) & 63) * la];
                    // starts lie inside the mask, so q <= 23; the clamp keeps a stray byte (0xFF outside the mask) from ever
                    // indexing past the 384 bins into a neighbouring wave's table
                    atomicAdd(&s_R[(q < (unsigned)RL_LEVELS ? q : (unsigned)RL_LEVELS - 1u) * RL_MAXLEN + len1], 1u);
                }
            }
            wave_lds_fence();
            rl_features(Wt, s_R, lane, Np, out + i * (int64_t)F + col0 + 17 * d, dbg_bins ? dbg_bins + (i * 4 + d) * RL_BINS : nullptr);
            wave_lds_fence();
        }
    }
}

// ---- patch sizes 65..512: one 256-thread workgroup per nucleus (see k_geometry.hip, "patch sizes 65..128" and "129..512") ----
// The level map (0xFF outside the mask) is kept as bytes -- in LDS up to 128 x 128, in the workgroup's slice of a scratch buffer in
// device memory above (GLOBAL_MAP: 256 KB at 512 x 512; written and read by this workgroup only, with workgroup barriers between);
// every thread walks its pixels, and a pixel that starts a run (its predecessor along the direction is outside the mask or of
// another level) follows the run for at most 16 steps and bumps one of the 384 bins; wave 0 turns the bins into the 17 features
// with the same code as the wave kernel.
template <typename T, bool GLOBAL_MAP>
__global__ __launch_bounds__(256) void k_glrlm_wide(int64_t n, int ps, const T *__restrict__ patches, const uint64_t *__restrict__ mask_rows,
                                                    float *__restrict__ out, int F, int col0, uint32_t *__restrict__ dbg_bins, uint8_t *__restrict__ map_scratch)
{
    __shared__ uint8_t s_q_lds[GLOBAL_MAP ? 16 : NFX_WIDE_PS * NFX_WIDE_PS];
    __shared__ unsigned s_R[RL_BINS];
    __shared__ int s_np;
    const int t = threadIdx.x, lane = t & 63;
    const int P = ps * ps, W = (ps + 63) >> 6;
    uint8_t *const s_q = GLOBAL_MAP ? map_scratch + (size_t)blockIdx.x * P : s_q_lds;
    RlWeights Wt;
    Wt.init(lane);
    for (int k = t; k < RL_BINS; k += 256) s_R[k] = 0u;
    for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
        if (t == 0) s_np = 0;
        __syncthreads(); // (also: every thread is done with the previous nucleus's map)
        const T *pp = patches + (size_t)i * 3 * P;
        const uint64_t *rows = mask_rows + (size_t)i * ps * W;
        int np = 0;
        for (int p = t; p < P; p += 256) {
            const int r = p / ps, c = p - r * ps;
            unsigned q = 0xFFu;
            if ((rows[W * r + (c >> 6)] >> (c & 63)) & 1ull) {
                float gs = RPx<T>::ld(pp, p) + RPx<T>::ld(pp, P + p);
                gs = gs + RPx<T>::ld(pp, 2 * P + p);
                gs = sizeof(T) == 1 ? third_of_u8_sum(gs) : gs / 3.0f; // texture.rs:189
                int qi = (int)floorf(gs * (float)RL_LEVELS);
                qi = qi < 0 ? 0 : (qi > RL_LEVELS - 1 ? RL_LEVELS - 1 : qi);
                q = (unsigned)qi;
                ++np;
            }
            s_q[p] = (uint8_t)q;
        }
        np = wave_total_i32(np);
        if (lane == 0 && np) atomicAdd(&s_np, np);
        __syncthreads();
        const double Np = (double)s_np; // pixel_count, texture.rs:196
        for (int d = 0; d < 4; ++d) {
            const int dx = kRlDirs[d][0], dy = kRlDirs[d][1];
            for (int p = t; p < P; p += 256) {
                const unsigned q = s_q[p];
                if (q == 0xFFu) continue;
                const int r = p / ps, c = p - r * ps;
                const int rp = r - dy, cp = c - dx;
                if (rp >= 0 && rp < ps && cp >= 0 && cp < ps && s_q[rp * ps + cp] == q) continue; // not a run start
                int len = 1, rn = r + dy, cn = c + dx;
                while (len < RL_MAXLEN && rn >= 0 && rn < ps && cn >= 0 && cn < ps && s_q[rn * ps + cn] == q) { ++len; rn += dy; cn += dx; }
                atomicAdd(&s_R[q * RL_MAXLEN + (len - 1)], 1u); // runs longer than 16 share the last bin
            }
            __syncthreads();
            if (t < 64) rl_features(Wt, s_R, lane, Np, out + i * (int64_t)F + col0 + 17 * d, dbg_bins ? dbg_bins + (i * 4 + d) * RL_BINS : nullptr);
            __syncthreads();
        }
    }
}

static constexpr int RL_HUGE_GRID = 512; // workgroups of the launch over patches above 128 x 128: one map slice each
size_t glrlm_scratch_bytes(const Job &job)
{
    if (job.ps <= NFX_WIDE_PS) return 0;
    const size_t g = (size_t)(job.n < RL_HUGE_GRID ? (job.n > 0 ? job.n : 1) : RL_HUGE_GRID);
    return g * (size_t)job.ps * job.ps + 256;
}

hipError_t launch_glrlm(nfx_ctx *ctx, const Job &job, int col0, void *scratch)
{
    if (job.n == 0) return hipSuccess;
    ScopedTiming t(ctx, KS_GLRLM);
    if (job.ps > NFX_WIDE_PS) {
        if (!scratch) return hipErrorInvalidValue;
        const unsigned hgrid = (unsigned)(job.n < RL_HUGE_GRID ? job.n : RL_HUGE_GRID);
        uint8_t *map = reinterpret_cast<uint8_t *>(scratch);
        if (job.patches_f32)
            hipLaunchKernelGGL((k_glrlm_wide<float, true>), dim3(hgrid), dim3(256), 0, ctx->stream, job.n, job.ps, reinterpret_cast<const float *>(job.patches),
                               job.mask_rows, job.out, job.F, col0, job.dbg_glrlm, map);
        else
            hipLaunchKernelGGL((k_glrlm_wide<uint8_t, true>), dim3(hgrid), dim3(256), 0, ctx->stream, job.n, job.ps, reinterpret_cast<const uint8_t *>(job.patches),
                               job.mask_rows, job.out, job.F, col0, job.dbg_glrlm, map);
        return hipGetLastError();
    }
    if (job.ps > NFX_ROW_WORD_PS) {
        const unsigned wgrid = (unsigned)(job.n < 256 * 8 ? job.n : 256 * 8);
        if (job.patches_f32)
            hipLaunchKernelGGL((k_glrlm_wide<float, false>), dim3(wgrid), dim3(256), 0, ctx->stream, job.n, job.ps, reinterpret_cast<const float *>(job.patches),
                               job.mask_rows, job.out, job.F, col0, job.dbg_glrlm, (uint8_t *)nullptr);
        else
            hipLaunchKernelGGL((k_glrlm_wide<uint8_t, false>), dim3(wgrid), dim3(256), 0, ctx->stream, job.n, job.ps, reinterpret_cast<const uint8_t *>(job.patches),
                               job.mask_rows, job.out, job.F, col0, job.dbg_glrlm, (uint8_t *)nullptr);
        return hipGetLastError();
    }
    const int64_t maxb = 256 * 7 * 2, need = (job.n + 3) / 4; // 7 workgroups (28 waves) fit a CU's LDS
    const unsigned grid = (unsigned)(need < maxb ? need : maxb);
    if (job.patches_f32)
        hipLaunchKernelGGL(k_glrlm<float>, dim3(grid), dim3(256), 0, ctx->stream, job.n, job.ps,
                           reinterpret_cast<const float *>(job.patches), job.mask_rows, job.out, job.F, col0, job.dbg_glrlm);
    else
        hipLaunchKernelGGL(k_glrlm<uint8_t>, dim3(grid), dim3(256), 0, ctx->stream, job.n, job.ps,
                           reinterpret_cast<const uint8_t *>(job.patches), job.mask_rows, job.out, job.F, col0, job.dbg_glrlm);
    return hipGetLastError();
}

}  // namespace nfx
