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
//   * ONE WAVEFRONT PER NUCLEUS (single-wave workgroups, persistent): no cross-wave barrier or reduction.
//     k_glcm_classify sorts the nuclei into six lists by mask size (= list and hash capacity); each list has its own
//     instantiation, so a wave knows its next nucleus and requests that nucleus's whole u8 patch (16-byte pieces,
//     in registers) and mask row while it works on the current one: no HBM latency on the critical path;
//   * the mask's pixels are compacted into an LDS list, one word per pixel: r << 24 | c << 16 | level128 << 8 |
//     level254.  Levels 64 and 32 are level128 >> 1 and >> 2 (floor(128 g) >> 1 == floor(64 g) exactly), and the
//     patch is quantised with exact 3-instruction divisions by 255 and 3 (exact_div.h);
//   * co-occurrence cells are counted with RETURNING LDS atomics.  The value the atomic returns is the
//     pair's rank k inside its cell, and
//         sum_cells G^2     = sum_pairs 2(2k+1)            (4(2k+1) on the diagonal)
//         sum_cells G ln G  = sum_pairs 2[h(k+1)-h(k)],  h(x) = x ln x   (diag: H(x) = 2x ln 2x)
//     telescope over the ranks (difference tables for every possible rank live in global memory, L1/L2
//     resident; the look-up for a pair is issued one loop iteration after its rank is known), so the
//     multiplicities are never read back and no cell needs an owner;
//   * gray levels are counted RELATIVE to the nucleus's lowest level (GlGeom): outside the top class the dense tables and the
//     histograms cover 3/4 of the scale, and a nucleus whose levels span more is handed to the top class before any counting;
//   * L=32 and L=64 use dense triangular tables of packed 16-bit counters (5 KB; 3 KB for the 3/4 span); L=128 and L=254 use
//     open-addressing HASH tables -- 3/4 slot per mask pixel of the class's capacity for the typical texture, with a
//     hand-over to the next class for the nucleus that fills it (template parameter FULL) -- instead
//     of 16.5 / 64.8 KB dense ones, keyed by hi << 8 | lo and filled insert-first (a new cell costs one LDS
//     round trip): a wave's whole state is 16 KB, ten nuclei in flight per CU.  Tables are reset by sweeping
//     them with 16-byte stores;
//   * levels 32/64/128 share one pass over the pairs of an offset and leave a one-byte memo per pixel (list
//     distance to the partner, backward-neighbour flag); the level-254 pass in the same table memory re-finds
//     its pairs from the memo without touching the mask;
//   * every lane walks TWO list entries per iteration through a branch-free body (two independent LDS chains);
//   * the marginal histogram p_x is filled per PIXEL (degree = number of pairs the pixel is in) and only at the
//     finest nested level; sum s n_s and sum s^2 n_s come out of the sum histogram when it is folded; the
//     remaining power sums (d, d^2 with d=|a-b|), 1/(1+d^2) and the rank terms are per pair.
// The ten sums of every (level, offset) combination are parked in a 1280-byte slot the wave owns in global memory (one
// slot per resident wave: 14 MB in all, rewritten for every nucleus, so it lives in the XCD's L2 and never travels to
// HBM); after the fourth offset sixteen lanes read them back and evaluate the 14 features of their combination.
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "nfx_internal.h"
#include "exact_div.h"
#include "wave_ops.h"

namespace nfx {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// NFX_GL_DERIVE (round 5; 0 = the round 1-4 form, kept for A/B builds): the pass over levels 32 / 64 / 128 counts level 128 only, pair by pair; levels 64
// and 32 nest in it (floor(128 g) >> k), so their dense tables, sum and difference histograms are filled afterwards from the DISTINCT level-128 cells (a
// sweep over the hash slots: count-weighted plain adds instead of four returning atomics per pair on the addresses with the highest multiplicity) and
// their sum G^2 / sum G ln G from a sweep over the final dense counts.  Every sum the slots' order could touch is an integer, so a nucleus still gets the
// same bits in whichever class it runs.  -7.0 % at 48 x 48, -4.3 % at 64 x 64 (profiles/r05_glcm_derive.txt); counts == the oracle's as before.
#ifndef NFX_GL_DERIVE
#define NFX_GL_DERIVE 1
#endif
static constexpr int GL_NACC = 10;
static constexpr int GL_RANKS = 4096;      // rank range of the global difference tables (a cell holds at most 4032 pairs)
static constexpr int GL_COUNTS = 8200;     // count range of the x ln x table (a marginal bin holds at most 2 * 4032)
static constexpr unsigned GL_EMPTY = 0xFFFFFFFFu;
// Table and histogram geometry of one instantiation.  FULL (the top class of a patch size, a small call's single launch): every
// level can occur.  Otherwise the structures are sized for the TYPICAL nucleus -- a nucleus's gray levels span 3/4 of the scale at
// most (R128 = 96 of 128 levels, R254 = 192 of 254; on the synthetic H&E nuclei the spans are 76 / 150 at the median, 90 / 182 at
// the maximum): levels are taken relative to the nucleus's lowest one, the dense tables of levels 32 / 64 and the histograms cover
// the span only -- 3 KB of LDS per wave less, i.e. one or two more waves per CU in every class -- and a nucleus whose span
// does not fit is handed to the top class's list before any counting is done.
template <bool FULL>
struct GlGeom {
    static constexpr int R128 = FULL ? 128 : 96, R64 = R128 / 2, R32 = R128 / 4;     // level spans the dense tables hold
    static constexpr int R254 = FULL ? 254 : 192;
    static constexpr int D32_WORDS = ((R32 * (R32 + 1) / 2 + 1) / 2 + 7) & ~7;       // packed 16-bit triangular cells, 32-byte pieces
    static constexpr int D64_WORDS = ((R64 * (R64 + 1) / 2 + 1) / 2 + 7) & ~7;
    static constexpr int D64_OFF = D32_WORDS;                                         // pass A: d32 | d64 | h128 (hash)
    static constexpr int H128_OFF = D32_WORDS + D64_WORDS;                             // pass B: h254 over the whole table memory
    // histogram words, every segment a whole number of 64-word slices (lane + 64 m addressing in the folds):
    //   pass A: hx128 at 0 (128) | hs32 at 128 (64) | hs64 at 192 (128) | hs128 at 320 (64 HS128_SLICES)
    //   pass B: hx254 at 0 (64 HXB) | hs254 at 64 HXB (64 HSB)
    static constexpr int HS128_SLICES = FULL ? 4 : 3;
    static constexpr int HXB = FULL ? 4 : 3, HSB = 2 * HXB;
    // (NFX_GL_DERIVE) one more slice behind hs128: the DIFFERENCE histograms of levels 64 (low half of word d) and 32 (high half): pairs with |a - b| = d
    static constexpr int PD_OFF = 320 + 64 * HS128_SLICES;
    static constexpr int HIST_A = 320 + 64 * HS128_SLICES + (NFX_GL_DERIVE ? 64 : 0), HIST_B = 64 * (HXB + HSB);
    static constexpr int HIST_WORDS = HIST_A > HIST_B ? HIST_A : HIST_B;
    static_assert(2 * R32 <= 64 && 2 * R64 <= 128 && 2 * R128 <= 64 * HS128_SLICES && R254 <= 64 * HXB && 2 * R254 <= 64 * HSB, "histogram segments");
};

__constant__ int kGlcmOffsets[4][2] = {{0, 1}, {1, 1}, {1, 0}, {1, -1}}; // texture.rs:20 (dy,dx)

static constexpr int GL_ROWEND = 1048;     // words of the largest dense triangular table (64 levels: 2080 packed 16-bit cells), rounded up
struct GlcmLuts {           // device tables built on the host
    double dO[GL_RANKS];    // 2[h(k+1)-h(k)]
    double dD[GL_RANKS];    // H(k+1)-H(k)
    double xlnx[GL_COUNTS]; // x ln x
    unsigned short rowend[GL_ROWEND]; // (NFX_GL_DERIVE) word w of a dense triangular table holds cells 2w, 2w + 1; rowend[w] = index of the DIAGONAL cell of the row cell 2w lies in
};

// table look-ups by a 32-bit BYTE offset from a wave-uniform base: the address is one shift and the scalar-base form of the load
// (global_load_dwordx2 v, v_offset, s[base]) instead of a zero extension and a 64-bit shift-add per look-up (-0.5 % / -0.7 % at 48 / 64)
static __device__ __forceinline__ double lut_at(const double *__restrict__ tab, unsigned idx)
{
    return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(tab) + (size_t)(idx << 3)); // idx < 2^17
}

struct LevelAcc {
    unsigned S1, S2, D1, D2, asmn;
    float idm;
    double ents;
};

static __device__ __forceinline__ unsigned quant(float gray, int L)
{
    int q = (int)floorf(gray * (float)L);
    q = q < 0 ? 0 : (q > L - 1 ? L - 1 : q);
    return (unsigned)q;
}

template <typename T>
static __device__ __forceinline__ float gray_at(const T *pp, int P, int p)
{
    float gs;
    if (sizeof(T) == 1) {
        const uint8_t *pb = reinterpret_cast<const uint8_t *>(pp);
        gs = (float)pb[p] / 255.0f + (float)pb[P + p] / 255.0f; // utils.rs:172
        gs = gs + (float)pb[2 * P + p] / 255.0f;
    } else {
        const float *pf = reinterpret_cast<const float *>(pp);
        gs = pf[p] + pf[P + p];
        gs = gs + pf[2 * P + p];
    }
    return gs / 3.0f; // texture.rs:36
}

// hash key of the unordered cell {a, b}: at most 0xFDFD, below the dummy (0xFFFE) and empty (0xFFFF) keys
static __device__ __forceinline__ unsigned cell_key(unsigned a, unsigned b)
{
    const unsigned lo = a < b ? a : b, hi = a < b ? b : a;
    return (hi << 8) | lo;
}

static __device__ __forceinline__ unsigned tri_index(unsigned a, unsigned b)
{
    const unsigned lo = a < b ? a : b, hi = a < b ? b : a;
    return hi * (hi + 1u) / 2u + lo;
}

// open-addressing hash tables hold one word per cell: key << 16 | count.

// ---- two list entries per lane and iteration -------------------------------------------------------------------
// The pair loops are chains of dependent LDS round trips (position -> mask rows -> partner level -> cell counter ->
// next probe) and LDS capacity caps a CU at ~10 waves, so the latency has to be hidden inside the wave: every lane
// walks TWO list entries per iteration through one branch-free body, giving the scheduler two independent chains.
// "Branch-free": an entry without a partner adds 0 to valid counters and aims its compare-and-swap at the lane's
// private dummy word (key 0xFFFE, never empty, never equal), so no exec-mask region splits the basic block.
struct Located {
    unsigned qa, qb; // packed levels of the pixel and of its partner
    bool fwd;        // the partner (r+dy, c+dx) is in the mask: this entry owns a pair
    unsigned deg;    // number of pairs the pixel takes part in (forward + backward)
    unsigned memo;   // what pass B needs to find the same partner again: list distance (1..126, 0 = none) | bwd << 7
};

static __device__ __forceinline__ Located locate(int e, bool valid, int ps, int dy, int dx, const unsigned *s_q,
                                                 const uint64_t *s_rows1, const unsigned short *s_rowstart)
{
    const int es = valid ? e : 0;
    Located L;
    L.qa = s_q[es];
    const int r = (int)(L.qa >> 24), c = (int)((L.qa >> 16) & 255u);
    const int c2 = c + dx, cb = c - dx;
    const uint64_t w2 = s_rows1[r + 1 + dy], wb = s_rows1[r + 1 - dy];
    const unsigned rs = s_rowstart[r + dy];
    L.fwd = valid && (unsigned)c2 < (unsigned)ps && ((w2 >> (c2 & 63)) & 1ull);
    const bool bwd = valid && (unsigned)cb < (unsigned)ps && ((wb >> (cb & 63)) & 1ull);
    L.deg = (L.fwd ? 1u : 0u) + (bwd ? 1u : 0u);
    // without a partner this is still a list index (<= K): the value read is masked into range and adds 0 everywhere
    const unsigned pidx = rs + (unsigned)__popcll(w2 & ((1ull << (c2 & 63)) - 1ull));
    L.qb = s_q[pidx];
    L.memo = (L.fwd ? pidx - (unsigned)es : 0u) | (bwd ? 128u : 0u); // the partner sits in this or the next row: < 2 ps entries on
    return L;
}

// pass B: the same pixel and partner through the memo pass A left behind (no mask arithmetic)
static __device__ __forceinline__ Located relocate(int e, bool valid, const unsigned *s_q, const unsigned char *s_memo)
{
    const int es = valid ? e : 0;
    const unsigned m = valid ? s_memo[es] : 0u;
    Located L;
    L.qa = s_q[es];
    const unsigned delta = m & 127u;
    L.fwd = delta != 0u;
    L.deg = (L.fwd ? 1u : 0u) + (m >> 7);
    L.qb = s_q[es + (int)delta];
    L.memo = m;
    return L;
}

// branch-free accumulation of one pair at one level; returns the rank-table index for the deferred look-up
static __device__ __forceinline__ unsigned pair_acc(bool fwd, unsigned a, unsigned b, unsigned rk, LevelAcc &A)
{
    const unsigned lo = a < b ? a : b, hi = a < b ? b : a;
    const unsigned d = fwd ? hi - lo : 0u; // the sums over s = a + b come out of the sum histogram when it is folded
    A.D1 += d; A.D2 += d * d;
    A.idm += fwd ? __builtin_amdgcn_rcpf(1.0f + (float)(d * d)) : 0.0f; // 1/(1+d^2), 1 ulp
    const bool diag = lo == hi;
    A.asmn += fwd ? (diag ? 4u : 2u) * (2u * rk + 1u) : 0u;
    return fwd ? (diag ? (unsigned)GL_RANKS : 0u) + rk : 0u;
}

// ---- the two list entries of an iteration side by side in the halves of one register ------------------------------
// Levels are at most 253, triangular indices of the dense tables at most 2079 and cell keys at most 0xFDFD: the
// ordering, the differences, the table indices and the hash keys of BOTH entries are packed 16-bit instructions, and the
// sums of |a-b| and (a-b)^2 take one v_dot2_u32_u16 each for the pair.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ u16x2 as_u16x2(unsigned v) { return __builtin_bit_cast(u16x2, v); }
static __device__ __forceinline__ unsigned as_u32(u16x2 v) { return __builtin_bit_cast(unsigned, v); }
struct Ordered2 { u16x2 lo, hi, dd; }; // min, max and difference of the entries' two levels (.x: first entry, .y: second)
static __device__ __forceinline__ Ordered2 order2(u16x2 a, u16x2 b)
{
    Ordered2 p;
    p.lo = __builtin_elementwise_min(a, b);
    p.hi = __builtin_elementwise_max(a, b);
    p.dd = p.hi - p.lo;
    return p;
}
static __device__ __forceinline__ u16x2 tri_index2(const Ordered2 &p) { return ((p.hi * (p.hi + (u16x2){1, 1})) >> (u16x2){1, 1}) + p.lo; }
static __device__ __forceinline__ u16x2 cell_key2(const Ordered2 &p) { return (p.hi << (u16x2){8, 8}) | p.lo; }
// byte `B` of x in the low half, byte `B` of y in the high half (one v_perm_b32)
template <int B>
static __device__ __forceinline__ u16x2 bytes2(unsigned x, unsigned y)
{
    return as_u16x2(__builtin_amdgcn_perm(y, x, 0x0c000c00u | (unsigned)B | ((unsigned)(4 + B) << 16)));
}

// pair_acc for both entries at one level; fmask = 0xFFFF in the half of an entry that owns a pair
static __device__ __forceinline__ void pair_acc2(bool fx, bool fy, unsigned fmask, const Ordered2 &p, unsigned rkx, unsigned rky, LevelAcc &A,
                                                 unsigned &idx_x, unsigned &idx_y)
{
    const u16x2 d = as_u16x2(as_u32(p.dd) & fmask); // the sums over s = a + b come out of the sum histogram when it is folded
    A.D1 = __builtin_amdgcn_udot2(d, (u16x2){1, 1}, A.D1, false);
    A.D2 = __builtin_amdgcn_udot2(d, d, A.D2, false);
    const u16x2 d2 = d * d;                                             // <= 253^2 = 64 009
    A.idm += fx ? __builtin_amdgcn_rcpf(1.0f + (float)d2.x) : 0.0f;     // 1/(1+d^2), 1 ulp; first entry first, as pair_acc
    A.idm += fy ? __builtin_amdgcn_rcpf(1.0f + (float)d2.y) : 0.0f;
    const bool dgx = p.dd.x == 0, dgy = p.dd.y == 0;
    A.asmn += fx ? (dgx ? 4u : 2u) * (2u * rkx + 1u) : 0u;
    A.asmn += fy ? (dgy ? 4u : 2u) * (2u * rky + 1u) : 0u;
    idx_x = fx ? (dgx ? (unsigned)GL_RANKS : 0u) + rkx : 0u;
    idx_y = fy ? (dgy ? (unsigned)GL_RANKS : 0u) + rky : 0u;
}

static __device__ __forceinline__ unsigned dense_rank_if(unsigned *tab, unsigned idx, bool fwd)
{
    const unsigned sh = (idx & 1u) * 16u;
    return (atomicAdd(&tab[idx >> 1], fwd ? 1u << sh : 0u) >> sh) & 0xFFFFu;
}

static constexpr unsigned GL_DUMMY = 0xFFFE0000u;

// ranks of two keys in the open-addressing table; both probes of a round are issued back to back
// SLOTS is any table size up to 2^16: sixteen hash bits are scaled into [0, SLOTS)
// one probe step of one search: claim the slot if it is empty, count in it if it holds the key, else move on
template <int SLOTS>
static __device__ __forceinline__ void probe_step(unsigned *tab, unsigned *dummy, unsigned key, unsigned &h, bool &act, unsigned &rk)
{
    unsigned *const slot = act ? tab + h : dummy;
    const unsigned prev = atomicCAS(slot, GL_EMPTY, (key << 16) | 1u);
    const bool same = act && (prev >> 16) == key;
    const unsigned r = atomicAdd(same ? slot : dummy, same ? 1u : 0u);
    if (same) rk = r & 0xFFFFu;
    act = act && prev != GL_EMPTY && !same;
    h = h + 1u == (unsigned)SLOTS ? 0u : h + 1u;
}

template <int SLOTS>
static __device__ __forceinline__ void hash_rank2(unsigned *tab, unsigned *dummy, unsigned key0, bool act0, unsigned key1, bool act1,
                                                  unsigned &rk0, unsigned &rk1, bool &full)
{
    static_assert(SLOTS > 0 && SLOTS <= 65536, "table size");
    // HIP's __umul24 returns int: shift the product as unsigned, or half of the keys start at a negative slot
    unsigned h0 = (unsigned)__umul24(key0, 0x9E3779u) >> 16, h1 = (unsigned)__umul24(key1, 0x9E3779u) >> 16;
    h0 = (unsigned)__umul24(h0, (unsigned)SLOTS) >> 16;
    h1 = (unsigned)__umul24(h1, (unsigned)SLOTS) >> 16;
    rk0 = 0u; rk1 = 0u;
    // first probe of both searches, back to back: it settles almost nine searches in ten
    {
        unsigned *const slot0 = act0 ? tab + h0 : dummy, *const slot1 = act1 ? tab + h1 : dummy;
        const unsigned prev0 = atomicCAS(slot0, GL_EMPTY, (key0 << 16) | 1u);
        const unsigned prev1 = atomicCAS(slot1, GL_EMPTY, (key1 << 16) | 1u);
        const bool same0 = act0 && (prev0 >> 16) == key0, same1 = act1 && (prev1 >> 16) == key1;
        const unsigned r0 = atomicAdd(same0 ? slot0 : dummy, same0 ? 1u : 0u);
        const unsigned r1 = atomicAdd(same1 ? slot1 : dummy, same1 ? 1u : 0u);
        if (same0) rk0 = r0 & 0xFFFFu;
        if (same1) rk1 = r1 & 0xFFFFu;
        act0 = act0 && prev0 != GL_EMPTY && !same0;
        act1 = act1 && prev1 != GL_EMPTY && !same1;
        h0 = h0 + 1u == (unsigned)SLOTS ? 0u : h0 + 1u;
        h1 = h1 + 1u == (unsigned)SLOTS ? 0u : h1 + 1u;
    }
    // what is left is thin (a few lanes, and hardly ever both searches of a lane): the lane's unsettled search -- the
    // second one if the first is done -- continues alone, at half the LDS operations per round
    if (__any(act0 || act1)) {
        const bool second = !act0 && act1;
        unsigned key = second ? key1 : key0, h = second ? h1 : h0, rk = 0u;
        bool act = act0 || act1;
        const bool both = act0 && act1;
        int rounds = 0; // a full sweep of the table always ends a search; the bound only guards against a corrupted table
        while (act && ++rounds <= SLOTS) probe_step<SLOTS>(tab, dummy, key, h, act, rk);
        full = full || act; // a whole sweep without a free slot or the key: the table is full (typical-size tables, see k_glcm)
        // (`rk` stays 0 for a search that ended by claiming an empty slot)
        if (second) rk1 = (act1 ? rk : rk1); else rk0 = (act0 ? rk : rk0);
        if (__any(both)) { // both searches of a lane were unsettled: now its second one
            act = both; key = key1; h = h1; rk = 0u; rounds = 0;
            while (act && ++rounds <= SLOTS) probe_step<SLOTS>(tab, dummy, key, h, act, rk);
            full = full || act;
            if (both) rk1 = rk;
        }
    }
}

static __device__ __forceinline__ void sweep(unsigned *tab, int words, unsigned value, int lane)
{
    uint4 *t4 = reinterpret_cast<uint4 *>(tab);
    const uint4 v = make_uint4(value, value, value, value);
    for (int k = lane; k < words / 4; k += 64) t4[k] = v;
}

// ---- mask-size classes ---------------------------------------------------------------------------------------
// One list of nucleus indices per hash capacity, filled by k_glcm_classify (order inside a list is irrelevant).
// A wave's LDS footprint grows with its class's capacity and LDS is what limits the waves per CU (11 / 9 / 8 / 6 / 5 / 3 for
// the six capacities at 48 x 48), so the steps are fine where the nuclei are: half-steps between the powers of two.
static constexpr int GL_CLASSES = 6;
static constexpr int GL_CHUNK = 2; // list entries a wave takes per visit to its class's cursor
static __device__ __host__ __forceinline__ int glcm_class(int K) { return K <= 512 ? 0 : K <= 768 ? 1 : K <= 1024 ? 2 : K <= 1536 ? 3 : K <= 2048 ? 4 : 5; }

__global__ __launch_bounds__(256) void k_glcm_classify(int64_t n, int ps, const uint64_t *__restrict__ mask_rows, const int *__restrict__ areas,
                                                       int *__restrict__ lists, int *__restrict__ counts, int force_class)
{
    // sixteen lanes per nucleus, four nuclei per lane group: 64 nuclei per block.  List slots are handed out inside the
    // block (LDS), so the four global counters see one atomic per block and class, not one per nucleus.
    constexpr int PER_GROUP = 4;
    __shared__ int s_cnt[GL_CLASSES], s_base[GL_CLASSES];
    if (threadIdx.x < GL_CLASSES) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int group = threadIdx.x >> 4, sub = threadIdx.x & 15;
    int cls[PER_GROUP], rank[PER_GROUP];
#pragma unroll
    for (int u = 0; u < PER_GROUP; ++u) {
        const int64_t i = ((int64_t)blockIdx.x * 16 + group) * PER_GROUP + u;
        int K = 0;
        if (areas) {
            if (i < n) K = areas[i]; // the rasteriser's count (all sixteen lanes read the same word)
        } else {
            if (i < n)
                for (int r = sub; r < ps; r += 16) K += __popcll(mask_rows[i * ps + r]);
            K += __shfl_xor(K, 1, 64); K += __shfl_xor(K, 2, 64); K += __shfl_xor(K, 4, 64); K += __shfl_xor(K, 8, 64);
        }
        cls[u] = force_class >= 0 ? force_class : glcm_class(K); // a small call runs as ONE launch of the class every mask fits
        rank[u] = (i < n && sub == 0) ? atomicAdd(&s_cnt[cls[u]], 1) : 0;
    }
    __syncthreads();
    if (threadIdx.x < GL_CLASSES) s_base[threadIdx.x] = s_cnt[threadIdx.x] ? atomicAdd(&counts[threadIdx.x], s_cnt[threadIdx.x]) : 0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PER_GROUP; ++u) {
        const int64_t i = ((int64_t)blockIdx.x * 16 + group) * PER_GROUP + u;
        if (i < n && sub == 0) lists[(int64_t)cls[u] * n + s_base[cls[u]] + rank[u]] = (int)i;
    }
}

// The 14 Haralick features of one (level, offset) combination from its ten sums; one lane per combination.
// dbg (test hook, nfx_debug_texture_counts): the six exact integers among the sums.
// WIDE: patches above 64 x 64 hold more pairs than the x ln x table covers; that (register-rich) kernel takes plain logarithms.
// base: the level the nucleus's levels were counted relative to (0 for the full-range structures): only the sums over s = a + b see it.
template <bool WIDE>
static __device__ void glcm_features(const double (&a)[GL_NACC], double base, const GlcmLuts *__restrict__ lut, float *__restrict__ o, int64_t *__restrict__ dbg)
{
    const double N = a[0], D1 = a[3], D2 = a[4], IDM = a[5], ASMN = a[6], ENTS = a[7], HXS = a[8], HSS = a[9];
    // s_abs = s_rel + 2 base for each of the N pairs: exact in doubles (integers far below 2^53)
    const double S1 = a[1] + 2.0 * base * N, S2 = a[2] + 4.0 * base * a[1] + 4.0 * base * base * N;
    if (dbg) {
        dbg[0] = (int64_t)N; dbg[1] = (int64_t)S1; dbg[2] = (int64_t)S2; dbg[3] = (int64_t)D1; dbg[4] = (int64_t)D2; dbg[5] = (int64_t)ASMN;
    }
    if (N == 0.0) {
        const float qnan = __builtin_nanf("");
#pragma unroll
        for (int k = 0; k < 14; ++k) o[k] = qnan;
        return;
    }
    // The moment-type sums are divided exactly: a flat nucleus must give var = a^2 - a^2 = 0 and correlation 0/0 = NaN as the
    // reference's dense sums do, not a rounding residue.  The entropy-type sums, which only ever meet a subtraction of unlike
    // numbers, share one reciprocal (each f64 division is a dozen instructions on 16 of the wave's 64 lanes).
    const double T2 = 2.0 * N;
    const double rN = 1.0 / N, rT = 0.5 * rN;
    const double mu = S1 / T2;
    const double ei2 = (S2 + D2) / (4.0 * N);
    const double var = ei2 - mu * mu;
    const double eij = (S2 - D2) / (4.0 * N);
    const double con = D2 / N, dis = D1 / N, sa = S1 / N;
    const double sv = S2 / N - sa * sa;
    // All pairs in one (diagonal) cell -- a flat nucleus -- is known exactly: sum G^2 == (sum G)^2 in integers.  The three
    // entropies are then exactly 0, as a sum over cells finds them; the closed forms below would leave ln T - (T ln T)/T,
    // a rounding residue, and information_measure_correlation1 = 0/0 must come out NaN, not residue/residue.
    const bool one_cell = ASMN == T2 * T2;
    // ln of the two integers from the x ln x table (N <= 4032 pairs): two loads instead of two f64 logarithm expansions,
    // whose registers this spot of the kernel does not have
    const double lnT = WIDE ? log(T2) : lut->xlnx[(int)T2] * rT;
    const double se = one_cell ? 0.0 : (WIDE ? log(N) - HSS * rN : (lut->xlnx[(int)N] - HSS) * rN);
    const double ent = one_cell ? 0.0 : lnT - ENTS * rT;
    const double asm_ = ASMN / (T2 * T2);
    const double dv = con - dis * dis;
    const double hx = one_cell ? 0.0 : lnT - HXS * rT;
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
    o[9] = (float)(IDM * rN);
    o[10] = (float)dis;
    o[11] = (float)dv;
    o[12] = (float)((ent - hxy1) / hx);
    o[13] = (float)sqrt(a2);
}

// ---- the kernel -------------------------------------------------------------------------------------------
// QCAP: size class; serves the nuclei of its class list (K <= QCAP mask pixels).
// NPRE: 16-byte pieces of a u8 patch per lane (3 ps^2 / 1024 rounded up: 3, 7 or 12 for ps <= 32, 48, 64)
// FULL: the class's hash table holds a cell per list entry at 8/11 load, whatever the texture (the top class of a patch size, and
//       a small call's single launch).  Otherwise the table is sized for the TYPICAL nucleus -- 3/4 slot per entry: on tissue-like
//       textures a third to a half of the pairs open a new cell -- which is one more wave per CU in every class; a nucleus whose
//       cells do not fit (white noise) is abandoned at the end of the pass that filled the table and handed to the next class,
//       whose table holds a cell for every pair it can have (3/4 of a capacity at least 4/3 as large).
// Occupancy asked of the compiler: three waves per SIMD (168 registers) wherever the class's LDS lets a CU hold more than eight waves.  At
// 64 x 64 (twelve prefetch registers of four words) that costs 8 spilled registers in the small classes and buys a ninth and tenth wave
// per CU: 31.34 -> 30.87 ms per 600 k nuclei (round 3; rounds 1-2 kept those classes at two waves per SIMD, 189 registers).
template <typename T, int QCAP, int NPRE, bool FULL>
__global__ __launch_bounds__(64, (QCAP <= (FULL ? 768 : 1024) ? 3 : (QCAP <= 2048 ? 2 : 1))) void k_glcm(const int *__restrict__ list, const int *__restrict__ count, int *__restrict__ cursor,
                                             int *__restrict__ next_list, int *__restrict__ next_count, int *__restrict__ top_list, int *__restrict__ top_count, int ps,
                                             const T *__restrict__ patches, const uint64_t *__restrict__ mask_rows,
                                             const GlcmLuts *__restrict__ lut, double *__restrict__ slots, float *__restrict__ out, int F,
                                             int col0, int64_t *__restrict__ dbg)
{
    using G = GlGeom<FULL>;
    constexpr int GL_HIST_WORDS = G::HIST_WORDS, GL_D64_OFF = G::D64_OFF, GL_H128_OFF = G::H128_OFF;
    // never less than what lets the table memory (with the pass-A memo behind it: nothing of a nucleus lives there yet) double as
    // the staging area of the NPRE KiB patch
    constexpr int MEMO_WORDS = (QCAP + 64) / 4;
    constexpr int SLOTS_MIN = NPRE * 256 - GL_HIST_WORDS - GL_H128_OFF - MEMO_WORDS;
    constexpr int SLOTS_WANT = FULL ? 11 * QCAP / 8 : 3 * QCAP / 4;
    constexpr int SLOTS_B = ((QCAP - GL_H128_OFF + 3) / 4) * 4; // the level-254 pass needs a slot per pair in the whole table memory
    constexpr int SLOTS_AB = SLOTS_WANT > SLOTS_MIN ? SLOTS_WANT : SLOTS_MIN;
    constexpr int SLOTS = SLOTS_AB > SLOTS_B ? SLOTS_AB : SLOTS_B;
    static_assert(FULL || GL_H128_OFF + SLOTS >= QCAP, "the level-254 pass hashes into the whole table memory: a cell per pair fits");
    constexpr int TAB_WORDS = GL_H128_OFF + SLOTS;
    __shared__ __attribute__((aligned(16))) unsigned s_mem[TAB_WORDS + GL_HIST_WORDS + MEMO_WORDS]; // tables | histograms | memo
    __shared__ unsigned s_q[QCAP + 1];         // list entry: r << 24 | c << 16 | level128 << 8 | level254 (levels 64 and 32
                                             // are level128 >> 1 and >> 2: floor(128 g) >> 1 == floor(64 g))
    __shared__ uint64_t s_rows1[66];         // mask row r at [r + 1]; zero rows either side
    __shared__ unsigned short s_rowstart[66];
    __shared__ unsigned s_dummy[8];    // parking words for the searches of lanes without a pair (lane & 7)
    __shared__ __attribute__((aligned(16))) double s_sums[4 * GL_NACC]; // one offset's sums (four levels), stored as one 320-byte piece
    unsigned *const s_tab = s_mem, *const s_hist = s_mem + TAB_WORDS;
    unsigned char *const s_memo = reinterpret_cast<unsigned char *>(s_mem + TAB_WORDS + GL_HIST_WORDS); // [QCAP + 64]

    int lane = threadIdx.x;
    const int P = ps * ps;
    // The class list is handed out in CHUNKS of GL_CHUNK entries (single entries when the list is short against the grid: a
    // call of one reference batch wants every nucleus on its own wave): a wave's first chunk is the static one of its block index,
    // every further one comes from the class's cursor (an atomic whose answer is needed GL_CHUNK - 1 nuclei later, so its
    // latency never shows).  Nuclei of one class differ 2-3x in work; with a static stride the launch ended when the
    // unluckiest wave did (-6 % measured for the dynamic hand-out).
    const int cnt = *count;
    const int chunk = cnt >= GL_CHUNK * (int)gridDim.x ? GL_CHUNK : 1; // wave-uniform
    int j = (int)blockIdx.x * chunk;
    if (j >= cnt) return;
    int chunk_pos = 0;              // position of j inside its chunk
    int next_base = 0x7FFFFFF0;     // lane 0: first entry of the wave's next chunk (valid from chunk_pos 0 of the current one on)
    const double *__restrict__ rank_tab = lut->dO; // dD follows it
    // u8 patches travel through registers: the next nucleus's patch and mask row are requested as soon as the
    // current ones have been put down in LDS, so their HBM latency hides behind a whole nucleus of work
    const int n16 = (3 * P) / 16;
    static_assert(sizeof(T) != 1 || (int)sizeof(s_mem) >= NPRE * 1024, "the staging area holds all NPRE KiB the prefetch registers put down");
    const bool stage = sizeof(T) == 1 && ((3 * P) & 15) == 0 && 3 * P <= (int)sizeof(s_mem) && 3 * P <= NPRE * 1024 && (reinterpret_cast<uintptr_t>(patches) & 15) == 0;
    u32x4 pre[NPRE > 0 ? NPRE : 1]; // a native vector type: HIP's uint4 struct is copied with memcpy, which keeps the array in scratch memory
    uint64_t row_next;
    int i_next = list[j];
#define NFX_GLCM_REQUEST(I)                                                                                       \
    do {                                                                                                          \
        const int64_t i_ = (I);                                                                                   \
        row_next = lane < ps ? mask_rows[i_ * ps + lane] : 0ull;                                                  \
        if (stage) {                                                                                              \
            const u32x4 *src_ = reinterpret_cast<const u32x4 *>(patches + (size_t)i_ * 3 * P);                    \
            _Pragma("unroll") for (int k_ = 0; k_ < NPRE; ++k_) pre[k_] = src_[lane + 64 * k_ < n16 ? lane + 64 * k_ : 0]; \
        }                                                                                                         \
    } while (0)
    // no branches inside: all NPRE pieces are always requested (indices past the patch clamp to piece 0, one shared
    // cache line), so the pieces stay plain register values -- a conditionally written array is demoted to scratch
    // memory, and every load then waits for its own scratch store
    NFX_GLCM_REQUEST(i_next);
    if (lane < 2) s_rows1[lane ? ps + 1 : 0] = 0ull;
    if (lane < 2) s_rowstart[64 + lane] = 0;

    double *const slot = slots + (size_t)blockIdx.x * 16 * GL_NACC; // this wave's parking slot: [offset][level][GL_NACC]
    int64_t i_done = -1; // the nucleus whose sums sit in the slot, features not yet evaluated
    int done_b128 = 0, done_b254 = 0; // ... and the levels its counts were taken relative to (wave-uniform)
#pragma unroll 1
    for (;;) {
        const bool more = j < cnt; // wave-uniform
        const int64_t i = i_next;
        const uint64_t myrow = row_next;
        if (more && stage) {
            u32x4 *dst = reinterpret_cast<u32x4 *>(s_mem);
#pragma unroll
            for (int k = 0; k < NPRE; ++k) dst[lane + 64 * k] = pre[k]; // s_mem holds NPRE KiB: pieces past the patch land in table / memo memory that is swept or rewritten below
        }
        // ---- the 14 features of the PREVIOUS nucleus's 16 combinations (texture.rs:81-157 order: level-major, then offset) ----
        // evaluated here, between putting down this nucleus's patch and requesting the next one's: the only point of the loop
        // where the prefetch registers are dead, so the f64 logarithms do not push the kernel over its register budget.
        // The slot was written by other lanes of this wave: the stores have to be complete (release = s_waitcnt vmcnt(0)) and
        // the loads must not be served from a line the CU's L1 kept from an earlier nucleus (agent-scope loads bypass L1);
        // this nucleus's first slot store comes two passes later, after the loads' data has been used.
        __builtin_amdgcn_sched_barrier(0); // the patch pieces are put down before, the next request is issued after: not across
        // every lane-derived address and constant of the nucleus below is recomputed per nucleus (a handful of instructions)
        // instead of sitting in ~100 hoisted registers across this point
        asm volatile("" : "+v"(lane));
        if (i_done >= 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            if (lane < 16) {
                const int lvl = lane >> 2, oi = lane & 3;
                const double *src = slot + (oi * 4 + lvl) * GL_NACC;
                double a[GL_NACC];
#pragma unroll
                for (int k = 0; k < GL_NACC; ++k)
                    a[k] = __longlong_as_double(__hip_atomic_load(reinterpret_cast<const long long *>(src) + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                const double base = (double)(lvl == 3 ? done_b254 : (done_b128 >> (2 - lvl))); // levels 32, 64, 128 nest: floor(128 g) >> k
                glcm_features<false>(a, base, lut, out + i_done * (int64_t)F + col0 + lane * 14, dbg ? dbg + (i_done * 16 + lane) * 6 : nullptr);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!more) break;
        i_done = i;
        if (chunk_pos == 0 && lane == 0) next_base = (int)gridDim.x * chunk + atomicAdd(cursor, chunk);
        const int jn = chunk_pos < chunk - 1 ? j + 1 : __builtin_amdgcn_readfirstlane(next_base);
        chunk_pos = chunk_pos < chunk - 1 ? chunk_pos + 1 : 0;
        if (jn < cnt) {
            i_next = list[jn];
            NFX_GLCM_REQUEST(i_next);
        }
        // ---- row prefix ---------------------------------------------------------------------------------------
        const int rcnt = __popcll(myrow);
        const int inc = wave_inclusive_scan_i32(rcnt);
        const int K = __builtin_amdgcn_readlane(inc, 63);
        const int excl = inc - rcnt;
        if (lane < ps) s_rows1[lane + 1] = myrow;
        s_rowstart[lane] = (unsigned short)excl;
        // ---- position list: stores only, no LDS round trip in the loop ---------------------------------------
        const unsigned row_lo = (unsigned)myrow, row_hi = (unsigned)(myrow >> 32);
        for (int r = 0; r < ps; ++r) {
            const unsigned wlo = __builtin_amdgcn_readlane(row_lo, r), whi = __builtin_amdgcn_readlane(row_hi, r);
            if ((wlo | whi) == 0u) continue;
            const int base = __builtin_amdgcn_readlane(excl, r);
            const uint64_t w = ((uint64_t)whi << 32) | wlo;
            if ((w >> lane) & 1ull) s_q[base + __builtin_amdgcn_mbcnt_hi(whi, __builtin_amdgcn_mbcnt_lo(wlo, 0u))] = ((unsigned)r << 24) | ((unsigned)lane << 16);
        }
        wave_lds_fence();
        // ---- every masked pixel is read and quantised once, all lanes busy --------------------------------------
        const T *pp = patches + (size_t)i * 3 * P;
        int mx128 = 0, mn128n = 0, mx254 = 0, mn254n = 0; // maxima of q and of (top - q): minima through the same non-negative maximum
        for (int e = lane; e < K; e += 128) { // two entries per lane: both read chains in flight
            const int e2 = e + 64 < K ? e + 64 : e;
            const unsigned pos = s_q[e], pos2 = s_q[e2];
            const int p = (int)(pos >> 24) * ps + (int)((pos >> 16) & 255u), p2 = (int)(pos2 >> 24) * ps + (int)((pos2 >> 16) & 255u);
            float g, g2;
            if (stage) {
                const uint8_t *sb = reinterpret_cast<const uint8_t *>(s_mem);
                const unsigned r0 = sb[p], g0 = sb[P + p], b0 = sb[2 * P + p], r1 = sb[p2], g1 = sb[P + p2], b1 = sb[2 * P + p2];
                float gs = u8_unit(r0) + u8_unit(g0), gs2 = u8_unit(r1) + u8_unit(g1); // utils.rs:172
                gs = gs + u8_unit(b0);
                gs2 = gs2 + u8_unit(b1);
                g = third_of_u8_sum(gs);                                                // texture.rs:36
                g2 = third_of_u8_sum(gs2);
            } else {
                g = gray_at<T>(pp, P, p);
                g2 = gray_at<T>(pp, P, p2);
            }
            const unsigned qa128 = quant(g, 128), qa254 = quant(g, 254), qb128 = quant(g2, 128), qb254 = quant(g2, 254); // texture.rs:19
            s_q[e] = pos | (qa128 << 8) | qa254;
            if (e + 64 < K) s_q[e + 64] = pos2 | (qb128 << 8) | qb254;
            // (entry e2 repeats entry e when e + 64 is past the list: harmless for a minimum / maximum)
            mx128 = max(mx128, (int)max(qa128, qb128)); mn128n = max(mn128n, 127 - (int)min(qa128, qb128));
            mx254 = max(mx254, (int)max(qa254, qb254)); mn254n = max(mn254n, 253 - (int)min(qa254, qb254));
        }
        // ---- the nucleus's level span: counts are taken relative to its lowest level (GlGeom), in EVERY instantiation -- a
        // nucleus's bins meet the same lanes and its sums the same order whichever class it runs in: identical bits -----------
        const int hi128 = wave_max_nonneg_i32(mx128), lo128 = 127 - wave_max_nonneg_i32(mn128n);
        const int hi254 = wave_max_nonneg_i32(mx254), lo254 = 253 - wave_max_nonneg_i32(mn254n);
        const int b128 = K > 0 ? (lo128 & ~3) : 0; // a multiple of 4: levels 64 and 32 stay shifts of the relative level 128
        const int b254 = K > 0 ? lo254 : 0;
        const bool span_fits = FULL || K == 0 || (hi128 - b128 < G::R128 && hi254 - b254 < G::R254); // wave-uniform
        const u16x2 base2 = {(unsigned short)b128, (unsigned short)b128}, base3 = {(unsigned short)b254, (unsigned short)b254};
        if (lane == 0) s_q[K] = ((unsigned)b128 << 8) | (unsigned)b254; // what an entry without a partner may read as its partner: a word of the list's own kind
        wave_lds_fence();
        sweep(s_hist, GL_HIST_WORDS, 0u, lane);
        if (lane < 8) s_dummy[lane] = GL_DUMMY;
        bool abandoned = false;
#pragma unroll 1
        for (int oi = 0; oi < (span_fits ? 4 : 0); ++oi) {
            const int dy = kGlcmOffsets[oi][0], dx = kGlcmOffsets[oi][1];
            sweep(s_tab, GL_H128_OFF, 0u, lane);               // pass A's dense tables start from zero
            sweep(s_tab + GL_H128_OFF, SLOTS, GL_EMPTY, lane); // and its hash table from empty
            wave_lds_fence();
            // ================= pass A: levels 32, 64, 128 ==========================================================
            // histogram words: hx128 0..127 | hs32 128..190 | hs64 192..318 | hs128 320..574.  The marginal histogram
            // is filled per PIXEL (a pixel of level a that takes part in deg pairs adds deg to p_x(a)) and only at
            // the finest of the nested levels: floor(128 g) >> 1 = floor(64 g), so hx64 and hx32 fold out of hx128.
            LevelAcc A0 = {0, 0, 0, 0, 0, 0.0f, 0.0}, A1 = A0, A2 = A0, A3 = A0;
            unsigned np = 0;
            unsigned pend[2][3] = {{0u, 0u, 0u}, {0u, 0u, 0u}}; // rank-table indices of the previous iteration (entry 0 holds 0.0)
            bool table_full = false;
#if NFX_GL_DERIVE
            // level 128 alone is counted pair by pair (its marginal and sum histograms, its cells in the hash table with their ranks)
#pragma unroll 1
            for (int e = lane; e < K; e += 128) {
                const double ev0 = lut_at(rank_tab, pend[0][2]), ev1 = lut_at(rank_tab, pend[1][2]);
                const Located X = locate(e, true, ps, dy, dx, s_q, s_rows1, s_rowstart);
                const Located Y = locate(e + 64, e + 64 < K, ps, dy, dx, s_q, s_rows1, s_rowstart);
                const u16x2 a2 = bytes2<1>(X.qa, Y.qa) - base2, b2 = bytes2<1>(X.qb, Y.qb) - base2;
                const Ordered2 P2 = order2(a2, b2);
                const unsigned fmask = (X.fwd ? 0xFFFFu : 0u) | (Y.fwd ? 0xFFFF0000u : 0u);
                s_memo[e] = (unsigned char)X.memo;
                s_memo[e + 64] = (unsigned char)Y.memo;
                atomicAdd(&s_hist[a2.x], X.deg);
                atomicAdd(&s_hist[a2.y], Y.deg);
                np += (X.fwd ? 1u : 0u) + (Y.fwd ? 1u : 0u);
                const u16x2 s2 = a2 + b2;
                atomicAdd(&s_hist[320 + s2.x], X.fwd ? 1u : 0u);
                atomicAdd(&s_hist[320 + s2.y], Y.fwd ? 1u : 0u);
                unsigned rx2, ry2;
                const u16x2 k2 = cell_key2(P2);
                hash_rank2<SLOTS>(s_tab + GL_H128_OFF, s_dummy + (lane & 7), k2.x, X.fwd, k2.y, Y.fwd, rx2, ry2, table_full);
                pair_acc2(X.fwd, Y.fwd, fmask, P2, rx2, ry2, A2, pend[0][2], pend[1][2]);
                A2.ents += ev0 + ev1;
            }
            A2.ents += lut_at(rank_tab, pend[0][2]) + lut_at(rank_tab, pend[1][2]);
            wave_lds_fence();
            if (!FULL && __any(table_full)) { abandoned = true; break; } // wave-uniform
            // levels 64 and 32 from the DISTINCT level-128 cells: slot word = key << 16 | pairs, key = hi << 8 | lo (relative levels, lo <= hi).  A cell of
            // c pairs adds c to its coarse cell (dense triangular tables, packed 16-bit counts), to the sum histogram at lo' + hi' and to the difference
            // histogram at hi' - lo' -- integers, so the order the slots are met in (it depends on the class's table size) does not show in any sum
            for (int sl = lane; sl < SLOTS; sl += 64) {
                const unsigned wd = s_tab[GL_H128_OFF + sl];
                if (wd != GL_EMPTY) {
                    const unsigned c = wd & 0xFFFFu, lo = (wd >> 16) & 255u, hi = wd >> 24;
                    const unsigned lo1 = lo >> 1, hi1 = hi >> 1, lo0 = lo >> 2, hi0 = hi >> 2;
                    const unsigned t1 = ((hi1 * (hi1 + 1u)) >> 1) + lo1, t0 = ((hi0 * (hi0 + 1u)) >> 1) + lo0;
                    atomicAdd(&s_tab[GL_D64_OFF + (t1 >> 1)], c << ((t1 & 1u) * 16u));
                    atomicAdd(&s_tab[t0 >> 1], c << ((t0 & 1u) * 16u));
                    atomicAdd(&s_hist[192 + lo1 + hi1], c);
                    atomicAdd(&s_hist[128 + lo0 + hi0], c);
                    atomicAdd(&s_hist[G::PD_OFF + (hi1 - lo1)], c);
                    atomicAdd(&s_hist[G::PD_OFF + (hi0 - lo0)], c << 16);
                }
            }
            wave_lds_fence();
            {   // sum G^2 and sum G ln G of levels 64 and 32 from the final counts of their dense tables (only the rows the nucleus's span reaches): a cell of
                // m pairs stands for G[a][b] = G[b][a] = m off the diagonal (2 m^2, 2 m ln m) and for G[a][a] = 2 m on it (4 m^2, 2m ln 2m); which cells are
                // diagonal comes from the row-end table.  Look-ups of one trip are used in the next (their round trip hides behind it)
                const int span64 = (hi128 - b128) >> 1, span32 = span64 >> 1;
                const int words64 = ((span64 * (span64 + 3)) >> 2) + 1, words32 = ((span32 * (span32 + 3)) >> 2) + 1; // words up to the last row's diagonal cell
                auto dense_sums = [&](const unsigned *tab, int words, unsigned &asmn, double &ents) {
                    double p0 = 0.0, p1 = 0.0;
                    for (int w = lane; w < words; w += 64) {
                        const unsigned v = tab[w], re = lut->rowend[w];
                        const unsigned m0 = v & 0xFFFFu, m1 = v >> 16;
                        const bool dg0 = (unsigned)(2 * w) == re, dg1 = (unsigned)(2 * w + 1) == re;
                        asmn += (dg0 ? 4u : 2u) * m0 * m0 + (dg1 ? 4u : 2u) * m1 * m1;
                        ents += p0 + p1;
                        const double x0 = lut_at(lut->xlnx, dg0 ? 2u * m0 : m0), x1 = lut_at(lut->xlnx, dg1 ? 2u * m1 : m1);
                        p0 = dg0 ? x0 : x0 + x0;
                        p1 = dg1 ? x1 : x1 + x1;
                    }
                    ents += p0 + p1;
                };
                dense_sums(s_tab + GL_D64_OFF, words64, A1.asmn, A1.ents);
                dense_sums(s_tab, words32, A0.asmn, A0.ents);
                // the difference histograms: lane d holds the pairs with |a - b| = d (level 64 in the low half, level 32 in the high half)
                const unsigned pdw = s_hist[G::PD_OFF + lane], p64 = pdw & 0xFFFFu, p32 = pdw >> 16, dl = (unsigned)lane;
                const float rinv = __builtin_amdgcn_rcpf(1.0f + (float)(dl * dl));
                A1.D1 = dl * p64; A1.D2 = dl * dl * p64; A1.idm = (float)p64 * rinv;
                A0.D1 = dl * p32; A0.D2 = dl * dl * p32; A0.idm = (float)p32 * rinv;
            }
#else
#pragma unroll 1
            for (int e = lane; e < K; e += 128) {
                double ev[2][3];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int l = 0; l < 3; ++l) ev[u][l] = lut_at(rank_tab, pend[u][l]);
                const Located X = locate(e, true, ps, dy, dx, s_q, s_rows1, s_rowstart);
                const Located Y = locate(e + 64, e + 64 < K, ps, dy, dx, s_q, s_rows1, s_rowstart);
                // levels of both entries packed: .x = entry e, .y = entry e + 64 (level128 is byte 1 of a list word; 64 and 32 by shift)
                const u16x2 a2 = bytes2<1>(X.qa, Y.qa) - base2, b2 = bytes2<1>(X.qb, Y.qb) - base2; // (an entry without a partner reads a list word all the same)
                const u16x2 a1 = a2 >> (u16x2){1, 1}, b1 = b2 >> (u16x2){1, 1};
                const u16x2 a0 = a2 >> (u16x2){2, 2}, b0 = b2 >> (u16x2){2, 2};
                const Ordered2 P0 = order2(a0, b0), P1 = order2(a1, b1), P2 = order2(a2, b2);
                const unsigned fmask = (X.fwd ? 0xFFFFu : 0u) | (Y.fwd ? 0xFFFF0000u : 0u);
                s_memo[e] = (unsigned char)X.memo;
                s_memo[e + 64] = (unsigned char)Y.memo;
                atomicAdd(&s_hist[a2.x], X.deg);
                atomicAdd(&s_hist[a2.y], Y.deg);
                np += (X.fwd ? 1u : 0u) + (Y.fwd ? 1u : 0u);
                const u16x2 t0 = tri_index2(P0), t1 = tri_index2(P1);
                const unsigned rx0 = dense_rank_if(s_tab, t0.x, X.fwd);
                const unsigned ry0 = dense_rank_if(s_tab, t0.y, Y.fwd);
                const unsigned rx1 = dense_rank_if(s_tab + GL_D64_OFF, t1.x, X.fwd);
                const unsigned ry1 = dense_rank_if(s_tab + GL_D64_OFF, t1.y, Y.fwd);
                const u16x2 s0 = a0 + b0, s1 = a1 + b1, s2 = a2 + b2;
                atomicAdd(&s_hist[128 + s0.x], X.fwd ? 1u : 0u);
                atomicAdd(&s_hist[128 + s0.y], Y.fwd ? 1u : 0u);
                atomicAdd(&s_hist[192 + s1.x], X.fwd ? 1u : 0u);
                atomicAdd(&s_hist[192 + s1.y], Y.fwd ? 1u : 0u);
                atomicAdd(&s_hist[320 + s2.x], X.fwd ? 1u : 0u);
                atomicAdd(&s_hist[320 + s2.y], Y.fwd ? 1u : 0u);
                unsigned rx2, ry2;
                const u16x2 k2 = cell_key2(P2);
                hash_rank2<SLOTS>(s_tab + GL_H128_OFF, s_dummy + (lane & 7), k2.x, X.fwd, k2.y, Y.fwd, rx2, ry2, table_full);
                pair_acc2(X.fwd, Y.fwd, fmask, P0, rx0, ry0, A0, pend[0][0], pend[1][0]);
                pair_acc2(X.fwd, Y.fwd, fmask, P1, rx1, ry1, A1, pend[0][1], pend[1][1]);
                pair_acc2(X.fwd, Y.fwd, fmask, P2, rx2, ry2, A2, pend[0][2], pend[1][2]);
                A0.ents += ev[0][0] + ev[1][0]; A1.ents += ev[0][1] + ev[1][1]; A2.ents += ev[0][2] + ev[1][2];
            }
            A0.ents += lut_at(rank_tab, pend[0][0]) + lut_at(rank_tab, pend[1][0]);
            A1.ents += lut_at(rank_tab, pend[0][1]) + lut_at(rank_tab, pend[1][1]);
            A2.ents += lut_at(rank_tab, pend[0][2]) + lut_at(rank_tab, pend[1][2]);
            wave_lds_fence();
            if (!FULL && __any(table_full)) { abandoned = true; break; } // wave-uniform
#endif
            unsigned np_w;
            {
                // fold: read and clear the histogram words, request every x ln x together, and only then do the
                // table sweep and the power-sum reductions, so the look-ups' round trip hides behind them
                const uint2 h2 = reinterpret_cast<const uint2 *>(s_hist)[lane]; // hx128[2 lane], hx128[2 lane + 1]
                constexpr int NS = G::HS128_SLICES;                               // slices of the level-128 sum histogram: 4, or 3 (zeros beyond)
                unsigned hv[7];
#pragma unroll
                for (int m = 0; m < 7; ++m) hv[m] = m < 3 + NS ? s_hist[128 + lane + 64 * m] : 0u;
                const unsigned n64 = h2.x + h2.y;                       // hx64[lane]
                const unsigned n32 = n64 + (unsigned)__shfl_down((int)n64, 1, 64); // hx32[lane / 2] on even lanes
                const unsigned n32e = (lane & 1) ? 0u : n32;
                const double f0 = lut_at(lut->xlnx, h2.x), f1 = lut_at(lut->xlnx, h2.y);
                const double f64_ = lut_at(lut->xlnx, n64), f32_ = lut_at(lut->xlnx, n32e);
                double fs[7];
#pragma unroll
                for (int m = 0; m < 7; ++m) fs[m] = m < 3 + NS ? lut_at(lut->xlnx, hv[m]) : 0.0;
                sweep(s_hist, G::HIST_A, 0u, lane);
                sweep(s_tab, TAB_WORDS, GL_EMPTY, lane); // pass B's hash table is the WHOLE table memory (d32, d64 and h128 are
                                                          // done): more slots per entry, so fewer searches need a second probe
                                                          // and the wave's iteration -- as slow as its unluckiest key -- ends sooner
                const double hx128 = f0 + f1;
                const double hx64 = f64_, hx32 = f32_;
                double e[7];
#pragma unroll
                for (int m = 0; m < 7; ++m) e[m] = fs[m];
                const double hs32 = e[0], hs64 = e[1] + e[2], hs128 = (e[3] + e[4]) + (e[5] + e[6]); // (e[6] is an exact zero in the 3-slice layout)
                {   // sum s n_s and sum s^2 n_s: word 128 + lane + 64 m holds n_s of s = lane (level 32), lane + 64 (m - 1), lane + 64 (m - 3)
                    const unsigned s0 = (unsigned)lane, s1 = s0 + 64u, s2 = s0 + 128u, s3 = s0 + 192u;
                    A0.S1 = s0 * hv[0];                A0.S2 = s0 * s0 * hv[0];
                    A1.S1 = s0 * hv[1] + s1 * hv[2];   A1.S2 = s0 * s0 * hv[1] + s1 * s1 * hv[2];
                    A2.S1 = (s0 * hv[3] + s1 * hv[4]) + (s2 * hv[5] + s3 * hv[6]);
                    A2.S2 = (s0 * s0 * hv[3] + s1 * s1 * hv[4]) + (s2 * s2 * hv[5] + s3 * s3 * hv[6]);
                }
                // 28 wave totals at once (wave_ops.h: reduce-scatter butterflies): lane 48 + j ends up with one of them
                unsigned U[16] = {A0.S1, A0.S2, A0.D1, A0.D2, A0.asmn, A1.S1, A1.S2, A1.D1, A1.D2, A1.asmn,
                                  A2.S1, A2.S2, A2.D1, A2.D2, A2.asmn, np};
                double Dv[8] = {A0.ents, hx32, hs32, A1.ents, hx64, hs64, A2.ents, hx128};
                float Fv[4] = {A0.idm, A1.idm, A2.idm, 0.0f};
                const unsigned ut = wave_reduce_scatter(U, lane);
                const double dt = wave_reduce_scatter(Dv, lane);
                const float ft = wave_reduce_scatter(Fv, lane);
                const double hs128t = wave_total_f64(hs128);
                np_w = (unsigned)__builtin_amdgcn_readlane((int)ut, 63); // scatter_index<16>(15) == 15
                if (lane >= 48) {
                    const int j = lane - 48;
                    const int iu = scatter_index<16>(j), id = scatter_index<8>(j & 7), ifl = scatter_index<4>(j & 3);
                    if (iu < 15) { // level * 5 + {S1, S2, D1, D2, asmn} -> slots 1, 2, 3, 4, 6
                        const int lvl = iu / 5, f = iu - 5 * lvl;
                        s_sums[lvl * GL_NACC + (f < 4 ? 1 + f : 6)] = (double)ut;
                    }
                    s_sums[(id / 3) * GL_NACC + 7 + id % 3] = dt;                 // level * 3 + {ents, hxs, hss} (hss of level 128 below)
                    if (ifl < 3) s_sums[ifl * GL_NACC + 5] = (double)ft;         // idm
                }
                if (lane == 0) {
                    s_sums[0 * GL_NACC] = s_sums[1 * GL_NACC] = s_sums[2 * GL_NACC] = (double)np_w;
                    s_sums[2 * GL_NACC + 9] = hs128t;
                }
            }
            wave_lds_fence();
            // ================= pass B: level 254 ===================================================================
            // histogram words: hx254 at 0 (HXB slices of 64) | hs254 behind it (2 HXB slices); levels relative to the nucleus's lowest
            unsigned pend3[2] = {0u, 0u};
            bool table_full_b = false; // (cannot happen: TAB_WORDS >= QCAP, asserted above)
#pragma unroll 1
            for (int e = lane; e < K; e += 128) {
                const double ev30 = lut_at(rank_tab, pend3[0]), ev31 = lut_at(rank_tab, pend3[1]);
                const Located X = relocate(e, true, s_q, s_memo);
                const Located Y = relocate(e + 64, e + 64 < K, s_q, s_memo);
                const u16x2 a3 = bytes2<0>(X.qa, Y.qa) - base3, b3 = bytes2<0>(X.qb, Y.qb) - base3; // level254 is byte 0 of a list word
                const Ordered2 P3 = order2(a3, b3);
                const unsigned fmask = (X.fwd ? 0xFFFFu : 0u) | (Y.fwd ? 0xFFFF0000u : 0u);
                atomicAdd(&s_hist[a3.x], X.deg);
                atomicAdd(&s_hist[a3.y], Y.deg);
                const u16x2 s3 = a3 + b3;
                atomicAdd(&s_hist[64 * G::HXB + s3.x], X.fwd ? 1u : 0u);
                atomicAdd(&s_hist[64 * G::HXB + s3.y], Y.fwd ? 1u : 0u);
                unsigned rx, ry;
                const u16x2 k3 = cell_key2(P3);
                hash_rank2<TAB_WORDS>(s_tab, s_dummy + (lane & 7), k3.x, X.fwd, k3.y, Y.fwd, rx, ry, table_full_b);
                pair_acc2(X.fwd, Y.fwd, fmask, P3, rx, ry, A3, pend3[0], pend3[1]);
                A3.ents += ev30 + ev31;
            }
            A3.ents += lut_at(rank_tab, pend3[0]) + lut_at(rank_tab, pend3[1]);
            wave_lds_fence();
            {
                constexpr int HXB = G::HXB, HSB = G::HSB; // slices of hx254 and of hs254 that exist in this layout (4 + 8, or 3 + 6)
                // the sums below are written for the full layout; a slice the 3-slice layout lacks is an exact zero in it (and holds
                // only zeros for such a nucleus in the full one), so both layouts add the same numbers in the same order
                unsigned hx_[4], hs_[8];
#pragma unroll
                for (int m = 0; m < 4; ++m) hx_[m] = m < HXB ? s_hist[lane + 64 * m] : 0u;
#pragma unroll
                for (int m = 0; m < 8; ++m) hs_[m] = m < HSB ? s_hist[64 * HXB + lane + 64 * m] : 0u;
                double ex[4], es[8];
#pragma unroll
                for (int m = 0; m < 4; ++m) ex[m] = m < HXB ? lut_at(lut->xlnx, hx_[m]) : 0.0;
#pragma unroll
                for (int m = 0; m < 8; ++m) es[m] = m < HSB ? lut_at(lut->xlnx, hs_[m]) : 0.0;
                sweep(s_hist, G::HIST_B, 0u, lane);
                A3.S1 = 0u; A3.S2 = 0u;
#pragma unroll
                for (int m = 0; m < 8; ++m) { const unsigned sv = (unsigned)lane + 64u * (unsigned)m; A3.S1 += sv * hs_[m]; A3.S2 += sv * sv * hs_[m]; }
                const double hx = (ex[0] + ex[1]) + (ex[2] + ex[3]);
                const double hs = ((es[0] + es[1]) + (es[2] + es[3])) + ((es[4] + es[5]) + (es[6] + es[7]));
                unsigned U[8] = {A3.S1, A3.S2, A3.D1, A3.D2, A3.asmn, 0u, 0u, 0u};
                double Dv[4] = {A3.ents, hx, hs, 0.0};
                const unsigned ut = wave_reduce_scatter(U, lane);
                const double dt = wave_reduce_scatter(Dv, lane);
                const float idm = wave_total_f32(A3.idm);
                if (lane >= 48) {
                    const int j = lane - 48;
                    const int iu = scatter_index<8>(j & 7), id = scatter_index<4>(j & 3);
                    if (iu < 5) s_sums[3 * GL_NACC + (iu < 4 ? 1 + iu : 6)] = (double)ut;
                    if (id < 3) s_sums[3 * GL_NACC + 7 + id] = dt;
                }
                if (lane == 0) { s_sums[3 * GL_NACC] = (double)np_w; s_sums[3 * GL_NACC + 5] = (double)idm; }
            }
            wave_lds_fence();
            // lane-0 stores straight to memory would touch five partial lines per combination
            if (lane < 4 * GL_NACC) slot[oi * 4 * GL_NACC + lane] = s_sums[lane];
        }
        if (abandoned) { // more distinct level-128 cells than the typical-size table holds: the next class takes the nucleus
            if (lane == 0) next_list[atomicAdd(next_count, 1)] = (int)i;
            i_done = -1; // nothing of it is evaluated here
        } else if (!span_fits) { // gray levels spread over more of the scale than the typical-size tables cover: the top class (full range) takes it
            if (lane == 0) top_list[atomicAdd(top_count, 1)] = (int)i;
            i_done = -1;
        }
        done_b128 = b128; done_b254 = b254;
        j = jn;
    }
}

// ---- patch sizes 65..128: one 256-thread workgroup per nucleus (see k_geometry.hip, "patch sizes 65..128") -----------------
// Up to 16 384 mask pixels do not fit one wave's LDS lists.  Here a WORKGROUP owns the nucleus: the masked pixels are compacted into an
// LDS list (position and levels, as in the wave kernel), and each of the 16 (level, offset) combinations is one pass of all 256
// threads over the list: the partner is located through the mask rows, the unordered cell {a, b} is counted in an open-addressing
// LDS hash table (claim with compare-and-swap, count with a plain add), the marginal and the sum histogram with LDS atomics, the pair
// sums in registers.  Afterwards the table and the histograms are swept: sum G^2 and the three entropy sums come from the FINAL
// counts, the latter in 32.32 FIXED POINT (integer additions: whatever order the waves' atomics landed in and wherever a cell
// found its slot, the totals are the same bits -- a multi-wave table has no reproducible insertion order to telescope ranks over).
// Two capacities: masks up to 6144 pixels (62 KB of LDS: two workgroups per CU) and up to 128 x 128 (150 KB); both launches walk
// all nuclei through one cursor and skip the other's.  The ten sums feed the same glcm_features as the wave kernel.
static constexpr long long GW_FIX = 1ll << 32;
static __device__ __forceinline__ long long gw_xlnx_fix(const GlcmLuts *__restrict__ lut, unsigned g) // x ln x in 32.32 fixed point
{
    const double v = g < (unsigned)GL_COUNTS ? lut->xlnx[g] : (double)g * log((double)g);
    return (long long)(v * (double)GW_FIX + 0.5);
}
static __device__ __forceinline__ unsigned long long gw_wave_total_u64(unsigned long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (unsigned long long)__shfl_xor((long long)v, o, 64);
    return v;
}

template <typename T, int KCAP, int SLOTS>
__global__ __launch_bounds__(256) void k_glcm_wide(int64_t n, int ps, const T *__restrict__ patches, const uint64_t *__restrict__ mask_rows,
                                                   const GlcmLuts *__restrict__ lut, int *__restrict__ cursor, int kmin, float *__restrict__ out, int F,
                                                   int col0, int64_t *__restrict__ dbg)
{
    __shared__ unsigned s_q[KCAP];                       // r << 24 | c << 16 | level128 << 8 | level254
    __shared__ unsigned s_tab[SLOTS];                    // key << 16 | count of the unordered cell (key = hi << 8 | lo)
    __shared__ unsigned s_hx[256], s_hs[512];
    __shared__ uint64_t s_rows[NFX_MAX_PATCH_SIZE][2];
    __shared__ unsigned short s_start[NFX_MAX_PATCH_SIZE + 1][2]; // list index of the first pixel of (row, word)
    __shared__ unsigned long long s_acc[10];             // np, S1, S2, D1, D2, asmn | ents, hxs, hss (fixed point) | spare
    __shared__ double s_idm[4];                          // per wave: added in wave order
    __shared__ double s_sums[16][GL_NACC];
    __shared__ int s_i, s_cnt[4];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int P = ps * ps;
    for (;;) {
        if (t == 0) s_i = atomicAdd(cursor, 1);
        __syncthreads();
        const int64_t i = s_i;
        if (i >= n) break;
        // ---- mask rows, the list's row starts, the mask's size ----------------------------------------------------------------
        for (int k = t; k < 2 * ps; k += 256) s_rows[k >> 1][k & 1] = mask_rows[(size_t)i * ps * 2 + k];
        __syncthreads();
        {
            const int r = t >> 1, w = t & 1; // thread = (row, word)
            const int cnt = r < ps ? __popcll(s_rows[r][w]) : 0;
            const int incl = wave_inclusive_scan_i32(cnt);
            if (lane == 63) s_cnt[wv] = incl;
            __syncthreads();
            int base = 0;
            for (int k = 0; k < wv; ++k) base += s_cnt[k];
            if (r < ps) s_start[r][w] = (unsigned short)(base + incl - cnt);
        }
        const int K = (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]);
        if ((KCAP > 8192) != (K > kmin)) { __syncthreads(); continue; } // the other capacity's launch takes it (uniform over the workgroup)
        // ---- the list: positions, then levels --------------------------------------------------------------------------------
        {
            const int r = t >> 1, w = t & 1;
            if (r < ps) {
                uint64_t bits = s_rows[r][w];
                unsigned at = s_start[r][w];
                while (bits) {
                    const int c = 64 * w + __ffsll((long long)bits) - 1;
                    bits &= bits - 1ull;
                    s_q[at++] = ((unsigned)r << 24) | ((unsigned)c << 16);
                }
            }
        }
        __syncthreads();
        const T *pp = patches + (size_t)i * 3 * P;
        for (int e = t; e < K; e += 256) {
            const unsigned pos = s_q[e];
            const int p = (int)(pos >> 24) * ps + (int)((pos >> 16) & 255u);
            float g;
            if (sizeof(T) == 1) {
                const uint8_t *pb = reinterpret_cast<const uint8_t *>(pp);
                float gs = u8_unit(pb[p]) + u8_unit(pb[P + p]); // utils.rs:172
                gs = gs + u8_unit(pb[2 * P + p]);
                g = third_of_u8_sum(gs);                          // texture.rs:36
            } else {
                g = gray_at<T>(pp, P, p);
            }
            s_q[e] = pos | (quant(g, 128) << 8) | quant(g, 254); // texture.rs:19
        }
        __syncthreads();
        // ---- the sixteen combinations ------------------------------------------------------------------------------------------
#pragma unroll 1
        for (int li = 0; li < 4; ++li) {
#pragma unroll 1
            for (int oi = 0; oi < 4; ++oi) {
                const int dy = kGlcmOffsets[oi][0], dx = kGlcmOffsets[oi][1];
                for (int k = t; k < SLOTS; k += 256) s_tab[k] = GL_EMPTY;
                s_hx[t] = 0u; s_hs[t] = 0u; s_hs[256 + t] = 0u;
                if (t < 10) s_acc[t] = 0ull;
                __syncthreads();
                unsigned long long np = 0, S1 = 0, S2 = 0, D1 = 0, D2 = 0;
                double idm = 0.0;
                for (int e = t; e < K; e += 256) {
                    const unsigned qa = s_q[e];
                    const int r2 = (int)(qa >> 24) + dy, c2 = (int)((qa >> 16) & 255u) + dx;
                    if ((unsigned)r2 >= (unsigned)ps || (unsigned)c2 >= (unsigned)ps) continue;
                    const uint64_t w2 = s_rows[r2][c2 >> 6];
                    if (!((w2 >> (c2 & 63)) & 1ull)) continue;
                    const unsigned qb = s_q[s_start[r2][c2 >> 6] + (unsigned)__popcll(w2 & ((1ull << (c2 & 63)) - 1ull))];
                    unsigned a, b;
                    if (li == 3) { a = qa & 255u; b = qb & 255u; }
                    else { a = ((qa >> 8) & 255u) >> (2 - li); b = ((qb >> 8) & 255u) >> (2 - li); } // floor(128 g) >> k = floor(2^(7-k) g)
                    const unsigned lo = a < b ? a : b, hi = a < b ? b : a, d = hi - lo, sm = a + b;
                    np += 1; S1 += sm; S2 += (unsigned long long)sm * sm; D1 += d; D2 += (unsigned long long)d * d;
                    idm += 1.0 / (1.0 + (double)(d * d));
                    atomicAdd(&s_hx[a], 1u);
                    atomicAdd(&s_hx[b], 1u);
                    atomicAdd(&s_hs[sm], 1u);
                    const unsigned key = (hi << 8) | lo;
                    unsigned h = (unsigned)__umul24(key, 0x9E3779u) >> 8;
                    h = (unsigned)(((unsigned long long)(h & 0xFFFFu) * (unsigned)SLOTS) >> 16);
                    for (int probes = 0; probes < SLOTS; ++probes) { // a cell per pair fits (SLOTS > KCAP): the search always ends
                        const unsigned prev = atomicCAS(&s_tab[h], GL_EMPTY, (key << 16) | 1u);
                        if (prev == GL_EMPTY) break;
                        if ((prev >> 16) == key) { atomicAdd(&s_tab[h], 1u); break; }
                        h = h + 1u == (unsigned)SLOTS ? 0u : h + 1u;
                    }
                }
                __syncthreads();
                // ---- sweep: sum G^2 and sum G ln G over the symmetric matrix from the final counts; marginal and sum entropies ----
                unsigned long long asmn = 0;
                long long ents = 0, hxs = 0, hss = 0;
                for (int k = t; k < SLOTS; k += 256) {
                    const unsigned v = s_tab[k];
                    if (v == GL_EMPTY) continue;
                    const unsigned cnt = v & 0xFFFFu, key = v >> 16;
                    const bool diag = (key >> 8) == (key & 255u);
                    const unsigned gsym = diag ? 2u * cnt : cnt;   // the symmetric matrix holds 2 cnt on the diagonal, cnt twice off it
                    asmn += (unsigned long long)gsym * gsym * (diag ? 1u : 2u);
                    ents += gw_xlnx_fix(lut, gsym) * (diag ? 1 : 2);
                }
                hxs += gw_xlnx_fix(lut, s_hx[t]);
                hss += gw_xlnx_fix(lut, s_hs[t]) + gw_xlnx_fix(lut, s_hs[256 + t]);
                {
                    unsigned long long v[9] = {np, S1, S2, D1, D2, asmn, (unsigned long long)ents, (unsigned long long)hxs, (unsigned long long)hss};
#pragma unroll
                    for (int k = 0; k < 9; ++k) {
                        const unsigned long long w = gw_wave_total_u64(v[k]);
                        if (lane == 0) atomicAdd(&s_acc[k], w);
                    }
                    const double wi = wave_total_f64(idm);
                    if (lane == 0) s_idm[wv] = wi;
                }
                __syncthreads();
                if (t == 0) {
                    double *a = s_sums[li * 4 + oi];
                    a[0] = (double)s_acc[0]; a[1] = (double)s_acc[1]; a[2] = (double)s_acc[2]; a[3] = (double)s_acc[3]; a[4] = (double)s_acc[4];
                    a[5] = (s_idm[0] + s_idm[1]) + (s_idm[2] + s_idm[3]); a[6] = (double)s_acc[5];
                    a[7] = (double)(long long)s_acc[6] / (double)GW_FIX; a[8] = (double)(long long)s_acc[7] / (double)GW_FIX; a[9] = (double)(long long)s_acc[8] / (double)GW_FIX;
                }
                __syncthreads();
            }
        }
        if (t < 16) { // combination t = level * 4 + offset, the column order of texture.rs:81-157
            double a[GL_NACC];
#pragma unroll
            for (int k = 0; k < GL_NACC; ++k) a[k] = s_sums[t][k];
            glcm_features<true>(a, 0.0, lut, out + i * (int64_t)F + col0 + t * 14, dbg ? dbg + (i * 16 + t) * 6 : nullptr);
        }
        __syncthreads();
    }
}

// ---- patch sizes 129..512: dense count tables in device memory ------------------------------------------------------------------
// Up to 2^18 mask pixels: neither a pixel list nor a hash table of that size fits the LDS, and with that many pairs a dense L x L table
// is no longer mostly zeros.  One 256-thread workgroup per nucleus owns a slice of scratch memory: a level map of the patch (one
// 16-bit word per pixel: level128 << 8 | level254, 0xFFFF outside the mask) and ONE dense table of 254 x 254 counters, reused by the
// sixteen (level, offset) combinations.  A pass: clear L^2 counters, every thread walks its pixels, finds the partner through the
// level map and bumps the unordered cell [lo][hi] with a device-memory atomic (they execute in L2; the table is cleared and read
// with agent-scope stores and loads, which bypass this CU's L1), the marginal and the sum histogram with LDS atomics, the pair
// sums in registers; then a sweep over the table as in k_glcm_wide (sum G^2 and the entropy sums from the FINAL counts, in fixed
// point: the same bits whatever order the atomics landed in).  A correctness path: no claim to speed.
static constexpr int GS_GRID = 256;          // workgroups (= scratch slices) of the launch
static constexpr int GS_TAB = 254 * 254;     // counters per slice
static size_t glcm_slab_bytes(int64_t n, int ps)
{
    const size_t g = (size_t)(n < GS_GRID ? (n > 0 ? n : 1) : GS_GRID);
    return g * (sizeof(unsigned) * GS_TAB + sizeof(unsigned short) * (size_t)ps * ps) + 512;
}

template <typename T>
__global__ __launch_bounds__(256) void k_glcm_slab(int64_t n, int ps, const T *__restrict__ patches, const uint64_t *__restrict__ mask_rows,
                                                   const GlcmLuts *__restrict__ lut, unsigned *__restrict__ tab_all, unsigned short *__restrict__ lvl_all,
                                                   float *__restrict__ out, int F, int col0, int64_t *__restrict__ dbg)
{
    __shared__ unsigned s_hx[256], s_hs[512];
    __shared__ unsigned long long s_acc[10];             // np, S1, S2, D1, D2, asmn | ents, hxs, hss (fixed point) | spare
    __shared__ double s_idm[4];                          // per wave: added in wave order
    __shared__ double s_sums[16][GL_NACC];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int P = ps * ps, W = (ps + 63) >> 6;
    unsigned *const tab = tab_all + (size_t)blockIdx.x * GS_TAB;
    unsigned short *const lvl = lvl_all + (size_t)blockIdx.x * P;
    for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
        __syncthreads(); // every thread is done with the previous nucleus's map and sums
        const T *pp = patches + (size_t)i * 3 * P;
        const uint64_t *rows = mask_rows + (size_t)i * ps * W;
        for (int p = t; p < P; p += 256) {
            const int r = p / ps, c = p - r * ps;
            unsigned code = 0xFFFFu;
            if ((rows[W * r + (c >> 6)] >> (c & 63)) & 1ull) {
                float g;
                if (sizeof(T) == 1) {
                    const uint8_t *pb = reinterpret_cast<const uint8_t *>(pp);
                    float gs = u8_unit(pb[p]) + u8_unit(pb[P + p]); // utils.rs:172
                    gs = gs + u8_unit(pb[2 * P + p]);
                    g = third_of_u8_sum(gs);                          // texture.rs:36
                } else {
                    g = gray_at<T>(pp, P, p);
                }
                code = (quant(g, 128) << 8) | quant(g, 254);          // texture.rs:19 (level128 <= 127: never 0xFFFF)
            }
            lvl[p] = (unsigned short)code;
        }
        __syncthreads();
#pragma unroll 1
        for (int li = 0; li < 4; ++li) {
            const int L = li == 3 ? 254 : (32 << li);
#pragma unroll 1
            for (int oi = 0; oi < 4; ++oi) {
                const int dy = kGlcmOffsets[oi][0], dx = kGlcmOffsets[oi][1];
                for (int k = t; k < L * L; k += 256) __hip_atomic_store(&tab[k], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_hx[t] = 0u; s_hs[t] = 0u; s_hs[256 + t] = 0u;
                if (t < 10) s_acc[t] = 0ull;
                __syncthreads(); // (workgroup release / acquire: the cleared counters are in L2 before the first atomic reaches it)
                unsigned long long np = 0, S1 = 0, S2 = 0, D1 = 0, D2 = 0;
                double idm = 0.0;
                for (int p = t; p < P; p += 256) {
                    const unsigned qa = lvl[p];
                    if (qa == 0xFFFFu) continue;
                    const int r = p / ps, c = p - r * ps;
                    const int r2 = r + dy, c2 = c + dx;
                    if ((unsigned)r2 >= (unsigned)ps || (unsigned)c2 >= (unsigned)ps) continue;
                    const unsigned qb = lvl[r2 * ps + c2];
                    if (qb == 0xFFFFu) continue;
                    unsigned a, b;
                    if (li == 3) { a = qa & 255u; b = qb & 255u; }
                    else { a = (qa >> 8) >> (2 - li); b = (qb >> 8) >> (2 - li); } // floor(128 g) >> k = floor(2^(7-k) g)
                    const unsigned lo = a < b ? a : b, hi = a < b ? b : a, d = hi - lo, sm = a + b;
                    np += 1; S1 += sm; S2 += (unsigned long long)sm * sm; D1 += d; D2 += (unsigned long long)d * d;
                    idm += 1.0 / (1.0 + (double)(d * d));
                    atomicAdd(&s_hx[a], 1u);
                    atomicAdd(&s_hx[b], 1u);
                    atomicAdd(&s_hs[sm], 1u);
                    __hip_atomic_fetch_add(&tab[lo * L + hi], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads(); // every wave's atomics have been performed (its vmcnt has drained) before any counter is read back
                // ---- sweep: sum G^2 and sum G ln G over the symmetric matrix from the final counts; marginal and sum entropies ----
                unsigned long long asmn = 0;
                long long ents = 0, hxs = 0, hss = 0;
                for (int k = t; k < L * L; k += 256) {
                    const unsigned cnt = __hip_atomic_load(&tab[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (!cnt) continue;
                    const int lo = k / L, hi = k - lo * L;
                    const bool diag = lo == hi;
                    const unsigned gsym = diag ? 2u * cnt : cnt;   // the symmetric matrix holds 2 cnt on the diagonal, cnt twice off it
                    asmn += (unsigned long long)gsym * gsym * (diag ? 1u : 2u);
                    ents += gw_xlnx_fix(lut, gsym) * (diag ? 1 : 2);
                }
                hxs += gw_xlnx_fix(lut, s_hx[t]);
                hss += gw_xlnx_fix(lut, s_hs[t]) + gw_xlnx_fix(lut, s_hs[256 + t]);
                {
                    unsigned long long v[9] = {np, S1, S2, D1, D2, asmn, (unsigned long long)ents, (unsigned long long)hxs, (unsigned long long)hss};
#pragma unroll
                    for (int k = 0; k < 9; ++k) {
                        const unsigned long long w = gw_wave_total_u64(v[k]);
                        if (lane == 0) atomicAdd(&s_acc[k], w);
                    }
                    const double wi = wave_total_f64(idm);
                    if (lane == 0) s_idm[wv] = wi;
                }
                __syncthreads();
                if (t == 0) {
                    double *a = s_sums[li * 4 + oi];
                    a[0] = (double)s_acc[0]; a[1] = (double)s_acc[1]; a[2] = (double)s_acc[2]; a[3] = (double)s_acc[3]; a[4] = (double)s_acc[4];
                    a[5] = (s_idm[0] + s_idm[1]) + (s_idm[2] + s_idm[3]); a[6] = (double)s_acc[5];
                    a[7] = (double)(long long)s_acc[6] / (double)GW_FIX; a[8] = (double)(long long)s_acc[7] / (double)GW_FIX; a[9] = (double)(long long)s_acc[8] / (double)GW_FIX;
                }
                __syncthreads();
            }
        }
        if (t < 16) { // combination t = level * 4 + offset, the column order of texture.rs:81-157
            double a[GL_NACC];
#pragma unroll
            for (int k = 0; k < GL_NACC; ++k) a[k] = s_sums[t][k];
            glcm_features<true>(a, 0.0, lut, out + i * (int64_t)F + col0 + t * 14, dbg ? dbg + (i * 16 + t) * 6 : nullptr);
        }
    }
}

#ifndef NFX_GL_ROUNDS
#define NFX_GL_ROUNDS 2
#endif
static constexpr int GL_MAX_GRID = 256 * 12 * NFX_GL_ROUNDS; // resident single-wave workgroups of the largest launch = parking slots
static size_t glcm_slots_bytes() { return sizeof(double) * 16 * GL_NACC * (size_t)GL_MAX_GRID; }
static constexpr int GW_SMALL_K = 6144; // masks up to this many pixels take the small-capacity launch of the wide kernel
size_t glcm_scratch_bytes(const Job &job)
{
    if (job.ps > NFX_WIDE_PS) return glcm_slab_bytes(job.n, job.ps); // k_glcm_slab: count tables and level maps
    if (job.ps > 64) return 256; // k_glcm_wide: the two launches' cursors
    const size_t n = (size_t)(job.n > 0 ? job.n : 1);
    return glcm_slots_bytes() + sizeof(int) * (GL_CLASSES * n + 64);
}

hipError_t prepare_glcm(nfx_ctx *ctx)
{
    if (ctx->glcm_lut) return hipSuccess;
    if (tables_locked(ctx)) return hipErrorNotReady; // device-memory calls on a reserved context and captures never allocate: nfx_ctx_reserve builds the tables
    std::vector<GlcmLuts> hv(1);
    GlcmLuts &h = hv[0];
    auto xl = [](double x) { return x > 0.0 ? x * std::log(x) : 0.0; };
    for (int k = 0; k < GL_RANKS; ++k) {
        h.dO[k] = 2.0 * (xl(k + 1.0) - xl((double)k));                    // dO
        h.dD[k] = xl(2.0 * (k + 1.0)) - xl(2.0 * (double)k);              // dD: H(x) = 2x ln 2x
    }
    for (int k = 0; k < GL_COUNTS; ++k) h.xlnx[k] = xl((double)k);        // x ln x
    for (int w = 0, hi = 0; w < GL_ROWEND; ++w) {                          // row of cell 2w: the largest hi with hi (hi + 1) / 2 <= 2w
        while ((hi + 1) * (hi + 2) / 2 <= 2 * w) ++hi;
        h.rowend[w] = (unsigned short)(hi * (hi + 3) / 2);
    }
    static_assert(offsetof(GlcmLuts, xlnx) == sizeof(double) * 2 * GL_RANKS && offsetof(GlcmLuts, rowend) == sizeof(double) * (2 * GL_RANKS + GL_COUNTS), "table layout");
    double *d = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d), sizeof(GlcmLuts));
    if (e != hipSuccess) return e;
    e = hipMemcpy(d, &h, sizeof(GlcmLuts), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d); return e; }
    ctx->glcm_lut = d; // published only once it is complete
    return hipSuccess;
}

// One launch per size class whose capacity the patch can reach.  The grid is four rounds of the workgroups a CU really
// holds (asked of the runtime once per instantiation: LDS allocation granularity and register budget included); blocks
// beyond a list's length exit at once.
template <typename T, int QCAP, int NPRE, bool FULL>
static void launch_class(nfx_ctx *ctx, const Job &job, const GlcmLuts *lut, double *slots, const int *lists, const int *counts, int cls, int top, int col0)
{
    // same answer for every context (one device type per process); contexts are driven from concurrent host threads, so the cached
    // answer is an atomic: two threads may both ask the runtime the first time, they store the same value
    static std::atomic<int> wgs_cached{0};
    int wgs_per_cu = wgs_cached.load(std::memory_order_relaxed);
    if (wgs_per_cu == 0) {
        int b = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, k_glcm<T, QCAP, NPRE, FULL>, 64, 0) != hipSuccess || b < 1) { (void)hipGetLastError(); b = 1; }
        wgs_per_cu = b > 12 ? 12 : b;
        wgs_cached.store(wgs_per_cu, std::memory_order_relaxed);
    }
    const int64_t m = 256ll * wgs_per_cu * NFX_GL_ROUNDS;
    static_assert(256 * 12 * NFX_GL_ROUNDS <= GL_MAX_GRID, "every launch's waves have a parking slot");
    const unsigned grid = (unsigned)(job.n < m ? job.n : m);
    // a nucleus that does not fit a typical-size table moves on to the next class's list (that launch has not started yet)
    int *next_list = FULL ? nullptr : const_cast<int *>(lists) + (int64_t)(cls + 1) * job.n;
    int *next_count = FULL ? nullptr : const_cast<int *>(counts) + cls + 1;
    // ... and one whose gray levels span more of the scale than the typical-size tables to the top class's (full-range structures)
    int *top_list = FULL ? nullptr : const_cast<int *>(lists) + (int64_t)top * job.n;
    int *top_count = FULL ? nullptr : const_cast<int *>(counts) + top;
    hipLaunchKernelGGL((k_glcm<T, QCAP, NPRE, FULL>), dim3(grid), dim3(64), 0, ctx->stream, lists + (int64_t)cls * job.n, counts + cls, const_cast<int *>(counts) + GL_CLASSES + cls,
                       next_list, next_count, top_list, top_count, job.ps,
                       reinterpret_cast<const T *>(job.patches), job.mask_rows, lut, slots, job.out, job.F, col0, job.dbg_glcm);
}

// At 64 x 64 the two smallest classes read their (at most 768) mask pixels where they lie instead of staging the 12 KB patch through LDS: the
// staging area no longer sets those classes' LDS size, there are no prefetch registers, and a CU holds twelve waves instead of ten
// (30.90 -> 30.51 ms per 600 k nuclei; a third unstaged class changes nothing).  Same arithmetic on the same bytes: identical rows.
#ifndef NFX_GL_NOSTAGE_CLS
#define NFX_GL_NOSTAGE_CLS 2
#endif
template <typename T, int NPRE>
static void launch_classes(nfx_ctx *ctx, const Job &job, const GlcmLuts *lut, double *slots, const int *lists, const int *counts, int col0, int only)
{
    const int top = glcm_class(job.ps * job.ps); // the class every mask of this patch size fits: its tables are full-size
    auto run = [&](int cls) { return cls <= top && (only < 0 || only == cls); };
#define NFX_GL_CLASS(CLS, QCAP)                                                                                         \
    if (run(CLS)) {                                                                                                     \
        if (CLS == top) launch_class<T, QCAP, NPRE, true>(ctx, job, lut, slots, lists, counts, CLS, top, col0);              \
        else launch_class<T, QCAP, (NPRE == 12 && CLS < NFX_GL_NOSTAGE_CLS) ? 0 : NPRE, false>(ctx, job, lut, slots, lists, counts, CLS, top, col0); \
    }
    NFX_GL_CLASS(0, 512)
    NFX_GL_CLASS(1, 768)
    NFX_GL_CLASS(2, 1024)
    NFX_GL_CLASS(3, 1536)
    NFX_GL_CLASS(4, 2048)
    NFX_GL_CLASS(5, 4096)
#undef NFX_GL_CLASS
}

hipError_t launch_glcm(nfx_ctx *ctx, const Job &job, int col0, void *scratch)
{
    if (job.n == 0) return hipSuccess;
    if (job.n > 0x7FFFFFFFll) return hipErrorInvalidValue; // class lists hold 32-bit nucleus indices
    hipError_t e = prepare_glcm(ctx);
    if (e != hipSuccess) return e;
    ScopedTiming t(ctx, KS_GLCM);
    if (job.ps > NFX_WIDE_PS) {
        const GlcmLuts *hlut = reinterpret_cast<const GlcmLuts *>(ctx->glcm_lut);
        const unsigned g = (unsigned)(job.n < GS_GRID ? job.n : GS_GRID);
        unsigned *tab = reinterpret_cast<unsigned *>(scratch);
        unsigned short *lvl = reinterpret_cast<unsigned short *>(tab + (size_t)g * GS_TAB);
        if (job.patches_f32)
            hipLaunchKernelGGL(k_glcm_slab<float>, dim3(g), dim3(256), 0, ctx->stream, job.n, job.ps, reinterpret_cast<const float *>(job.patches), job.mask_rows, hlut, tab, lvl,
                               job.out, job.F, col0, job.dbg_glcm);
        else
            hipLaunchKernelGGL(k_glcm_slab<uint8_t>, dim3(g), dim3(256), 0, ctx->stream, job.n, job.ps, reinterpret_cast<const uint8_t *>(job.patches), job.mask_rows, hlut, tab, lvl,
                               job.out, job.F, col0, job.dbg_glcm);
        return hipGetLastError();
    }
    if (job.ps > 64) {
        const GlcmLuts *wlut = reinterpret_cast<const GlcmLuts *>(ctx->glcm_lut);
        int *cursors = reinterpret_cast<int *>(scratch);
        e = hipMemsetAsync(cursors, 0, 2 * sizeof(int), ctx->stream);
        if (e != hipSuccess) return e;
        const unsigned g_small = (unsigned)(job.n < 256 * 2 ? job.n : 256 * 2), g_large = (unsigned)(job.n < 256 ? job.n : 256);
        if (job.patches_f32) {
            const float *pf = reinterpret_cast<const float *>(job.patches);
            hipLaunchKernelGGL((k_glcm_wide<float, GW_SMALL_K, 8192>), dim3(g_small), dim3(256), 0, ctx->stream, job.n, job.ps, pf, job.mask_rows, wlut, cursors, GW_SMALL_K, job.out, job.F, col0, job.dbg_glcm);
            hipLaunchKernelGGL((k_glcm_wide<float, 16384, 20480>), dim3(g_large), dim3(256), 0, ctx->stream, job.n, job.ps, pf, job.mask_rows, wlut, cursors + 1, GW_SMALL_K, job.out, job.F, col0, job.dbg_glcm);
        } else {
            const uint8_t *pb = reinterpret_cast<const uint8_t *>(job.patches);
            hipLaunchKernelGGL((k_glcm_wide<uint8_t, GW_SMALL_K, 8192>), dim3(g_small), dim3(256), 0, ctx->stream, job.n, job.ps, pb, job.mask_rows, wlut, cursors, GW_SMALL_K, job.out, job.F, col0, job.dbg_glcm);
            hipLaunchKernelGGL((k_glcm_wide<uint8_t, 16384, 20480>), dim3(g_large), dim3(256), 0, ctx->stream, job.n, job.ps, pb, job.mask_rows, wlut, cursors + 1, GW_SMALL_K, job.out, job.F, col0, job.dbg_glcm);
        }
        return hipGetLastError();
    }
    double *slots = reinterpret_cast<double *>(scratch);
    int *lists = reinterpret_cast<int *>(reinterpret_cast<char *>(scratch) + glcm_slots_bytes());
    int *counts = lists + GL_CLASSES * job.n;
    const GlcmLuts *lut = reinterpret_cast<const GlcmLuts *>(ctx->glcm_lut);
    e = hipMemsetAsync(counts, 0, sizeof(int) * 2 * GL_CLASSES, ctx->stream); // list lengths | list cursors
    if (e != hipSuccess) return e;
    // A call of a few hundred nuclei (one reference batch through the trait entry) is latency: six launches, each as long as
    // its slowest nucleus.  It runs as ONE launch of the class that holds every mask the patch size allows -- its waves all fit
    // the chip at once (3 per CU in the largest class).
    const int only = job.n <= 768 ? glcm_class(job.ps * job.ps) : -1;
    hipLaunchKernelGGL(k_glcm_classify, dim3((unsigned)((job.n + 63) / 64)), dim3(256), 0, ctx->stream, job.n, job.ps, job.mask_rows, job.areas, lists, counts, only);
    if (job.patches_f32) launch_classes<float, 1>(ctx, job, lut, slots, lists, counts, col0, only);
    else if (job.ps <= 32) launch_classes<uint8_t, 3>(ctx, job, lut, slots, lists, counts, col0, only);
    else if (job.ps <= 48) launch_classes<uint8_t, 7>(ctx, job, lut, slots, lists, counts, col0, only);
    else launch_classes<uint8_t, 12>(ctx, job, lut, slots, lists, counts, col0, only);
    return hipGetLastError();
}

}  // namespace nfx
