// wave_ops.h -- wavefront-wide reductions on gfx950 with DPP (row_shr-free butterfly inside a row of 16,
// row_bcast15 / row_bcast31 across rows).  No LDS traffic, no address registers; the total is returned
// in every lane (read back from lane 63 through an SGPR).  Call only from wave-uniform control flow.
#pragma once
#include <hip/hip_runtime.h>

namespace nfx {

template <int CTRL, int ROW_MASK>
static __device__ __forceinline__ int dpp_i32(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, false);
}

static __device__ __forceinline__ int wave_total_i32(int v)
{
    v += dpp_i32<0xB1, 0xF>(v);  // quad_perm [1,0,3,2]
    v += dpp_i32<0x4E, 0xF>(v);  // quad_perm [2,3,0,1]
    v += dpp_i32<0x141, 0xF>(v); // row_half_mirror
    v += dpp_i32<0x140, 0xF>(v); // row_mirror: every lane holds its row's sum
    v += dpp_i32<0x142, 0xA>(v); // row_bcast15 into rows 1 and 3
    v += dpp_i32<0x143, 0xC>(v); // row_bcast31 into rows 2 and 3: lanes 48..63 hold the total
    return __builtin_amdgcn_readlane(v, 63);
}
static __device__ __forceinline__ unsigned wave_total_u32(unsigned v) { return (unsigned)wave_total_i32((int)v); }

// maximum of NON-NEGATIVE values over the wave (0, the fill value of the masked DPP steps, is their neutral element)
static __device__ __forceinline__ int wave_max_nonneg_i32(int v)
{
    v = max(v, dpp_i32<0xB1, 0xF>(v));
    v = max(v, dpp_i32<0x4E, 0xF>(v));
    v = max(v, dpp_i32<0x141, 0xF>(v));
    v = max(v, dpp_i32<0x140, 0xF>(v));
    v = max(v, dpp_i32<0x142, 0xA>(v));
    v = max(v, dpp_i32<0x143, 0xC>(v));
    return __builtin_amdgcn_readlane(v, 63);
}

// inclusive prefix sum over the 64 lanes, DPP only: Hillis-Steele inside each row of 16 (row_shr 1, 2, 4, 8 with zero
// fill), then the row totals carried across rows (row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3)
static __device__ __forceinline__ int wave_inclusive_scan_i32(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true); // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true); // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true); // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true); // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false); // row_bcast15
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false); // row_bcast31
    return v;
}

template <int CTRL, int ROW_MASK>
static __device__ __forceinline__ float dpp_f32(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
static __device__ __forceinline__ float wave_total_f32(float v)
{
    v += dpp_f32<0xB1, 0xF>(v);
    v += dpp_f32<0x4E, 0xF>(v);
    v += dpp_f32<0x141, 0xF>(v);
    v += dpp_f32<0x140, 0xF>(v);
    v += dpp_f32<0x142, 0xA>(v);
    v += dpp_f32<0x143, 0xC>(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

template <int CTRL, int ROW_MASK>
static __device__ __forceinline__ double dpp_f64(double v)
{
    const int lo = dpp_i32<CTRL, ROW_MASK>(__double2loint(v));
    const int hi = dpp_i32<CTRL, ROW_MASK>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
static __device__ __forceinline__ double wave_total_f64(double v)
{
    v += dpp_f64<0xB1, 0xF>(v);
    v += dpp_f64<0x4E, 0xF>(v);
    v += dpp_f64<0x141, 0xF>(v);
    v += dpp_f64<0x140, 0xF>(v);
    v += dpp_f64<0x142, 0xA>(v);
    v += dpp_f64<0x143, 0xC>(v);
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// sum within each aligned group of 16 lanes (a DPP "row"); every lane of the row gets the row's sum
static __device__ __forceinline__ double row16_total_f64(double v)
{
    v += dpp_f64<0xB1, 0xF>(v);
    v += dpp_f64<0x4E, 0xF>(v);
    v += dpp_f64<0x141, 0xF>(v);
    v += dpp_f64<0x140, 0xF>(v);
    return v;
}

// ---- many totals at once: reduce-scatter butterfly ------------------------------------------------------------------
// N (a power of two <= 16) values per lane are summed across the wave together: at step s lanes that differ in bit s
// exchange the half of the values they do not keep, so the number of values halves while the number of lanes summed
// doubles -- N-1 exchange-adds plus (6 - log2 N) ordinary steps, instead of 6 N.  Afterwards lane l holds the wave
// total of value scatter_index<N>(l & 15).
template <int MASK>
static __device__ __forceinline__ int xchg_i32(int v) // the value of the lane whose index differs in bit MASK (within a row of 16)
{
    if constexpr (MASK == 1) return dpp_i32<0xB1, 0xF>(v);       // quad_perm [1,0,3,2]
    else if constexpr (MASK == 2) return dpp_i32<0x4E, 0xF>(v);  // quad_perm [2,3,0,1]
    else if constexpr (MASK == 4) {                               // banks 0,2 read lane + 4 (row_shl:4), banks 1,3 lane - 4 (row_shr:4)
        const int t = __builtin_amdgcn_update_dpp(0, v, 0x104, 0xF, 0x5, false);
        return __builtin_amdgcn_update_dpp(t, v, 0x114, 0xF, 0xA, false);
    } else return dpp_i32<0x128, 0xF>(v);                         // row_ror:8
}
template <int MASK> static __device__ __forceinline__ unsigned xchg(unsigned v) { return (unsigned)xchg_i32<MASK>((int)v); }
template <int MASK> static __device__ __forceinline__ float xchg(float v) { return __int_as_float(xchg_i32<MASK>(__float_as_int(v))); }
template <int MASK> static __device__ __forceinline__ double xchg(double v)
{
    return __hiloint2double(xchg_i32<MASK>(__double2hiint(v)), xchg_i32<MASK>(__double2loint(v)));
}

// which of the N values a lane with l & 15 == j ends up holding
template <int N>
static __device__ __forceinline__ int scatter_index(int j)
{
    int idx = 0;
    if (N >= 2) idx += (j & 1) ? N / 2 : 0;
    if (N >= 4) idx += (j & 2) ? N / 4 : 0;
    if (N >= 8) idx += (j & 4) ? N / 8 : 0;
    if (N >= 16) idx += (j & 8) ? N / 16 : 0;
    return idx;
}

template <int MASK, int COUNT, typename V, int N>
static __device__ __forceinline__ void rs_step(V (&v)[N], int lane)
{
    if constexpr (COUNT > 1) { // halve: keep one half, send the other to the partner lane
        const bool up = (lane & MASK) != 0;
#pragma unroll
        for (int j = 0; j < COUNT / 2; ++j) {
            const V keep = up ? v[j + COUNT / 2] : v[j];
            const V send = up ? v[j] : v[j + COUNT / 2];
            v[j] = keep + xchg<MASK>(send);
        }
    } else {
        v[0] = v[0] + xchg<MASK>(v[0]);
    }
}

// lane-wise sums across the four rows of 16 (row_bcast would spread ONE lane of a row; here every lane of a row holds
// a different value): two cross-lane reads through the LDS crossbar per 32-bit word
static __device__ __forceinline__ unsigned across_rows(unsigned v)
{
    v += (unsigned)__shfl_xor((int)v, 16, 64);
    v += (unsigned)__shfl_xor((int)v, 32, 64);
    return v;
}
static __device__ __forceinline__ float across_rows(float v)
{
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
static __device__ __forceinline__ double across_rows(double v)
{
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// v is clobbered; every lane l ends up with the total of value scatter_index<N>(l & 15)
template <typename V, int N>
static __device__ __forceinline__ V wave_reduce_scatter(V (&v)[N], int lane)
{
    static_assert(N == 1 || N == 2 || N == 4 || N == 8 || N == 16, "power of two");
    rs_step<1, N>(v, lane);
    rs_step<2, (N >= 2 ? N / 2 : 1)>(v, lane);
    rs_step<4, (N >= 4 ? N / 4 : 1)>(v, lane);
    rs_step<8, (N >= 8 ? N / 8 : 1)>(v, lane);
    return across_rows(v[0]);
}

// LDS ops of one wave execute in order, so a wave-private producer/consumer hand-off through LDS needs no
// s_barrier -- only the compiler must not reorder the accesses.
static __device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace nfx
