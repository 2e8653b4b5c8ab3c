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

// LDS ops of one wave execute in order, so a wave-private producer/consumer hand-off through LDS needs no
// s_barrier -- only the compiler must not reorder the accesses.
static __device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace nfx
