// exact_div.h -- correctly rounded division by the two constants of the pixel path in three instructions.
//
// The reference converts u8 pixels with `/ 255.0` (utils.rs:172) and forms gray with `/ 3.0` (texture.rs:36);
// both must be IEEE-exact here because quantisation thresholds sit on those values.  The compiler's IEEE division
// is an eleven-instruction sequence with a quarter-rate reciprocal.  For a constant divisor y with R = RN(1/y),
//     q = RN(x R);  r = fma(-y, q, x) (exact);  x / y = RN(q + r R)
// is correctly rounded (Markstein) unless y's significand is all ones; tests/test_exact_div.py additionally
// checks it exhaustively for every input the u8 path can produce (256 pixel values; 2^24 gray sums).
// Float patches (trait entry point) keep the ordinary division.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define NFX_HD __host__ __device__ __forceinline__
#else
#define NFX_HD inline
#endif

namespace nfx {

static NFX_HD float div_by_const(float x, float y, float ry)
{
    const float q = x * ry;
    const float r = fmaf(-y, q, x);
    return fmaf(r, ry, q);
}

// (float)u / 255.0f for u in 0..255
static NFX_HD float u8_unit(unsigned u) { return div_by_const((float)u, 255.0f, 1.0f / 255.0f); }

// (fl(fl(r + g) + b)) / 3.0f for r, g, b produced by u8_unit
static NFX_HD float third_of_u8_sum(float s) { return div_by_const(s, 3.0f, 1.0f / 3.0f); }

}  // namespace nfx
