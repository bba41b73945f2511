// pt_math.h -- the one floating-point evaluation every part of the renderer uses.
//
// WGSL leaves fusing and the accuracy of sin/cos/tan/pow/normalize to the implementation,
// and the shader's fract(sin(x)*43758.5453) hash (shaders/include/utils.wgsl:3-5) turns one
// ulp of difference into an unrelated random number.  Results are therefore reproducible
// only against one fixed evaluation, which this header pins down for host and device:
//   * IEEE-754 binary32, round-to-nearest-even, subnormals kept, no fast-math;
//   * a fused multiply-add exactly where pt_fma is written and nowhere else (the translation
//     units are built with -ffp-contract=off);
//   * correctly rounded '/' and sqrt; sphere normal / radius, sphere roots / 2a, cube-face
//     coordinates / major axis and colour / spp are value * (1/denominator); min/max = minNum/maxNum;
//   * sin/cos by a 3-step fused Cody-Waite reduction and fixed minimax polynomials.
// Host code (camera frame, scene set-up) and gfx950 kernels compile this same text.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_HD __host__ __device__ __forceinline__

struct v3 {
    float x, y, z;
};

PT_HD float pt_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
PT_HD float pt_min(float a, float b) { return __builtin_fminf(a, b); }
PT_HD float pt_max(float a, float b) { return __builtin_fmaxf(a, b); }
PT_HD float pt_abs(float a) { return __builtin_fabsf(a); }
// Correctly rounded 1/x and sqrt(x), written so that the kernels pay less than the compiler's general sequences
// (11 and 14 instructions) for ordinary operands: v_rcp_f32 and one Newton step are the correctly rounded reciprocal
// for every operand with binary exponent 1..252, v_sqrt_f32 followed by a test of its two neighbours is the correctly
// rounded root for every finite operand >= 2^-100 (both checked over all 2^32 bit patterns on the device:
// tools/ubench/exact_math.hip).  Anything else -- zeros, subnormals, infinities, NaNs, reciprocals that come out
// subnormal -- takes the general sequence, which whole waves skip when no lane needs it.  The host evaluates the same
// functions with the C operators: same values.
PT_HD float pt_rcp(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // (one wave-uniform branch: if any active lane needs the general sequence, all take it)
    const uint32_t ex = __float_as_uint(x) & 0x7F800000u;
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(ex - 0x00800000u < 0x7E000000u)) == 0ull, 1)) {
        const float r = __builtin_amdgcn_rcpf(x);
        return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
    }
    asm volatile("" ::: "memory");  // keeps the compiler from evaluating the rare branch for everybody and selecting
#endif
    return 1.0f / x;
}
PT_HD float pt_sqrt(float a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(__float_as_uint(a) - 0x0D800000u < 0x7F800000u - 0x0D800000u)) == 0ull, 1)) {
        const float s = __builtin_amdgcn_sqrtf(a);
        const float lo = __uint_as_float(__float_as_uint(s) - 1u), hi = __uint_as_float(__float_as_uint(s) + 1u);
        float r = __builtin_fmaf(-lo, s, a) <= 0.0f ? lo : s;
        r = __builtin_fmaf(-hi, s, a) > 0.0f ? hi : r;
        return r;
    }
    asm volatile("" ::: "memory");
#endif
    return __builtin_sqrtf(a);
}
PT_HD float pt_floor(float a) { return __builtin_floorf(a); }
PT_HD float pt_rint(float a) { return __builtin_rintf(a); }

PT_HD v3 V3(float x, float y, float z) { return v3{x, y, z}; }
PT_HD v3 operator+(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_HD v3 operator-(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_HD v3 operator*(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_HD v3 operator*(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
PT_HD v3 operator/(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
PT_HD v3 operator-(v3 a) { return V3(-a.x, -a.y, -a.z); }
// s*a + c, one rounding per component
PT_HD v3 v3_fma(float s, v3 a, v3 c) { return V3(pt_fma(s, a.x, c.x), pt_fma(s, a.y, c.y), pt_fma(s, a.z, c.z)); }
PT_HD float dot3(v3 a, v3 b) { return pt_fma(a.z, b.z, pt_fma(a.y, b.y, a.x * b.x)); }
PT_HD v3 cross3(v3 a, v3 b)
{
    return V3(pt_fma(a.y, b.z, -(a.z * b.y)), pt_fma(a.z, b.x, -(a.x * b.z)), pt_fma(a.x, b.y, -(a.y * b.x)));
}
PT_HD v3 normalize3(v3 a)
{
    const float inv = pt_rcp(pt_sqrt(dot3(a, a)));
    return a * inv;
}
// (1/x, 1/y, 1/z): one range test for the three
PT_HD v3 pt_rcp3(v3 d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t ex = __float_as_uint(d.x) & 0x7F800000u, ey = __float_as_uint(d.y) & 0x7F800000u, ez = __float_as_uint(d.z) & 0x7F800000u;
    const uint32_t lo = min(min(ex, ey), ez), hi = max(max(ex, ey), ez);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(lo >= 0x00800000u && hi < 0x7E800000u)) == 0ull, 1)) {
        const float rx = __builtin_amdgcn_rcpf(d.x), ry = __builtin_amdgcn_rcpf(d.y), rz = __builtin_amdgcn_rcpf(d.z);
        return V3(__builtin_fmaf(__builtin_fmaf(-d.x, rx, 1.0f), rx, rx), __builtin_fmaf(__builtin_fmaf(-d.y, ry, 1.0f), ry, ry),
                  __builtin_fmaf(__builtin_fmaf(-d.z, rz, 1.0f), rz, rz));
    }
    asm volatile("" ::: "memory");
#endif
    return V3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
}

// sin(x) and cos(x).  k = nearest integer to x*2/pi; r = x - k*pi/2 in three fused steps
// (pi/2 split hi/mid/lo); on [-pi/4, pi/4]
//   sin r = fma(s*r, fma(s, fma(s, S3, S2), S1), r)             s = r*r
//   cos r = fma(s*s, fma(s, fma(s, C3, C2), C1), fma(-0.5, s, 1))
// and the quadrant (k mod 4) picks the polynomial and the sign.  Both have the same Horner
// shape, so one evaluation with per-lane coefficients serves either.
struct pt_reduced {
    float r, s;
    int q;
};
PT_HD pt_reduced pt_reduce_pio2(float x)
{
    const float k = pt_rint(x * 0.63661977236758134f);
    float r = pt_fma(-k, 1.57079637050628662f, x);
    r = pt_fma(-k, -4.37113882867379e-08f, r);
    r = pt_fma(-k, -1.71512451000588e-15f, r);
    pt_reduced o;
    o.r = r;
    o.s = r * r;
    o.q = (k >= -2147483648.0f && k < 2147483648.0f) ? (int)k : 0;
    return o;
}
// use_cos selects the cosine polynomial; the caller applies the sign
PT_HD float pt_sincos_poly(const pt_reduced &a, bool use_cos)
{
    const float c3 = use_cos ? 2.443315711809948e-5f : -1.9515295891e-4f;
    const float c2 = use_cos ? -1.388731625493765e-3f : 8.3321608736e-3f;
    const float c1 = use_cos ? 4.166664568298827e-2f : -1.6666654611e-1f;
    float p = pt_fma(a.s, c3, c2);
    p = pt_fma(a.s, p, c1);
    const float m = a.s * (use_cos ? a.s : a.r);
    const float base = use_cos ? pt_fma(-0.5f, a.s, 1.0f) : a.r;
    return pt_fma(m, p, base);
}
PT_HD float pt_sin(float x)
{
    const pt_reduced a = pt_reduce_pio2(x);
    const float v = pt_sincos_poly(a, (a.q & 1) != 0);
    return (a.q & 2) ? -v : v;
}
PT_HD float pt_cos(float x)
{
    const pt_reduced a = pt_reduce_pio2(x);
    const float v = pt_sincos_poly(a, (a.q & 1) == 0);
    return ((a.q + 1) & 2) ? -v : v;
}
PT_HD void pt_sincos(float x, float &s_out, float &c_out)
{
    const pt_reduced a = pt_reduce_pio2(x);
    const float ps = pt_sincos_poly(a, false), pc = pt_sincos_poly(a, true);
    const float sv = (a.q & 1) ? pc : ps;
    const float cv = (a.q & 1) ? ps : pc;
    s_out = (a.q & 2) ? -sv : sv;
    c_out = ((a.q + 1) & 2) ? -cv : cv;
}
PT_HD float pt_tan(float x)
{
    float s, c;
    pt_sincos(x, s, c);
    return s / c;
}

// atan on [0,1] (odd polynomial), octant fix-up; used only by the equirect->cube kernel.
PT_HD float pt_atan_unit(float a)
{
    const float s = a * a;
    float p = pt_fma(s, 0.00282363896258175373077393f, -0.0159569028764963150024414f);
    p = pt_fma(s, p, 0.0425049886107444763183594f);
    p = pt_fma(s, p, -0.0748900920152664184570312f);
    p = pt_fma(s, p, 0.106347933411598205566406f);
    p = pt_fma(s, p, -0.142027363181114196777344f);
    p = pt_fma(s, p, 0.199926957488059997558594f);
    p = pt_fma(s, p, -0.333331018686294555664062f);
    return pt_fma(s * a, p, a);
}
PT_HD float pt_atan2(float y, float x)
{
    const float ax = pt_abs(x), ay = pt_abs(y);
    const float mx = pt_max(ax, ay), mn = pt_min(ax, ay);
    float a = (mx == 0.0f) ? 0.0f : pt_atan_unit(mn / mx);
    if (ay > ax) a = 1.57079637050628662f - a;
    if (x < 0.0f) a = 3.14159274101257324f - a;
    return (y < 0.0f) ? -a : a;
}
PT_HD float pt_asin(float x) { return pt_atan2(x, pt_sqrt((1.0f - x) * (1.0f + x))); }

PT_HD float pt_fract(float x) { return x - pt_floor(x); }

// rand(): shaders/include/utils.wgsl:3-5
PT_HD float pt_hash_rand(float sx, float sy)
{
    const float d = pt_fma(sy, 78.233f, sx * 12.9898f);
    return pt_fract(pt_sin(d) * 43758.5453f);
}

// randomUnit(): shaders/include/utils.wgsl:15-22
PT_HD v3 pt_random_unit(float sx, float sy)
{
    const float z = pt_fma(pt_hash_rand(sx, sy), 2.0f, -1.0f);
    const float a = (pt_hash_rand(sx * z, sy * z) * 2.0f) * 3.14159274101257324f;
    const float r = pt_sqrt(pt_fma(-z, z, 1.0f));
    float sa, ca;
    pt_sincos(a, sa, ca);
    return V3(r * ca, r * sa, z);
}

// hybridTaus(): shaders/include/utils.wgsl:33-58.  Advances st, returns the value in [0,1].
PT_HD uint32_t pt_tau_step(uint32_t z, int s1, int s2, int s3, uint32_t m)
{
    const uint32_t b = ((z << s1) ^ z) >> s2;
    return ((z & m) << s3) ^ b;
}
PT_HD float pt_hybrid_taus(uint32_t &sx, uint32_t &sy, uint32_t &sz, uint32_t &sw)
{
    sx = pt_tau_step(sx, 13, 19, 12, 4294967294u);
    sy = pt_tau_step(sy, 2, 25, 4, 4294967288u);
    sz = pt_tau_step(sz, 3, 11, 17, 4294967280u);
    sw = 1664525u * sw + 1013904223u;
    return 2.3283064365387e-10f * (float)(sx ^ sy ^ sz ^ sw);
}

// Camera frame shared by every pixel: shaders/compute.wgsl:93-107.
struct pt_frame {
    v3 origin, pixel00, du, dv;
};
PT_HD pt_frame pt_camera_frame(const float cam[16], uint32_t W, uint32_t H)
{
    const v3 origin = V3(cam[0], cam[1], cam[2]);
    const float focal = cam[3];
    const v3 forward = V3(cam[4], cam[5], cam[6]);
    const float vfov = cam[7];
    const v3 right = V3(cam[8], cam[9], cam[10]);
    const v3 up = V3(cam[12], cam[13], cam[14]);
    const float fw = (float)W, fh = (float)H;
    const float aspect = fw / fh;
    const float theta = vfov * 0.017453292519943295f;
    const float h = pt_tan(theta / 2.0f);
    const float vph = (2.0f * h) * focal;
    const float vpw = aspect * vph;
    const v3 U = right * vpw;
    const v3 Vv = up * (-vph);
    pt_frame f;
    f.du = U / fw;
    f.dv = Vv / fh;
    f.origin = origin;
    const v3 ul = v3_fma(focal, forward, origin) - (U + Vv) * 0.5f;
    f.pixel00 = ul + (f.du + f.dv) * 0.5f;
    return f;
}
