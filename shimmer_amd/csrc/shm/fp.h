// shm/fp.h — scalar float kernel of the Shimmer hot path, single-source for HIP device code and host code.
//
// Restates (reference paths relative to /root/reference/src):
//   float.rs:1-138      Float=f32, MACHINE_EPSILON, bits, next_float_up/down, gamma(n), *_round_up/down
//   math.rs:13-31       INV_PI..., sqr
//   math.rs:160-193     difference_of_products / sum_of_products (Kahan, explicit FMA)
//   math.rs:246-282     lerp, radians, safe_asin, safe_acos (calls asin: reference quirk), safe_sqrt
//
// Bit-exactness contract: every function here uses only IEEE-754 correctly rounded operations
// (+ - * / sqrt fma, comparisons, integer/bit ops). Translation units including this header MUST be
// compiled with -ffp-contract=off and without fast-math so that host (gcc/clang x86-64) and device
// (hipcc gfx950, correctly rounded div/sqrt, denormals on) produce identical bits. Explicit FMAs of
// the reference (f32::mul_add) are written as shm::fma().
//
// Transcendentals (sin, cos, atan2, asin, exp, log, atanh, cosh, hypot) are implemented here from
// those primitive operations (Cody–Waite reduction + minimax polynomials after Cephes' single
// precision kernels, public domain algorithm descriptions) because libm (host) and OCML (device)
// differ in the last ulp; the reference itself calls Rust std (platform libm), so these agree with
// it to ~1-2 ulp, not bitwise — documented as "function-level parity" in DESIGN.md.
#pragma once

#include <stdint.h>
#include <math.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SHM_HD __host__ __device__ inline
// large leaf routines that are called from many sites (image texture filtering): a real call keeps code size and compile time bounded
#define SHM_HD_NOINLINE __host__ __device__ inline __attribute__((noinline))
#else
#define SHM_HD inline
#define SHM_HD_NOINLINE inline
#endif

namespace shm {

typedef float Float;

constexpr Float PI_F = 3.14159265358979323846f;          // float.rs:14 (std::f32::consts::PI)
constexpr Float INV_PI = 0.31830988618379067154f;        // math.rs:13
constexpr Float INV_2PI = 0.15915494309189533577f;       // math.rs:14
constexpr Float INV_4PI = 0.07957747154594766788f;       // math.rs:15
constexpr Float ONE_MINUS_EPSILON = 0.99999994f;         // next_float_down(1.0) = 0x3f7fffff (bxdf.rs:1019)
constexpr Float PI_OVER_4 = 0.78539816339744830961f;     // math.rs:16
constexpr Float PI_OVER_2 = 1.57079632679489661923f;     // math.rs:17
constexpr Float MACHINE_EPSILON = 1.1920928955078125e-7f * 0.5f;  // float.rs:19 (f32::EPSILON * 0.5)
constexpr Float FLOAT_EPSILON = 1.1920928955078125e-7f;  // f32::EPSILON
constexpr Float FLOAT_MAX = 3.40282346638528859812e+38f; // f32::MAX
constexpr Float FLOAT_MIN = -3.40282346638528859812e+38f;// f32::MIN (most negative; math.rs:141)

SHM_HD uint32_t float_to_bits(Float f) {  // float.rs:21-29
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    uint32_t u; memcpy(&u, &f, 4); return u;
#endif
}
SHM_HD Float bits_to_float(uint32_t u) {  // float.rs:31-39
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    Float f; memcpy(&f, &u, 4); return f;
#endif
}

SHM_HD Float infinity() { return bits_to_float(0x7f800000u); }
SHM_HD bool is_nan(Float x) { return x != x; }
SHM_HD bool is_inf(Float x) { return (float_to_bits(x) & 0x7fffffffu) == 0x7f800000u; }
SHM_HD bool is_finite(Float x) { return (float_to_bits(x) & 0x7f800000u) != 0x7f800000u; }

SHM_HD Float fma(Float a, Float b, Float c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fmaf(a, b, c);
#else
    return __builtin_fmaf(a, b, c);
#endif
}
SHM_HD double fma64(double a, double b, double c) { return __builtin_fma(a, b, c); }
SHM_HD Float sqrt(Float x) { return __builtin_sqrtf(x); }
SHM_HD Float abs(Float x) { return bits_to_float(float_to_bits(x) & 0x7fffffffu); }
SHM_HD Float copysign(Float mag, Float sgn) {
    return bits_to_float((float_to_bits(mag) & 0x7fffffffu) | (float_to_bits(sgn) & 0x80000000u));
}
SHM_HD Float floor(Float x) { return __builtin_floorf(x); }
SHM_HD Float trunc(Float x) { return __builtin_truncf(x); }
SHM_HD Float ceil(Float x) { return __builtin_ceilf(x); }

// Rust f32::max / f32::min (IEEE maxNum/minNum: a NaN operand is ignored).
SHM_HD Float max(Float a, Float b) { return is_nan(b) ? a : (is_nan(a) ? b : (a > b ? a : b)); }
SHM_HD Float min(Float a, Float b) { return is_nan(b) ? a : (is_nan(a) ? b : (a < b ? a : b)); }
// Rust f32::clamp: NaN stays NaN.
SHM_HD Float clamp(Float x, Float lo, Float hi) {
    Float r = x;
    if (r < lo) r = lo;
    if (r > hi) r = hi;
    return r;
}
// Rust f32::round: half away from zero.
SHM_HD Float round(Float x) {
    Float t = trunc(x);
    if (abs(x - t) >= 0.5f) t += copysign(1.0f, x);
    return t;
}

SHM_HD Float sqr(Float x) { return x * x; }  // math.rs:20-26

// float.rs:53-68
SHM_HD Float next_float_up(Float v) {
    if (is_inf(v) && v > 0.0f) return v;
    if (v == -0.0f) v = 0.0f;
    uint32_t ui = float_to_bits(v);
    if (v >= 0.0f) ui += 1; else ui -= 1;
    return bits_to_float(ui);
}
// float.rs:70-86
SHM_HD Float next_float_down(Float v) {
    if (is_inf(v) && v < 0.0f) return v;
    if (v == 0.0f) v = -0.0f;
    uint32_t ui = float_to_bits(v);
    if (v > 0.0f) ui -= 1; else ui += 1;
    return bits_to_float(ui);
}

// float.rs:88-90
SHM_HD constexpr Float gamma(int n) {
    return ((Float)n * MACHINE_EPSILON) / (1.0f - (Float)n * MACHINE_EPSILON);
}

// float.rs:92-138
SHM_HD Float add_round_up(Float a, Float b) { return next_float_up(a + b); }
SHM_HD Float add_round_down(Float a, Float b) { return next_float_down(a + b); }
SHM_HD Float sub_round_up(Float a, Float b) { return next_float_up(a - b); }
SHM_HD Float sub_round_down(Float a, Float b) { return next_float_down(a - b); }
SHM_HD Float mul_round_up(Float a, Float b) { return next_float_up(a * b); }
SHM_HD Float mul_round_down(Float a, Float b) { return next_float_down(a * b); }
SHM_HD Float div_round_up(Float a, Float b) { return next_float_up(a / b); }
SHM_HD Float div_round_down(Float a, Float b) { return next_float_down(a / b); }
SHM_HD Float sqrt_round_up(Float a) { return next_float_up(sqrt(a)); }
SHM_HD Float sqrt_round_down(Float a) { return next_float_down(sqrt(a)); }

// math.rs:170-182 (f32)
SHM_HD Float difference_of_products(Float a, Float b, Float c, Float d) {
    Float cd = c * d;
    Float difference = fma(a, b, -cd);
    Float error = fma(-c, d, cd);
    return difference + error;
}
SHM_HD Float sum_of_products(Float a, Float b, Float c, Float d) {
    return difference_of_products(a, b, -c, d);
}
// math.rs:184-196 (f64)
SHM_HD double difference_of_products64(double a, double b, double c, double d) {
    double cd = c * d;
    double difference = fma64(a, b, -cd);
    double error = fma64(-c, d, cd);
    return difference + error;
}

// math.rs:246-252: a * (1 - t) + b * t
SHM_HD Float lerp(Float t, Float a, Float b) { return a * (1.0f - t) + b * t; }

SHM_HD Float radians(Float deg) { return (PI_F / 180.0f) * deg; }  // math.rs:254-256

SHM_HD Float safe_sqrt(Float x) { return sqrt(max(0.0f, x)); }  // math.rs:276-279

// ---------------------------------------------------------------------------------------------
// Transcendentals built from correctly rounded primitives only (see header comment).
// ---------------------------------------------------------------------------------------------

// Shared reduction for sin/cos: x = k*(pi/2) + r, |r| <= pi/4 (for |x| < ~1e5; beyond that the result
// is still deterministic but loses accuracy, which the path never reaches: arguments are <= ~3*pi).
SHM_HD void sincos_reduce(Float x, Float& r, int& q) {
    const Float TWO_OVER_PI = 0.63661977236758134308f;
    const Float PIO2_HI = 1.5703125f;                 // 7 significant bits: k*HI exact for |k| < 2^17
    const Float PIO2_MID = 4.837512969970703125e-4f;  // next 11 bits
    const Float PIO2_LO = 7.54978995489188216e-8f;    // remainder
    Float kf = floor(x * TWO_OVER_PI + 0.5f);
    r = fma(-kf, PIO2_HI, x);
    r = fma(-kf, PIO2_MID, r);
    r = fma(-kf, PIO2_LO, r);
    q = (int)((long long)kf & 3);
}
SHM_HD Float sin_poly(Float r) {  // |r| <= pi/4
    Float z = r * r;
    Float p = -1.9515295891e-4f;
    p = fma(p, z, 8.3321608736e-3f);
    p = fma(p, z, -1.6666654611e-1f);
    return fma(p * z, r, r);
}
SHM_HD Float cos_poly(Float r) {  // |r| <= pi/4
    Float z = r * r;
    Float p = 2.443315711809948e-5f;
    p = fma(p, z, -1.388731625493765e-3f);
    p = fma(p, z, 4.166664568298827e-2f);
    return fma(p * z, z, fma(-0.5f, z, 1.0f));
}
SHM_HD Float sin(Float x) {
    if (!is_finite(x)) return x - x;
    Float r; int q;
    sincos_reduce(x, r, q);
    Float s = (q & 1) ? cos_poly(r) : sin_poly(r);
    return (q & 2) ? -s : s;
}
SHM_HD Float cos(Float x) {
    if (!is_finite(x)) return x - x;
    Float r; int q;
    sincos_reduce(x, r, q);
    Float c = (q & 1) ? sin_poly(r) : cos_poly(r);
    return ((q + 1) & 2) ? -c : c;
}

// atan for x >= 0 (Cephes atanf scheme).
SHM_HD Float atan_pos(Float x) {
    Float y;
    if (x > 2.414213562373095f) {         // tan(3pi/8)
        y = PI_OVER_2;
        x = -(1.0f / x);
    } else if (x > 0.4142135623730950f) { // tan(pi/8)
        y = PI_OVER_4;
        x = (x - 1.0f) / (x + 1.0f);
    } else {
        y = 0.0f;
    }
    Float z = x * x;
    Float p = 8.05374449538e-2f;
    p = fma(p, z, -1.38776856032e-1f);
    p = fma(p, z, 1.99777106478e-1f);
    p = fma(p, z, -3.33329491539e-1f);
    return y + fma(p * z, x, x);
}
SHM_HD Float atan(Float x) { return copysign(atan_pos(abs(x)), x); }
SHM_HD Float atan2(Float y, Float x) {
    if (is_nan(x) || is_nan(y)) return x + y;
    if (y == 0.0f) {
        bool xneg = (float_to_bits(x) >> 31) != 0;
        return xneg ? copysign(PI_F, y) : copysign(0.0f, y);
    }
    if (x == 0.0f) return copysign(PI_OVER_2, y);
    if (is_inf(x)) {
        if (is_inf(y)) return copysign(x > 0.0f ? PI_OVER_4 : 3.0f * PI_OVER_4, y);
        return x > 0.0f ? copysign(0.0f, y) : copysign(PI_F, y);
    }
    if (is_inf(y)) return copysign(PI_OVER_2, y);
    Float a = atan_pos(abs(y / x));
    if (x < 0.0f) a = PI_F - a;
    return copysign(a, y);
}

// asin (Cephes asinf scheme); NaN for |x| > 1.
SHM_HD Float asin(Float x) {
    Float a = abs(x);
    if (a > 1.0f) return (x - x) / (x - x);
    Float z, w;
    bool big = a > 0.5f;
    if (big) {
        z = 0.5f * (1.0f - a);
        w = sqrt(z);
    } else {
        w = a;
        z = a * a;
    }
    Float p = 4.2163199048e-2f;
    p = fma(p, z, 2.4181311049e-2f);
    p = fma(p, z, 4.5470025998e-2f);
    p = fma(p, z, 7.4953002686e-2f);
    p = fma(p, z, 1.6666752422e-1f);
    Float r = fma(p * z, w, w);
    if (big) r = PI_OVER_2 - (r + r);
    return copysign(r, x);
}
SHM_HD Float acos(Float x) {  // host-side helpers only (Sphere::new); not on the device path
    Float a = abs(x);
    if (a > 1.0f) return (x - x) / (x - x);
    if (x > 0.5f) return 2.0f * asin(sqrt(0.5f * (1.0f - x)));
    if (x < -0.5f) return PI_F - 2.0f * asin(sqrt(0.5f * (1.0f + x)));
    return PI_OVER_2 - asin(x);
}

// math.rs:266-274 — safe_acos calls asin in the reference (quirk 5); preserved.
SHM_HD Float safe_asin(Float x) { return asin(clamp(x, -1.0f, 1.0f)); }
SHM_HD Float safe_acos(Float x) { return asin(clamp(x, -1.0f, 1.0f)); }

// exp (Cephes expf scheme).
SHM_HD Float exp(Float x) {
    if (is_nan(x)) return x;
    if (x > 88.72283905206835f) return infinity();
    if (x < -103.972084045410f) return 0.0f;
    const Float LOG2E = 1.44269504088896341f;
    const Float C1 = 0.693359375f;
    const Float C2 = -2.12194440e-4f;
    Float n = floor(fma(LOG2E, x, 0.5f));
    Float r = fma(-n, C1, x);
    r = fma(-n, C2, r);
    Float z = r * r;
    Float p = 1.9875691500e-4f;
    p = fma(p, r, 1.3981999507e-3f);
    p = fma(p, r, 8.3334519073e-3f);
    p = fma(p, r, 4.1665795894e-2f);
    p = fma(p, r, 1.6666665459e-1f);
    p = fma(p, r, 5.0000001201e-1f);
    Float y = fma(p, z, r) + 1.0f;
    // scale by 2^n in two steps so that denormal results round once at the end at worst.
    int ni = (int)n;
    int n1 = ni / 2, n2 = ni - n1;
    Float s1 = bits_to_float((uint32_t)(n1 + 127) << 23);
    Float s2 = bits_to_float((uint32_t)(n2 + 127) << 23);
    return (y * s1) * s2;
}

// natural log for finite x > 0 (Cephes logf scheme).
SHM_HD Float log(Float x) {
    if (is_nan(x)) return x;
    if (x < 0.0f) return (x - x) / (x - x);
    if (x == 0.0f) return -infinity();
    if (is_inf(x)) return x;
    uint32_t ux = float_to_bits(x);
    int e = 0;
    if ((ux & 0x7f800000u) == 0) {  // denormal: scale up
        x = x * 8388608.0f;
        ux = float_to_bits(x);
        e = -23;
    }
    e += (int)(ux >> 23) - 126;
    Float m = bits_to_float((ux & 0x007fffffu) | 0x3f000000u);  // in [0.5, 1)
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = m + m - 1.0f;
    } else {
        m = m - 1.0f;
    }
    Float z = m * m;
    Float p = 7.0376836292e-2f;
    p = fma(p, m, -1.1514610310e-1f);
    p = fma(p, m, 1.1676998740e-1f);
    p = fma(p, m, -1.2420140846e-1f);
    p = fma(p, m, 1.4249322787e-1f);
    p = fma(p, m, -1.6668057665e-1f);
    p = fma(p, m, 2.0000714765e-1f);
    p = fma(p, m, -2.4999993993e-1f);
    p = fma(p, m, 3.3333331174e-1f);
    Float y = m * z * p;
    Float fe = (Float)e;
    y = fma(fe, -2.12194440e-4f, y);
    y = fma(-0.5f, z, y);
    Float r = m + y;
    return fma(fe, 0.693359375f, r);
}
// base-2 log (f32::log2 in the reference is the platform's; this one shares log()'s reduction and polynomial, and is exact
// for powers of two: m = 0 there). Used for MIP level selection (mipmap.rs:150, 163).
SHM_HD Float log2(Float x) {
    if (is_nan(x)) return x;
    if (x < 0.0f) return (x - x) / (x - x);
    if (x == 0.0f) return -infinity();
    if (is_inf(x)) return x;
    uint32_t ux = float_to_bits(x);
    int e = 0;
    if ((ux & 0x7f800000u) == 0) {
        x = x * 8388608.0f;
        ux = float_to_bits(x);
        e = -23;
    }
    e += (int)(ux >> 23) - 126;
    Float m = bits_to_float((ux & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = m + m - 1.0f;
    } else {
        m = m - 1.0f;
    }
    Float z = m * m;
    Float p = 7.0376836292e-2f;
    p = fma(p, m, -1.1514610310e-1f);
    p = fma(p, m, 1.1676998740e-1f);
    p = fma(p, m, -1.2420140846e-1f);
    p = fma(p, m, 1.4249322787e-1f);
    p = fma(p, m, -1.6668057665e-1f);
    p = fma(p, m, 2.0000714765e-1f);
    p = fma(p, m, -2.4999993993e-1f);
    p = fma(p, m, 3.3333331174e-1f);
    Float y = m * z * p;
    y = fma(-0.5f, z, y);
    // (m + y) * log2(e), with log2(e) - 1 = 0.44269504088896340735992 split off as Cephes log2f does
    Float r = y * 0.44269504088896340735992f;
    r = fma(m, 0.44269504088896340735992f, r);
    r += y;
    r += m;
    return r + (Float)e;
}
// log1p via the classic u = 1 + y correction.
SHM_HD Float log1p(Float y) {
    Float u = 1.0f + y;
    if (u == 1.0f) return y;
    return log(u) * (y / (u - 1.0f));
}
// Rust std f32::atanh (1.7x): 0.5 * ((2.0 * x) / (1.0 - x)).ln_1p()
SHM_HD Float atanh(Float x) { return 0.5f * log1p((2.0f * x) / (1.0f - x)); }
SHM_HD Float cosh(Float x) {
    Float e = exp(abs(x));
    return 0.5f * e + 0.5f / e;
}
// hypot through f64 (exact products, one rounding in the sum, correctly rounded sqrt, one final rounding).
SHM_HD Float hypot(Float x, Float y) {
    double dx = (double)x, dy = (double)y;
    return (Float)__builtin_sqrt(dx * dx + dy * dy);
}

}  // namespace shm
