// shm/sampling.h — sampler stream + warps + MIS helpers used by the path.
//
// Restates (paths relative to /root/reference/src):
//   sampler.rs:82-137      IndependentSampler: get_1d / get_2d (x then y) / get_pixel_2d
//   sampling.rs:187-194    power_heuristic
//   sampling.rs:237-262    linear_pdf / sample_linear / invert_linear_sample
//   sampling.rs:280-345    sample_uniform_sphere, uniform_*_pdf (quirk 2: hemisphere pdf = 1/(4pi)),
//                          sample_cosine_hemisphere, sample_uniform_disk_concentric/_polar
//   sampling.rs:373-408    sample_uniform_triangle, sample_bilinear, bilinear_pdf
//   sampling.rs:412-499    sample_spherical_triangle (quirk 3: `b1 / b1 + b2` precedence preserved)
//   sampling.rs:581-641    invert_spherical_triangle_sample
//   vecmath/spherical.rs:5-7  spherical_triangle_area
//
// Sampler stream (DEFINED here; see DESIGN.md "Sampler"): the reference's start_pixel_sample is a no-op
// (sampler.rs:117-121) over one SmallRng shared by whatever tiles a rayon worker steals, which cannot be
// replayed in parallel. Its own TODO (sampler.rs:14-17, PBR-4e p.469) names the deterministic form
// implemented here: an independent PCG32 stream per pixel, sequence = hash(pixel, seed), advanced to
// sample_index * 65536 + dimension. Uniform f32 = (u32 >> 8) * 2^-24, the mapping rand 0.8.5's
// `Standard` distribution uses for f32 (Cargo.lock:718; source not vendored: parity unpinned).
#pragma once
#include "vec.h"

namespace shm {

#ifdef SHM_ORACLE_REFERENCE_STREAM
// oracle/oracle.cpp only (test infrastructure; the product is never compiled with this): the reference's OWN sampler stream — one SmallRng (rand 0.8.5: Xoshiro256++
// seeded through SplitMix64) per worker thread, sampler.rs:103-131 — for orc_render_reference_stream. A sampler that carries one draws every dimension from it.
struct RefStream {
    uint64_t s[4];
};
inline uint64_t ref_splitmix64_next(uint64_t& state) {  // rand_core 0.6 / rand 0.8.5 xoshiro256plusplus.rs seed_from_u64: SplitMix64, the published constants
    state += 0x9e3779b97f4a7c15ULL;
    uint64_t z = state;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
inline RefStream ref_stream_seed_from_u64(uint64_t seed) {
    RefStream r;
    for (int i = 0; i < 4; ++i) r.s[i] = ref_splitmix64_next(seed);
    return r;
}
inline uint64_t ref_rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
inline uint64_t ref_stream_next_u64(RefStream& r) {  // xoshiro256++ 1.0 (Blackman / Vigna, public domain)
    const uint64_t result = ref_rotl(r.s[0] + r.s[3], 23) + r.s[0];
    const uint64_t t = r.s[1] << 17;
    r.s[2] ^= r.s[0];
    r.s[3] ^= r.s[1];
    r.s[1] ^= r.s[2];
    r.s[0] ^= r.s[3];
    r.s[2] ^= t;
    r.s[3] = ref_rotl(r.s[3], 45);
    return result;
}
// rng.gen::<f32>() of rand 0.8.5: Standard draws next_u32() (= next_u64() >> 32 for this generator) and keeps its upper 24 bits: (u32 >> 8) * 2^-24
inline float ref_stream_next_f32(RefStream& r) { return (float)(uint32_t)(ref_stream_next_u64(r) >> 40) * 5.9604644775390625e-8f; }
#endif

struct Rng {
    uint64_t state;
    uint64_t inc;
#ifdef SHM_ORACLE_REFERENCE_STREAM
    RefStream* ref = nullptr;
#endif
};
constexpr uint64_t PCG32_DEFAULT_STATE = 0x853c49e6748fea9bULL;
constexpr uint64_t PCG32_MULT = 0x5851f42d4c957f2dULL;

SHM_HD uint32_t rng_next_u32(Rng& r) {
    uint64_t old = r.state;
    r.state = old * PCG32_MULT + r.inc;
    uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
}
SHM_HD void rng_set_sequence(Rng& r, uint64_t sequence_index, uint64_t seed) {
    r.state = 0u;
    r.inc = (sequence_index << 1u) | 1u;
    rng_next_u32(r);
    r.state += seed;
    rng_next_u32(r);
}
SHM_HD void rng_advance(Rng& r, uint64_t delta) {
    uint64_t cur_mult = PCG32_MULT, cur_plus = r.inc, acc_mult = 1u, acc_plus = 0u;
    while (delta > 0) {
        if (delta & 1) {
            acc_mult *= cur_mult;
            acc_plus = acc_plus * cur_mult + cur_plus;
        }
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
        delta >>= 1;
    }
    r.state = acc_mult * r.state + acc_plus;
}
// rng_advance(r, n * 65536) with the first sixteen squarings folded into constants: after 16 doublings cur_mult is
// MULT^(2^16) and cur_plus is inc * prod_{j<16}(MULT^(2^j) + 1) (mod 2^64), so the loop only runs over the bits of n. Same
// state, bit for bit, as the generic routine (checked for 1000 random (state, inc, n) when the constants were derived and by
// tests/test_oracle_golden.py::test_sampler_stream_properties).
SHM_HD void rng_advance_65536(Rng& r, uint64_t n) {
    uint64_t cur_mult = 0x902da3ff53640001ULL, cur_plus = r.inc * 0x39f376e3016b0000ULL, acc_mult = 1u, acc_plus = 0u;
    while (n > 0) {
        if (n & 1) {
            acc_mult *= cur_mult;
            acc_plus = acc_plus * cur_mult + cur_plus;
        }
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
        n >>= 1;
    }
    r.state = acc_mult * r.state + acc_plus;
}
SHM_HD uint64_t mix_bits(uint64_t v) {
    v ^= (v >> 31);
    v *= 0x7fb5d329728ea185ULL;
    v ^= (v >> 27);
    v *= 0x81dadef4bc2dd44dULL;
    v ^= (v >> 33);
    return v;
}
// start_pixel_sample(p, sample_index, dimension = 0) as the reference's TODO describes it.
SHM_HD Rng sampler_start_pixel_sample(int px, int py, int sample_index, uint64_t seed) {
    uint64_t h = mix_bits(((uint64_t)(uint32_t)px << 32) | (uint64_t)(uint32_t)py);
    h = mix_bits(h ^ (seed + 0x9e3779b97f4a7c15ULL));
    Rng r;
    rng_set_sequence(r, h, PCG32_DEFAULT_STATE);
    rng_advance_65536(r, (uint64_t)(uint32_t)sample_index);
    return r;
}
SHM_HD Float sampler_get_1d(Rng& r) {
#ifdef SHM_ORACLE_REFERENCE_STREAM
    if (r.ref) return ref_stream_next_f32(*r.ref);
#endif
    return (Float)(rng_next_u32(r) >> 8) * 5.9604644775390625e-8f;
}
SHM_HD V2 sampler_get_2d(Rng& r) {  // sampler.rs:127-131: x drawn first
    Float x = sampler_get_1d(r);
    Float y = sampler_get_1d(r);
    return v2(x, y);
}

// sampling.rs:201-244 (not reached by the three integrators — the uniform light sampler picks by index — but part of the
// reference's tested sampling surface: sampling.rs:806-836). Returns the offset or -1 for an empty weight list.
SHM_HD int sample_discrete(const Float* weights, int n, Float u, Float* pmf, Float* u_remapped) {
    if (n == 0) {
        if (pmf) *pmf = 0.0f;
        return -1;
    }
    Float sum_weights = 0.0f;
    for (int i = 0; i < n; ++i) sum_weights += weights[i];
    Float up = u * sum_weights;
    if (up == sum_weights) up = next_float_down(up);
    int offset = 0;
    Float sum = 0.0f;
    while (sum + weights[offset] <= up) {
        sum += weights[offset];
        offset += 1;
    }
    if (pmf) *pmf = weights[offset] / sum_weights;
    if (u_remapped) *u_remapped = min((up - sum) / weights[offset], 1.0f - 1.1920929e-7f);
    return offset;
}
// sampling.rs:187-194
SHM_HD Float power_heuristic(int nf, Float f_pdf, int ng, Float g_pdf) {
    Float f = (Float)nf * f_pdf;
    Float g = (Float)ng * g_pdf;
    if (is_inf(sqr(f))) return 1.0f;
    return (f * f) / (f * f + g * g);
}
// sampling.rs:237-244
SHM_HD Float linear_pdf(Float x, Float a, Float b) {
    if (x < 0.0f || x > 1.0f) return 0.0f;
    return 2.0f * lerp(x, a, b) / (a + b);
}
// sampling.rs:246-253
SHM_HD Float sample_linear(Float u, Float a, Float b) {
    if (u == 0.0f && a == 0.0f) return 0.0f;
    Float x = u * (a + b) / (a + sqrt(lerp(u, a * a, b * b)));
    return min(x, 1.0f - FLOAT_EPSILON);
}
// sampling.rs:280-289
SHM_HD V3 sample_uniform_sphere(V2 u) {
    Float z = 1.0f - 2.0f * u.x;
    Float r = safe_sqrt(1.0f - z * z);
    Float phi = 2.0f * PI_F * u.y;
    return v3(r * cos(phi), r * sin(phi), z);
}
// sampling.rs:295-304
SHM_HD V3 sample_uniform_hemisphere(V2 u) {
    Float z = u.x;
    Float r = safe_sqrt(1.0f - z * z);
    Float phi = 2.0f * PI_F * u.y;
    return v3(r * cos(phi), r * sin(phi), z);
}
SHM_HD Float uniform_sphere_pdf() { return INV_4PI; }
SHM_HD Float uniform_hemisphere_pdf(bool strict = false) { return strict ? INV_2PI : INV_4PI; }  // sampling.rs:306-308 (quirk 2: 1/(4 pi) there)
// sampling.rs:324-339
SHM_HD V2 sample_uniform_disk_concentric(V2 u) {
    V2 uo = 2.0f * u - v2(1.0f, 1.0f);
    if (uo.x == 0.0f && uo.y == 0.0f) return v2(0.0f, 0.0f);
    Float r, theta;
    if (abs(uo.x) > abs(uo.y)) {
        r = uo.x;
        theta = PI_OVER_4 * (uo.y / uo.x);
    } else {
        r = uo.y;
        theta = PI_OVER_2 - PI_OVER_4 * (uo.x / uo.y);
    }
    return r * v2(cos(theta), sin(theta));
}
// sampling.rs:310-318
SHM_HD V3 sample_cosine_hemisphere(V2 u) {
    V2 d = sample_uniform_disk_concentric(u);
    Float z = safe_sqrt(1.0f - sqr(d.x) - sqr(d.y));
    return v3(d.x, d.y, z);
}
SHM_HD Float cosine_hemisphere_pdf(Float cos_theta) { return cos_theta * INV_PI; }  // sampling.rs:320-322
// sampling.rs:341-345
SHM_HD V2 sample_uniform_disk_polar(V2 u) {
    Float r = sqrt(u.x);
    Float theta = 2.0f * PI_F * u.y;
    return v2(r * cos(theta), r * sin(theta));
}
// sampling.rs:373-384
SHM_HD void sample_uniform_triangle(V2 u, Float& b0, Float& b1, Float& b2) {
    if (u.x < u.y) {
        b0 = u.x / 2.0f;
        b1 = u.y - b0;
    } else {
        b1 = u.y / 2.0f;
        b0 = u.x - b1;
    }
    b2 = 1.0f - b1 - b0;
}
// sampling.rs:386-391
SHM_HD V2 sample_bilinear(V2 u, const Float w[4]) {
    Float y = sample_linear(u.y, w[0] + w[1], w[2] + w[3]);
    Float x = sample_linear(u.x, lerp(y, w[0], w[2]), lerp(y, w[1], w[3]));
    return v2(x, y);
}
// sampling.rs:393-408
SHM_HD Float bilinear_pdf(V2 p, const Float w[4]) {
    if (p.x < 0.0f || p.x > 1.0f || p.y < 0.0f || p.y > 1.0f) return 0.0f;
    if (w[0] + w[1] + w[2] + w[3] == 0.0f) return 1.0f;
    return 4.0f
           * ((1.0f - p.x) * (1.0f - p.y) * w[0] + p.x * (1.0f - p.y) * w[1] + (1.0f - p.x) * p.y * w[2]
              + p.x * p.y * w[3])
           / (w[0] + w[1] + w[2] + w[3]);
}
// vecmath/spherical.rs:5-7
SHM_HD Float spherical_triangle_area(V3 a, V3 b, V3 c) {
    return abs(2.0f * atan2(dot(a, cross(b, c)), 1.0f + dot(a, b) + dot(a, c) + dot(b, c)));
}

// sampling.rs:412-499. Returns pdf; b[3] barycentrics. Two reference behaviours in the barycentrics of the sampled direction w, kept by default and replaced by PBRT-v4's
// forms with `strict` (ShmRenderParams::disable_reference_quirks): the divisor is e1 . e1 where PBRT-v4 has s1 . e1 (sampling.rs:477: found in round 6 by
// tests/test_light_sampling_properties.py / test_direct_lighting_analytic.py — the point handed back is then NOT where w meets the triangle, and pdf_with_context of the
// direction to it is not the sample's density), and `b1 / b1 + b2` (:493-497, quirk 3).
SHM_HD Float sample_spherical_triangle(const V3 v[3], V3 p, V2 u, Float b[3], bool strict = false) {
    V3 a = v[0] - p, bb = v[1] - p, c = v[2] - p;
    a = normalize(a);
    bb = normalize(bb);
    c = normalize(c);
    V3 n_ab = cross(a, bb), n_bc = cross(bb, c), n_ca = cross(c, a);
    if (length_squared(n_ab) == 0.0f || length_squared(n_bc) == 0.0f || length_squared(n_ca) == 0.0f) {
        b[0] = b[1] = b[2] = 0.0f;
        return 0.0f;
    }
    n_ab = normalize(n_ab);
    n_bc = normalize(n_bc);
    n_ca = normalize(n_ca);

    Float alpha = angle_between(n_ab, -n_ca);
    Float beta = angle_between(n_bc, -n_ab);
    Float gamma_ = angle_between(n_ca, -n_bc);

    Float a_pi = alpha + beta + gamma_;
    Float ap_pi = lerp(u.x, PI_F, a_pi);
    Float area = a_pi - PI_F;
    Float pdf = (area <= 0.0f) ? 0.0f : 1.0f / area;

    Float cos_alpha = cos(alpha);
    Float sin_alpha = sin(alpha);
    Float sin_phi = sin(ap_pi) * cos_alpha - cos(ap_pi) * sin_alpha;
    Float cos_phi = cos(ap_pi) * cos_alpha + sin(ap_pi) * sin_alpha;
    Float k1 = cos_phi + cos_alpha;
    Float k2 = sin_phi - sin_alpha * dot(a, bb);
    Float cos_bp = (k2 + (difference_of_products(k2, cos_phi, k1, sin_phi)) * cos_alpha)
                   / (sum_of_products(k2, sin_phi, k1, cos_phi) * sin_alpha);
    cos_bp = clamp(cos_bp, -1.0f, 1.0f);

    Float sin_bp = safe_sqrt(1.0f - cos_bp * cos_bp);
    V3 cp = cos_bp * a + sin_bp * normalize(gram_schmidt(c, a));

    Float cos_theta = 1.0f - u.y * (1.0f - dot(cp, bb));
    Float sin_theta = safe_sqrt(1.0f - cos_theta * cos_theta);
    V3 w = cos_theta * bb + sin_theta * normalize(gram_schmidt(cp, bb));

    V3 e1 = v[1] - v[0];
    V3 e2 = v[2] - v[0];
    V3 s1 = cross(w, e2);
    Float divisor = strict ? dot(s1, e1) : dot(e1, e1);  // (sic, sampling.rs:477)
    if (divisor == 0.0f) {
        b[0] = b[1] = b[2] = 1.0f / 3.0f;
        return pdf;
    }
    Float inv_divisor = 1.0f / divisor;
    V3 s = p - v[0];
    Float b1 = dot(s, s1) * inv_divisor;
    Float b2 = dot(w, cross(s, e1)) * inv_divisor;
    b1 = clamp(b1, 0.0f, 1.0f);
    b2 = clamp(b2, 0.0f, 1.0f);
    if (b1 + b2 > 1.0f) {
        if (strict) {  // PBRT-v4, statement by statement: b1 /= b1 + b2; b2 /= b1 + b2;
            b1 = b1 / (b1 + b2);
            b2 = b2 / (b1 + b2);
        } else {
            // sampling.rs:493-497, reference precedence: (b1 / b1) + b2 and (b2 / b1) + b2 (quirk 3)
            Float nb1 = b1 / b1 + b2;
            Float nb2 = b2 / b1 + b2;
            b1 = nb1;
            b2 = nb2;
        }
    }
    b[0] = 1.0f - b1 - b2;
    b[1] = b1;
    b[2] = b2;
    return pdf;
}

// sampling.rs:581-641
SHM_HD V2 invert_spherical_triangle_sample(const V3 v[3], V3 p, V3 w) {
    V3 a = normalize(v[0] - p), b = normalize(v[1] - p), c = normalize(v[2] - p);
    V3 n_ab = cross(a, b), n_bc = cross(b, c), n_ca = cross(c, a);
    if (length_squared(n_ab) == 0.0f || length_squared(n_bc) == 0.0f || length_squared(n_ca) == 0.0f)
        return v2(0.0f, 0.0f);
    n_ab = normalize(n_ab);
    n_bc = normalize(n_bc);
    n_ca = normalize(n_ca);
    Float alpha = angle_between(n_ab, -n_ca);
    Float beta = angle_between(n_bc, -n_ab);
    Float gamma_ = angle_between(n_ca, -n_bc);

    V3 cp = normalize(cross(cross(b, w), cross(c, a)));
    if (dot(cp, a + c) < 0.0f) cp = -cp;

    Float u0;
    if (dot(a, cp) > 0.99999847691f) {
        u0 = 0.0f;
    } else {
        V3 n_cpb = cross(cp, b);
        V3 n_acp = cross(a, cp);
        if (length_squared(n_cpb) == 0.0f || length_squared(n_acp) == 0.0f) return v2(0.5f, 0.5f);
        n_cpb = normalize(n_cpb);
        n_acp = normalize(n_acp);
        Float ap = alpha + angle_between(n_ab, n_cpb) + angle_between(n_acp, -n_cpb) - PI_F;
        Float area = alpha + beta + gamma_ - PI_F;
        u0 = ap / area;
    }
    Float u1 = (1.0f - dot(w, b)) / (1.0f - dot(cp, b));
    return v2(clamp(u0, 0.0f, 1.0f), clamp(u1, 0.0f, 1.0f));
}

}  // namespace shm
