// shm/bxdf.h — scattering helpers, Trowbridge–Reitz, the four contracted BxDFs and the BSDF wrapper.
//
// Restates (paths relative to /root/reference/src):
//   vecmath/spherical.rs:28-92   cos_theta, cos2_theta, abs_cos_theta, sin2_theta, sin_theta, tan2_theta,
//                                cos_phi, sin_phi, same_hemisphere
//   scattering.rs:12-43          reflect, refract
//   scattering.rs:49-104         fresnel_dielectric, fresnel_complex (num-complex 0.4.4 arithmetic
//                                restated from its published definitions: parity unpinned), _spectral
//   scattering.rs:107-220        TrowbridgeReitzDistribution::{new,effectively_smooth,d,g1,lambda,g,d_w,pdf,
//                                sample_wm,roughness_to_alpha,regularize}
//   bxdf.rs:184-267              DiffuseBxDF
//   bxdf.rs:328-458              ConductorBxDF
//   bxdf.rs:518-795              DielectricBxDF
//   bxdf.rs:797-881              ThinDielectricBxDF
//   bxdf.rs:1702-1829            BSDFSample, BxDFReflTransFlags, BxDFFLags
//   bsdf.rs:22-111               BSDF::{new,f,sample_f,pdf,flags,regularize}
#pragma once
#include "sampling.h"
#include "spectrum.h"

namespace shm {

// spherical.rs
SHM_HD Float cos_theta(V3 w) { return w.z; }
SHM_HD Float cos2_theta(V3 w) { return w.z * w.z; }
SHM_HD Float abs_cos_theta(V3 w) { return abs(w.z); }
SHM_HD Float sin2_theta(V3 w) { return max(0.0f, 1.0f - cos2_theta(w)); }
SHM_HD Float sin_theta(V3 w) { return sqrt(sin2_theta(w)); }
SHM_HD Float tan2_theta(V3 w) { return sin2_theta(w) / cos2_theta(w); }
SHM_HD Float cos_phi(V3 w) {
    Float st = sin_theta(w);
    return (st == 0.0f) ? 1.0f : clamp(w.x / st, -1.0f, 1.0f);
}
SHM_HD Float sin_phi(V3 w) {
    Float st = sin_theta(w);
    return (st == 0.0f) ? 1.0f : clamp(w.y / st, -1.0f, 1.0f);  // spherical.rs:71-78 returns 1.0 here too
}
SHM_HD bool same_hemisphere(V3 w, V3 wp) { return w.z * wp.z > 0.0f; }

// scattering.rs:12-14
SHM_HD V3 reflect(V3 wo, V3 n) { return -wo + 2.0f * dot(wo, n) * n; }
// scattering.rs:21-43
SHM_HD bool refract(V3 wi, V3 n, Float eta, V3& wt, Float& etap) {
    Float cos_theta_i = dot(n, wi);
    if (cos_theta_i < 0.0f) {
        eta = 1.0f / eta;
        cos_theta_i = -cos_theta_i;
        n = -n;
    }
    Float sin2_theta_i = max(0.0f, 1.0f - sqr(cos_theta_i));
    Float sin2_theta_t = sin2_theta_i / sqr(eta);
    if (sin2_theta_t >= 1.0f) return false;
    Float cos_theta_t = sqrt(1.0f - sin2_theta_t);
    wt = -wi / eta + (cos_theta_i / eta - cos_theta_t) * n;
    etap = eta;
    return true;
}
// scattering.rs:49-70
SHM_HD Float fresnel_dielectric(Float cos_theta_i, Float eta) {
    cos_theta_i = clamp(cos_theta_i, -1.0f, 1.0f);
    if (cos_theta_i < 0.0f) {
        eta = 1.0f / eta;
        cos_theta_i = -cos_theta_i;
    }
    Float sin2_theta_i = 1.0f - cos_theta_i * cos_theta_i;
    Float sin2_theta_t = sin2_theta_i / (eta * eta);
    if (sin2_theta_t >= 1.0f) return 1.0f;
    Float cos_theta_t = safe_sqrt(1.0f - sin2_theta_t);
    Float r_parl = (eta * cos_theta_i - cos_theta_t) / (eta * cos_theta_i + cos_theta_t);
    Float r_perp = (cos_theta_i - eta * cos_theta_t) / (cos_theta_i + eta * cos_theta_t);
    return 0.5f * (r_parl * r_parl + r_perp * r_perp);
}

// num-complex 0.4.4 Complex<f32> (Cargo.lock; not vendored) — operations as that crate defines them.
struct Cx { Float re, im; };
SHM_HD Cx cx(Float re, Float im) { Cx c; c.re = re; c.im = im; return c; }
SHM_HD Cx cx_mul(Cx a, Cx b) { return cx(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
SHM_HD Float cx_norm_sqr(Cx a) { return a.re * a.re + a.im * a.im; }
SHM_HD Cx cx_div(Cx a, Cx b) {
    Float ns = cx_norm_sqr(b);
    Float re = a.re * b.re + a.im * b.im;
    Float im = a.im * b.re - a.re * b.im;
    return cx(re / ns, im / ns);
}
SHM_HD Cx real_div_cx(Float a, Cx b) {  // impl Div<Complex<T>> for T
    Float ns = cx_norm_sqr(b);
    return cx(a * b.re / ns, -a * b.im / ns);
}
SHM_HD Cx cx_sqrt(Cx z) {  // Complex::sqrt, num-complex 0.4.4
    bool im_zero = (z.im == 0.0f);
    bool re_zero = (z.re == 0.0f);
    if (im_zero) {
        bool re_pos = (float_to_bits(z.re) >> 31) == 0;
        if (re_pos) return cx(sqrt(z.re), z.im);
        Float im = sqrt(-z.re);
        bool im_pos = (float_to_bits(z.im) >> 31) == 0;
        return im_pos ? cx(0.0f, im) : cx(0.0f, -im);
    } else if (re_zero) {
        Float x = sqrt(abs(z.im) / 2.0f);
        bool im_pos = (float_to_bits(z.im) >> 31) == 0;
        return im_pos ? cx(x, x) : cx(x, -x);
    } else {
        Float r = hypot(z.re, z.im);
        Float theta = atan2(z.im, z.re);
        Float sr = sqrt(r);
        Float ht = theta / 2.0f;
        return cx(sr * cos(ht), sr * sin(ht));
    }
}
// scattering.rs:78-89
SHM_HD Float fresnel_complex(Float cos_theta_i, Cx eta) {
    cos_theta_i = clamp(cos_theta_i, 0.0f, 1.0f);
    Float sin2_theta_i = 1.0f - sqr(cos_theta_i);
    Cx sin2_theta_t = real_div_cx(sin2_theta_i, cx_mul(eta, eta));
    Cx cos_theta_t = cx_sqrt(cx(1.0f - sin2_theta_t.re, 0.0f - sin2_theta_t.im));
    Cx eci = cx(eta.re * cos_theta_i, eta.im * cos_theta_i);          // eta * cos_theta_i
    Cx r_parl = cx_div(cx(eci.re - cos_theta_t.re, eci.im - cos_theta_t.im),
                       cx(eci.re + cos_theta_t.re, eci.im + cos_theta_t.im));
    Cx ect = cx_mul(eta, cos_theta_t);                                // eta * cos_theta_t
    Cx r_perp = cx_div(cx(cos_theta_i - ect.re, 0.0f - ect.im), cx(cos_theta_i + ect.re, ect.im));
    return (cx_norm_sqr(r_parl) + cx_norm_sqr(r_perp)) / 2.0f;
}
// scattering.rs:92-104
SHM_HD Spec fresnel_complex_spectral(Float cos_theta_i, const Spec& eta, const Spec& k) {
    Spec s;
    for (int i = 0; i < NSPEC; ++i) s.v[i] = fresnel_complex(cos_theta_i, cx(eta.v[i], k.v[i]));
    return s;
}

// scattering.rs:107-220
struct TrowbridgeReitz {
    Float alpha_x, alpha_y;
    SHM_HD bool effectively_smooth() const { return alpha_x < 1e-3f && alpha_y < 1e-3f; }
    SHM_HD Float d(V3 wm) const {
        Float t2 = tan2_theta(wm);
        if (is_inf(t2)) return 0.0f;
        Float cos4_theta = sqr(cos2_theta(wm));
        if (cos4_theta < 1e-16f) return 0.0f;
        Float e = t2 * (sqr(cos_phi(wm) / alpha_x) + sqr(sin_phi(wm) / alpha_y));
        return 1.0f / (PI_F * alpha_x * alpha_y * cos4_theta * sqr(1.0f + e));
    }
    SHM_HD Float lambda(V3 w) const {
        Float t2 = tan2_theta(w);
        if (is_inf(t2)) return 0.0f;
        Float alpha2 = sqr(cos_phi(w) * alpha_x) + sqr(sin_phi(w) * alpha_y);
        return (-1.0f + sqrt(1.0f + alpha2 * t2)) / 2.0f;
    }
    SHM_HD Float g1(V3 w) const { return 1.0f / (1.0f + lambda(w)); }
    SHM_HD Float g(V3 wo, V3 wi) const { return 1.0f / (1.0f + lambda(wo) + lambda(wi)); }
    SHM_HD Float d_w(V3 w, V3 wm) const { return g1(w) / abs_cos_theta(w) * d(wm) * abs_dot(w, wm); }
    SHM_HD Float pdf(V3 w, V3 wm) const { return d_w(w, wm); }
    SHM_HD V3 sample_wm(V3 w, V2 u) const {
        V3 wh = normalize(v3(alpha_x * w.x, alpha_y * w.y, w.z));
        if (wh.z < 0.0f) wh = -wh;
        V3 t1 = (wh.z < 0.99999f) ? normalize(cross(v3(0.0f, 0.0f, 1.0f), wh)) : v3(1.0f, 0.0f, 0.0f);
        V3 t2 = cross(wh, t1);
        V2 p = sample_uniform_disk_polar(u);
        Float h = sqrt(1.0f - sqr(p.x));
        p.y = lerp((1.0f + wh.z) / 2.0f, h, p.y);
        Float pz = sqrt(max(0.0f, 1.0f - length_squared(p)));
        V3 nh = p.x * t1 + p.y * t2 + pz * wh;
        return normalize(v3(alpha_x * nh.x, alpha_y * nh.y, max(1e-6f, nh.z)));
    }
    SHM_HD void regularize() {
        if (alpha_x < 0.3f) alpha_x = clamp(2.0f * alpha_x, 0.1f, 0.3f);
        if (alpha_y < 0.3f) alpha_y = clamp(2.0f * alpha_y, 0.1f, 0.3f);
    }
};
SHM_HD TrowbridgeReitz trowbridge_reitz_new(Float ax, Float ay) {  // scattering.rs:113-126
    TrowbridgeReitz d;
    d.alpha_x = ax;
    d.alpha_y = ay;
    if (!d.effectively_smooth()) {
        d.alpha_x = max(d.alpha_x, 1e-4f);
        d.alpha_y = max(d.alpha_y, 1e-4f);
    }
    return d;
}
SHM_HD Float roughness_to_alpha(Float roughness) { return sqrt(roughness); }  // scattering.rs:208-210

// bxdf.rs:1773-1829
enum : uint32_t {
    BXDF_UNSET = 0,
    BXDF_REFLECTION = 1 << 0,
    BXDF_TRANSMISSION = 1 << 1,
    BXDF_DIFFUSE = 1 << 2,
    BXDF_GLOSSY = 1 << 3,
    BXDF_SPECULAR = 1 << 4,
    BXDF_DIFFUSE_REFLECTION = BXDF_DIFFUSE | BXDF_REFLECTION,
    BXDF_GLOSSY_REFLECTION = BXDF_GLOSSY | BXDF_REFLECTION,
    BXDF_GLOSSY_TRANSMISSION = BXDF_GLOSSY | BXDF_TRANSMISSION,
    BXDF_SPECULAR_REFLECTION = BXDF_SPECULAR | BXDF_REFLECTION,
    BXDF_SPECULAR_TRANSMISSION = BXDF_SPECULAR | BXDF_TRANSMISSION,
};
enum : uint32_t { REFLTRANS_REFLECTION = 1, REFLTRANS_TRANSMISSION = 2, REFLTRANS_ALL = 3 };
SHM_HD bool flags_is_reflective(uint32_t f) { return (f & BXDF_REFLECTION) != 0; }
SHM_HD bool flags_is_transmissive(uint32_t f) { return (f & BXDF_TRANSMISSION) != 0; }
SHM_HD bool flags_is_specular(uint32_t f) { return (f & BXDF_SPECULAR) != 0; }
SHM_HD bool flags_is_non_specular(uint32_t f) { return (f & (BXDF_DIFFUSE | BXDF_GLOSSY)) != 0; }

// bxdf.rs:1702-1742
struct BSDFSample {
    Spec f;
    V3 wi;
    Float pdf;
    uint32_t flags;
    Float eta;
    bool pdf_is_proportional;
};
SHM_HD BSDFSample bsdf_sample(const Spec& f, V3 wi, Float pdf, uint32_t flags, Float eta = 1.0f) {
    BSDFSample s;
    s.f = f; s.wi = wi; s.pdf = pdf; s.flags = flags; s.eta = eta; s.pdf_is_proportional = false;
    return s;
}

// One tagged struct for the four single-layer BxDFs (the reference's `enum BxDF`, bxdf.rs:96-103) ...
struct BaseBxDF {
    uint32_t kind;       // SHM_MATERIAL_DIFFUSE / CONDUCTOR / DIELECTRIC / THIN_DIELECTRIC
    Spec r;              // Diffuse: R ; Conductor: eta
    Spec k;              // Conductor: k
    Float eta;           // Dielectric / ThinDielectric
    TrowbridgeReitz mf;  // Conductor / Dielectric
};
// ... and the two coated ones, LayeredBxDF<Dielectric, Diffuse|Conductor, TWO_SIDED = true> (bxdf.rs:269-290, 460-480, 883-1620):
// for kind == SHM_MATERIAL_COATED_*, {eta, mf} are the top dielectric interface and {r, k, mf2} the bottom layer.
struct BxDF : BaseBxDF {
    TrowbridgeReitz mf2;  // CoatedConductor: the conductor's distribution
    Float thickness, g;
    Spec albedo;
    int max_depth, n_samples;
    int strict;  // ShmRenderParams::disable_reference_quirks: LayeredBxDF::pdf tests its reflected sample as PBRT-v4 does (0 = as the reference)
};
enum : int { MODE_RADIANCE = 0, MODE_IMPORTANCE = 1 };  // TransportMode

// ---- DiffuseBxDF, bxdf.rs:184-267 ----
SHM_HD Spec diffuse_f(const BaseBxDF& b, V3 wo, V3 wi) {
    if (!same_hemisphere(wo, wi)) return spec_const(0.0f);
    return b.r * INV_PI;
}
SHM_HD bool diffuse_sample_f(const BaseBxDF& b, V3 wo, V2 u, uint32_t sample_flags, BSDFSample& out) {
    if ((sample_flags & REFLTRANS_REFLECTION) == 0) return false;
    V3 wi = sample_cosine_hemisphere(u);
    if (wo.z < 0.0f) wi.z *= -1.0f;
    Float pdf = cosine_hemisphere_pdf(abs_cos_theta(wi));
    out = bsdf_sample(b.r * INV_PI, wi, pdf, BXDF_DIFFUSE_REFLECTION);
    return true;
}
SHM_HD Float diffuse_pdf(const BaseBxDF&, V3 wo, V3 wi, uint32_t sample_flags) {
    if ((sample_flags & REFLTRANS_REFLECTION) == 0 || !same_hemisphere(wo, wi)) return 0.0f;
    return cosine_hemisphere_pdf(abs_cos_theta(wi));
}

// ---- ConductorBxDF, bxdf.rs:328-458 ----
SHM_HD Spec conductor_f(const BaseBxDF& b, V3 wo, V3 wi) {
    if (!same_hemisphere(wo, wi)) return spec_const(0.0f);
    if (b.mf.effectively_smooth()) return spec_const(0.0f);
    Float cos_theta_o = abs_cos_theta(wo);
    Float cos_theta_i = abs_cos_theta(wi);
    if (cos_theta_i == 0.0f || cos_theta_o == 0.0f) return spec_const(0.0f);
    V3 wm = wi + wo;
    if (length_squared(wm) == 0.0f) return spec_const(0.0f);
    wm = normalize(wm);
    Spec f = fresnel_complex_spectral(abs_dot(wo, wm), b.r, b.k);
    return b.mf.d(wm) * f * b.mf.g(wo, wi) / (4.0f * cos_theta_o * cos_theta_i);
}
SHM_HD bool conductor_sample_f(const BaseBxDF& b, V3 wo, V2 u, uint32_t sample_flags, BSDFSample& out) {
    if ((sample_flags & REFLTRANS_REFLECTION) == 0) return false;
    if (b.mf.effectively_smooth()) {
        V3 wi = v3(-wo.x, -wo.y, wo.z);
        Spec f = fresnel_complex_spectral(abs_cos_theta(wi), b.r, b.k) / abs_cos_theta(wi);
        out = bsdf_sample(f, wi, 1.0f, BXDF_SPECULAR_REFLECTION);
        return true;
    }
    if (wo.z == 0.0f) return false;
    V3 wm = b.mf.sample_wm(wo, u);
    V3 wi = reflect(wo, wm);
    if (!same_hemisphere(wo, wi)) return false;
    Float pdf = b.mf.pdf(wo, wm) / (4.0f * abs_dot(wo, wm));
    Float cos_theta_o = abs_cos_theta(wo);
    Float cos_theta_i = abs_cos_theta(wi);
    if (cos_theta_i == 0.0f || cos_theta_o == 0.0f) return false;
    Spec fr = fresnel_complex_spectral(abs_dot(wo, wm), b.r, b.k);
    Spec f = b.mf.d(wm) * fr * b.mf.g(wo, wi) / (4.0f * cos_theta_o * cos_theta_i);
    out = bsdf_sample(f, wi, pdf, BXDF_GLOSSY_REFLECTION);
    return true;
}
SHM_HD Float conductor_pdf(const BaseBxDF& b, V3 wo, V3 wi, uint32_t sample_flags) {
    if ((sample_flags & REFLTRANS_REFLECTION) == 0 || !same_hemisphere(wo, wi) || b.mf.effectively_smooth())
        return 0.0f;
    V3 wm = wo + wi;
    if (length_squared(wm) == 0.0f) return 0.0f;
    wm = face_forward(normalize(wm), v3(0.0f, 0.0f, 1.0f));
    return b.mf.pdf(wo, wm) / (4.0f * abs_dot(wo, wm));
}

// ---- DielectricBxDF, bxdf.rs:518-795 (mode: the path itself only uses Radiance; LayeredBxDF samples wis in the other mode) ----
SHM_HD Spec dielectric_f(const BaseBxDF& b, V3 wo, V3 wi, int mode = MODE_RADIANCE) {
    if (b.eta == 1.0f || b.mf.effectively_smooth()) return spec_const(0.0f);
    Float cos_theta_o = cos_theta(wo);
    Float cos_theta_i = cos_theta(wi);
    bool reflect_ = cos_theta_i * cos_theta_o > 0.0f;
    Float etap = 1.0f;
    if (!reflect_) etap = (cos_theta_o > 0.0f) ? b.eta : (1.0f / b.eta);
    V3 wm = wi * etap + wo;
    if (cos_theta_i == 0.0f || cos_theta_o == 0.0f || length_squared(wm) == 0.0f) return spec_const(0.0f);
    wm = face_forward(normalize(wm), v3(0.0f, 0.0f, 1.0f));
    if (dot(wm, wi) * cos_theta_i < 0.0f || dot(wm, wo) * cos_theta_o < 0.0f) return spec_const(0.0f);
    Float f = fresnel_dielectric(dot(wo, wm), b.eta);
    if (reflect_) {
        return spec_const(b.mf.d(wm) * b.mf.g(wo, wi) * f / abs(4.0f * cos_theta_i * cos_theta_o));
    } else {
        Float denom = sqr(dot(wi, wm) + dot(wo, wm) / etap) * cos_theta_i * cos_theta_o;
        Float ft = b.mf.d(wm) * (1.0f - f) * b.mf.g(wo, wi) * abs(dot(wi, wm) * dot(wo, wm) / denom);
        if (mode == MODE_RADIANCE) ft /= sqr(etap);
        return spec_const(ft);
    }
}
SHM_HD bool dielectric_sample_f(const BaseBxDF& b, V3 wo, Float uc, V2 u, uint32_t sample_flags, BSDFSample& out, int mode = MODE_RADIANCE) {
    if (b.eta == 1.0f || b.mf.effectively_smooth()) {
        Float r = fresnel_dielectric(cos_theta(wo), b.eta);
        Float t = 1.0f - r;
        Float pr = r, pt = t;
        if ((sample_flags & REFLTRANS_REFLECTION) == 0) pr = 0.0f;
        if ((sample_flags & REFLTRANS_TRANSMISSION) == 0) pt = 0.0f;
        if (pr == 0.0f && pt == 0.0f) return false;
        if (uc < pr / (pr + pt)) {
            V3 wi = v3(-wo.x, -wo.y, wo.z);
            Spec fr = spec_const(r / abs_cos_theta(wi));
            out = bsdf_sample(fr, wi, pr / (pr + pt), BXDF_SPECULAR_REFLECTION);
            return true;
        } else {
            V3 wi; Float etap;
            if (!refract(wo, v3(0.0f, 0.0f, 1.0f), b.eta, wi, etap)) return false;
            Spec ft = spec_const(t / abs_cos_theta(wi));
            if (mode == MODE_RADIANCE) ft = ft / sqr(etap);
            out = bsdf_sample(ft, wi, pt / (pr + pt), BXDF_SPECULAR_TRANSMISSION, etap);
            return true;
        }
    } else {
        V3 wm = b.mf.sample_wm(wo, u);
        Float r = fresnel_dielectric(dot(wo, wm), b.eta);
        Float t = 1.0f - r;
        Float pr = r, pt = t;
        if ((sample_flags & REFLTRANS_REFLECTION) == 0) pr = 0.0f;
        if ((sample_flags & REFLTRANS_TRANSMISSION) == 0) pt = 0.0f;
        if (pr == 0.0f && pt == 0.0f) return false;
        if (uc < pr / (pr + pt)) {
            V3 wi = reflect(wo, wm);
            if (!same_hemisphere(wo, wi)) return false;
            Float pdf = b.mf.pdf(wo, wm) / (4.0f * abs_dot(wo, wm)) * pr / (pr + pt);
            Spec f = spec_const(b.mf.d(wm) * b.mf.g(wo, wi) * r / (4.0f * cos_theta(wi) * cos_theta(wo)));
            out = bsdf_sample(f, wi, pdf, BXDF_GLOSSY_REFLECTION);
            return true;
        } else {
            V3 wi; Float etap;
            if (!refract(wo, wm, b.eta, wi, etap)) return false;
            if (same_hemisphere(wo, wi) || wi.z == 0.0f) return false;
            Float denom = sqr(dot(wi, wm) + dot(wo, wm) / etap);
            Float dwm_dwi = abs_dot(wi, wm) / denom;
            Float pdf = b.mf.pdf(wo, wm) * dwm_dwi * pt / (pr + pt);
            Spec ft = spec_const(t * b.mf.d(wm) * b.mf.g(wo, wi)
                                 * abs(dot(wi, wm) * dot(wo, wm) / (cos_theta(wi) * cos_theta(wo) * denom)));
            if (mode == MODE_RADIANCE) ft = ft / sqr(etap);
            out = bsdf_sample(ft, wi, pdf, BXDF_GLOSSY_TRANSMISSION, etap);
            return true;
        }
    }
}
SHM_HD Float dielectric_pdf(const BaseBxDF& b, V3 wo, V3 wi, uint32_t sample_flags) {
    if (b.eta == 1.0f || b.mf.effectively_smooth()) return 0.0f;
    Float cos_theta_o = cos_theta(wo);
    Float cos_theta_i = cos_theta(wi);
    bool reflect_ = cos_theta_i * cos_theta_o > 0.0f;
    Float etap = 1.0f;
    if (!reflect_) etap = (cos_theta_o > 0.0f) ? b.eta : (1.0f / b.eta);
    V3 wm = wi * etap + wo;
    if (cos_theta_i == 0.0f || cos_theta_o == 0.0f || length_squared(wm) == 0.0f) return 0.0f;
    wm = face_forward(normalize(wm), v3(0.0f, 0.0f, 1.0f));
    if (dot(wm, wi) * cos_theta_i < 0.0f || dot(wm, wo) * cos_theta_o < 0.0f) return 0.0f;
    Float r = fresnel_dielectric(dot(wo, wm), b.eta);
    Float t = 1.0f - r;
    Float pr = r, pt = t;
    if ((sample_flags & REFLTRANS_REFLECTION) == 0) pr = 0.0f;
    if ((sample_flags & REFLTRANS_TRANSMISSION) == 0) pt = 0.0f;
    if (pr == 0.0f && pt == 0.0f) return 0.0f;
    if (reflect_) {
        return b.mf.pdf(wo, wm) / (4.0f * abs_dot(wo, wm)) * pr / (pr + pt);
    } else {
        Float denom = sqr(dot(wi, wm) + dot(wo, wm) / etap);
        Float dwm_dwi = abs_dot(wi, wm) / denom;
        return b.mf.pdf(wo, wm) * dwm_dwi * pt / (pr + pt);
    }
}
SHM_HD uint32_t dielectric_flags(const BaseBxDF& b) {
    uint32_t flags = (b.eta == 1.0f) ? BXDF_TRANSMISSION : (BXDF_REFLECTION | BXDF_TRANSMISSION);
    return flags | (b.mf.effectively_smooth() ? BXDF_SPECULAR : BXDF_GLOSSY);
}

// ---- ThinDielectricBxDF, bxdf.rs:797-881 ----
SHM_HD bool thin_dielectric_sample_f(const BaseBxDF& b, V3 wo, Float uc, uint32_t sample_flags, BSDFSample& out) {
    Float r = fresnel_dielectric(abs_cos_theta(wo), b.eta);
    Float t = 1.0f - r;
    if (r < 1.0f) {
        r += sqr(t) * r / (1.0f - sqr(r));
        t = 1.0f - r;
    }
    Float pr = r, pt = t;
    if ((sample_flags & REFLTRANS_REFLECTION) == 0) pr = 0.0f;
    if ((sample_flags & REFLTRANS_TRANSMISSION) == 0) pt = 0.0f;
    if (pr == 0.0f && pt == 0.0f) return false;
    if (uc < pr / (pr + pt)) {
        V3 wi = v3(-wo.x, -wo.y, wo.z);
        out = bsdf_sample(spec_const(r / abs_cos_theta(wi)), wi, pr / (pr + pt), BXDF_SPECULAR_REFLECTION);
    } else {
        V3 wi = -wo;
        out = bsdf_sample(spec_const(t / abs_cos_theta(wi)), wi, pt / (pr + pt), BXDF_SPECULAR_TRANSMISSION);
    }
    return true;
}

// ---- enum dispatch over the single-layer BxDFs, bxdf.rs:105-182 ----
SHM_HD uint32_t base_flags(const BaseBxDF& b) {
    switch (b.kind) {
        case SHM_MATERIAL_DIFFUSE: return is_zero(b.r) ? BXDF_UNSET : BXDF_DIFFUSE_REFLECTION;
        case SHM_MATERIAL_CONDUCTOR: return b.mf.effectively_smooth() ? BXDF_SPECULAR_REFLECTION : BXDF_GLOSSY_REFLECTION;
        case SHM_MATERIAL_DIELECTRIC: return dielectric_flags(b);
        default: return BXDF_REFLECTION | BXDF_TRANSMISSION | BXDF_SPECULAR;
    }
}
// SHM_BASE_BXDF_CALL: a translation unit may ask for these three dispatchers as REAL calls (the LayeredBxDF scatter kernel, whose three random
// walks reach them from ~30 sites: inlined, the kernel is 125 k instructions); everywhere else they inline as before. Same arithmetic either way.
#ifndef SHM_BASE_BXDF_CALL
#define SHM_BASE_BXDF_CALL SHM_HD
#endif
// As real calls the dispatchers take the interface BY VALUE and return their result by value: aggregates of up to 64 bytes travel in registers
// under the AMDGPU calling convention (BaseBxDF is 48 B, a BSDFSample with its flag 48 B), while a `const BaseBxDF&` or a `BSDFSample&` has to
// live in scratch memory — which is what round 2's layered scatter kernel did at every one of its ~30 call sites (416-464 B of scratch per
// lane, 7 GB written per launch: profiles/r02_staged_S3c.txt). The reference-shaped signatures below are inline wrappers.
// Behind the interface (12 registers) and wo (3) every further argument is a SCALAR: clang hands an aggregate that no longer fits the first 16
// argument registers over through the stack (round 4 found one scratch_store_dwordx3 + scratch_load_dwordx3 of `wi` around each of these calls),
// scalars always travel in registers.
struct BSDFSampleOpt { BSDFSample s; uint32_t ok; };
SHM_BASE_BXDF_CALL Spec base_f_v(BaseBxDF b, V3 wo, Float wi_x, Float wi_y, Float wi_z, int mode) {
    const V3 wi = v3(wi_x, wi_y, wi_z);
    switch (b.kind) {
        case SHM_MATERIAL_DIFFUSE: return diffuse_f(b, wo, wi);
        case SHM_MATERIAL_CONDUCTOR: return conductor_f(b, wo, wi);
        case SHM_MATERIAL_DIELECTRIC: return dielectric_f(b, wo, wi, mode);
        default: return spec_const(0.0f);
    }
}
SHM_BASE_BXDF_CALL BSDFSampleOpt base_sample_f_v(BaseBxDF b, V3 wo, Float uc, Float u_x, Float u_y, uint32_t sample_flags, int mode) {
    const V2 u = v2(u_x, u_y);
    BSDFSampleOpt r;
    r.s = bsdf_sample(spec_const(0.0f), v3s(0.0f), 0.0f, 0u);
    bool ok;
    switch (b.kind) {
        case SHM_MATERIAL_DIFFUSE: ok = diffuse_sample_f(b, wo, u, sample_flags, r.s); break;
        case SHM_MATERIAL_CONDUCTOR: ok = conductor_sample_f(b, wo, u, sample_flags, r.s); break;
        case SHM_MATERIAL_DIELECTRIC: ok = dielectric_sample_f(b, wo, uc, u, sample_flags, r.s, mode); break;
        default: ok = thin_dielectric_sample_f(b, wo, uc, sample_flags, r.s); break;
    }
    r.ok = ok ? 1u : 0u;
    return r;
}
SHM_BASE_BXDF_CALL Float base_pdf_v(BaseBxDF b, V3 wo, Float wi_x, Float wi_y, Float wi_z, uint32_t sample_flags) {
    const V3 wi = v3(wi_x, wi_y, wi_z);
    switch (b.kind) {
        case SHM_MATERIAL_DIFFUSE: return diffuse_pdf(b, wo, wi, sample_flags);
        case SHM_MATERIAL_CONDUCTOR: return conductor_pdf(b, wo, wi, sample_flags);
        case SHM_MATERIAL_DIELECTRIC: return dielectric_pdf(b, wo, wi, sample_flags);
        default: return 0.0f;
    }
}
SHM_HD Spec base_f(const BaseBxDF& b, V3 wo, V3 wi, int mode = MODE_RADIANCE) { return base_f_v(b, wo, wi.x, wi.y, wi.z, mode); }
SHM_HD bool base_sample_f(const BaseBxDF& b, V3 wo, Float uc, V2 u, uint32_t sample_flags, BSDFSample& out, int mode = MODE_RADIANCE) {
    const BSDFSampleOpt r = base_sample_f_v(b, wo, uc, u.x, u.y, sample_flags, mode);
    if (r.ok) out = r.s;   // (a failed sample leaves `out` as it was, as the by-reference form did)
    return r.ok != 0u;
}
SHM_HD Float base_pdf(const BaseBxDF& b, V3 wo, V3 wi, uint32_t sample_flags) { return base_pdf_v(b, wo, wi.x, wi.y, wi.z, sample_flags); }

// ---- Henyey-Greenstein, scattering.rs:231-260; HGPhaseFunction, media.rs:8-40 (p == pdf) ----
SHM_HD Float henyey_greenstein(Float cos_theta, Float g) {
    g = clamp(g, -0.99f, 0.99f);
    Float denom = 1.0f + sqr(g) + 2.0f * g * cos_theta;
    return INV_4PI * (1.0f - sqr(g)) / (denom * safe_sqrt(denom));
}
SHM_HD Float hg_p(Float g, V3 wo, V3 wi) { return henyey_greenstein(dot(wo, wi), g); }
SHM_HD V3 sample_henyey_greenstein(V3 wo, Float g, V2 u, Float& pdf) {
    g = clamp(g, -0.99f, 0.99f);
    Float cos_t;
    if (abs(g) < 1e-3f) cos_t = 1.0f - 2.0f * u.x;
    else cos_t = -1.0f / (2.0f * g) * (1.0f + sqr(g) - sqr((1.0f - sqr(g)) / (1.0f + g - 2.0f * g * u.x)));
    Float sin_t = safe_sqrt(1.0f - sqr(cos_t));
    Float phi = 2.0f * PI_F * u.y;
    Frame w_frame = frame_from_z(wo);
    V3 wi = w_frame.from_local(spherical_direction(sin_t, cos_t, phi));
    pdf = henyey_greenstein(cos_t, g);
    return wi;
}
// sampling.rs:789-792 (returns the exponential density at x, not a sample: reference behaviour preserved)
SHM_HD Float sample_exponential(Float x, Float a) { return a * exp(-a * x); }

// ---- LayeredBxDF<DielectricBxDF, DiffuseBxDF | ConductorBxDF, TWO_SIDED = true>, bxdf.rs:883-1620 ----
// The reference draws the inner random walk from SmallRng::from_entropy() (bxdf.rs:1014, 1292, 1426: "TODO Use a seed for
// this"), so its coated materials are not reproducible run to run. Defined here instead, the way PBRT-v4 does: a PCG32
// stream seeded from a hash of the arguments of the call (parity at this boundary is unpinned: no reference value exists).
SHM_HD uint64_t hash_f32(uint64_t h, Float x) { return mix_bits(h ^ ((uint64_t)float_to_bits(x) + 0x9e3779b97f4a7c15ULL)); }
SHM_HD uint64_t hash_v3(uint64_t h, V3 v) { return hash_f32(hash_f32(hash_f32(h, v.x), v.y), v.z); }
SHM_HD Rng layered_rng(uint64_t sequence, uint64_t seed) {
    Rng r;
    rng_set_sequence(r, sequence, seed);
    return r;
}
SHM_HD Float layered_r(Rng& rng) { return min(sampler_get_1d(rng), ONE_MINUS_EPSILON); }  // the closure `r` of bxdf.rs:1015-1020
SHM_HD V2 layered_r2(Rng& rng) { Float a = layered_r(rng); Float b = layered_r(rng); return v2(a, b); }

SHM_HD BaseBxDF layered_top(const BxDF& l) {
    BaseBxDF t;
    t.kind = SHM_MATERIAL_DIELECTRIC; t.r = spec_const(0.0f); t.k = spec_const(0.0f); t.eta = l.eta; t.mf = l.mf;
    return t;
}
SHM_HD BaseBxDF layered_bottom(const BxDF& l) {
    BaseBxDF b;
    b.kind = (l.kind == SHM_MATERIAL_COATED_DIFFUSE) ? (uint32_t)SHM_MATERIAL_DIFFUSE : (uint32_t)SHM_MATERIAL_CONDUCTOR;
    b.r = l.r; b.k = l.k; b.eta = 1.0f; b.mf = l.mf2;
    return b;
}
// LayeredBxDF::tr, bxdf.rs:925-933. `Float::MIN` is the most negative finite f32 there, so the early-out never fires.
SHM_HD Float layered_tr(Float dz, V3 w) {
    if (abs(dz) <= -3.40282347e+38f) return 1.0f;
    return exp(-abs(dz / w.z));
}
SHM_HD bool sample_unusable(const BSDFSample& s) { return is_zero(s.f) || s.pdf == 0.0f || s.wi.z == 0.0f; }

// Whether the BOTTOM interface of a LayeredBxDF kind can transmit. The two layered materials the reference has (CoatedDiffuse, CoatedConductor:
// material.rs:917-963, 1189-1256) put a DiffuseBxDF / ConductorBxDF there — reflection only. The opposite-hemisphere shortcut of layered_f and the
// deferred NEE's matching skip (k_scatter.inl) are valid ONLY under this; a future layered kind with a transmissive bottom must return true here.
SHM_HD constexpr bool layered_bottom_transmits(uint32_t kind) { return !(kind == SHM_MATERIAL_COATED_DIFFUSE || kind == SHM_MATERIAL_COATED_CONDUCTOR); }

// LayeredBxDF::f, bxdf.rs:941-1218. SHORTCUT = false evaluates the walk without the opposite-hemisphere early-out (tests compare the two).
template <bool SHORTCUT = true>
SHM_HD Spec layered_f(const BxDF& l, V3 wo, V3 wi, int mode) {
    Spec f = spec_const(0.0f);
    // wo and wi on opposite sides: the walk would have to LEAVE through the bottom interface, and the first thing every sample does for that is
    // bottom.sample_f(wi, .., TRANSMISSION) (bxdf.rs:1004-1012) — which a DiffuseBxDF / ConductorBxDF refuses (no transmission lobe): every
    // sample `continue`s, the sum stays 0 and 0 / n_samples is returned. Known from the two z signs alone, so the (local, unobservable) generator
    // and the top interface's sample are not even started; the GPU's deferred NEE uses the same test before it queues an evaluation.
    if (SHORTCUT && !layered_bottom_transmits(l.kind) && !same_hemisphere(wo, wi)) return f;
    if (wo.z < 0.0f) { wo = -wo; wi = -wi; }  // TWO_SIDED
    const bool entered_top = true;            // TWO_SIDED || wo.z > 0
    const BaseBxDF top = layered_top(l), bottom = layered_bottom(l);
    const BaseBxDF& enter_i = entered_top ? top : bottom;
    const bool exit_is_bottom = same_hemisphere(wo, wi) ^ entered_top;
    const BaseBxDF& exit_i = exit_is_bottom ? bottom : top;
    const BaseBxDF& non_exit_i = exit_is_bottom ? top : bottom;
    const Float exit_z = exit_is_bottom ? 0.0f : l.thickness;
    const uint32_t exit_flags = base_flags(exit_i), non_exit_flags = base_flags(non_exit_i);
    const Float n_samples_f = (Float)l.n_samples;
    if (same_hemisphere(wo, wi)) f = base_f(enter_i, wo, wi, mode) * n_samples_f;
    Rng rng = layered_rng(hash_v3(0x5eed0001ULL, wi), hash_v3(0ULL, wo));
    const int wis_mode = (mode == MODE_RADIANCE) ? MODE_IMPORTANCE : MODE_RADIANCE;
    for (int s = 0; s < l.n_samples; ++s) {
        Float uc = layered_r(rng);
        V2 u = layered_r2(rng);
        BSDFSample wos;
        if (!base_sample_f(enter_i, wo, uc, u, REFLTRANS_TRANSMISSION, wos, mode)) continue;
        if (sample_unusable(wos)) continue;
        uc = layered_r(rng);
        u = layered_r2(rng);
        BSDFSample wis;
        if (!base_sample_f(exit_i, wi, uc, u, REFLTRANS_TRANSMISSION, wis, wis_mode)) continue;
        if (sample_unusable(wis)) continue;
        Spec beta = wos.f * abs_cos_theta(wos.wi) / wos.pdf;
        Float z = entered_top ? l.thickness : 0.0f;
        V3 w = wos.wi;
        for (int depth = 0; depth < l.max_depth; ++depth) {
            if (depth > 3 && max_component_value(beta) < 0.25f) {
                Float q = max(0.0f, 1.0f - max_component_value(beta));
                if (layered_r(rng) < q) break;
                beta = beta / (1.0f - q);
            }
            if (is_zero(l.albedo)) {
                z = (z == l.thickness) ? 0.0f : l.thickness;
                beta = beta * layered_tr(l.thickness, w);
            } else {
                const Float sigma_t = 1.0f;
                Float dz = sample_exponential(layered_r(rng), sigma_t / abs(w.z));
                Float zp = (w.z > 0.0f) ? (z + dz) : (z - dz);
                if (z == zp) continue;
                if (0.0f < zp && zp < l.thickness) {
                    Float wt = 1.0f;
                    if (!flags_is_specular(exit_flags)) wt = power_heuristic(1, wis.pdf, 1, hg_p(l.g, -w, -wis.wi));
                    f = f + beta * l.albedo * hg_p(l.g, -w, -wis.wi) * wt * layered_tr(zp - exit_z, wis.wi) * wis.f / wis.pdf;
                    V2 up = layered_r2(rng);
                    Float ps_pdf;
                    V3 ps_wi = sample_henyey_greenstein(-w, l.g, up, ps_pdf);
                    if (ps_pdf == 0.0f || ps_wi.z == 0.0f) continue;
                    beta = beta * (l.albedo * ps_pdf / ps_pdf);
                    w = ps_wi;
                    z = zp;
                    if (((z < exit_z && w.z > 0.0f) || (z > exit_z && w.z < 0.0f)) && !flags_is_specular(exit_flags)) {
                        Spec f_exit = base_f(exit_i, -w, wi, mode);
                        if (!is_zero(f_exit)) {
                            Float exit_pdf = base_pdf(exit_i, -w, wi, REFLTRANS_TRANSMISSION);
                            Float wt2 = power_heuristic(1, ps_pdf, 1, exit_pdf);
                            f = f + beta * layered_tr(zp - exit_z, ps_wi) * f_exit * wt2;
                        }
                    }
                    continue;
                }
                z = clamp(zp, 0.0f, l.thickness);
            }
            if (z == exit_z) {
                Float uc2 = layered_r(rng);
                V2 u2 = layered_r2(rng);
                BSDFSample bs;
                if (!base_sample_f(exit_i, -w, uc2, u2, REFLTRANS_REFLECTION, bs, mode)) break;
                if (sample_unusable(bs)) break;
                beta = beta * (bs.f * abs_cos_theta(bs.wi) / bs.pdf);
                w = bs.wi;
            } else {
                if (!flags_is_specular(non_exit_flags)) {
                    Float wt = 1.0f;
                    if (!flags_is_specular(exit_flags)) wt = power_heuristic(1, wis.pdf, 1, base_pdf(non_exit_i, -w, -wis.wi, REFLTRANS_ALL));
                    f = f + beta * base_f(non_exit_i, -w, -wis.wi, mode) * abs_cos_theta(wis.wi) * wt * layered_tr(l.thickness, wis.wi) * wis.f / wis.pdf;
                }
                Float uc2 = layered_r(rng);
                V2 u2 = layered_r2(rng);
                BSDFSample bs;
                if (!base_sample_f(non_exit_i, -w, uc2, u2, REFLTRANS_REFLECTION, bs, mode)) break;
                if (sample_unusable(bs)) break;
                beta = beta * (bs.f * abs_cos_theta(bs.wi) / bs.pdf);
                w = bs.wi;
                if (!flags_is_specular(exit_flags)) {
                    Spec f_exit = base_f(exit_i, -w, wi, mode);
                    if (!is_zero(f_exit)) {
                        Float wt = 1.0f;
                        if (!flags_is_specular(non_exit_flags)) {
                            Float exit_pdf = base_pdf(exit_i, -w, wi, REFLTRANS_TRANSMISSION);
                            wt = power_heuristic(1, bs.pdf, 1, exit_pdf);
                        }
                        f = f + beta * layered_tr(l.thickness, bs.wi) * f_exit * wt;
                    }
                }
            }
        }
    }
    return f / n_samples_f;
}

// LayeredBxDF::sample_f, bxdf.rs:1220-1404 (sample_flags must be ALL there: assert)
SHM_HD bool layered_sample_f(const BxDF& l, V3 wo, Float uc, V2 u, int mode, BSDFSample& out) {
    bool flip_wi = false;
    if (wo.z < 0.0f) { wo = -wo; flip_wi = true; }  // TWO_SIDED
    const bool entered_top = true;
    const BaseBxDF top = layered_top(l), bottom = layered_bottom(l);
    BSDFSample bs;
    if (!base_sample_f(entered_top ? top : bottom, wo, uc, u, REFLTRANS_ALL, bs, mode)) return false;
    if (sample_unusable(bs)) return false;
    if (flags_is_reflective(bs.flags)) {
        if (flip_wi) bs.wi = -bs.wi;
        bs.pdf_is_proportional = true;
        out = bs;
        return true;
    }
    V3 w = bs.wi;
    bool specular_path = flags_is_specular(bs.flags);
    Rng rng = layered_rng(hash_f32(hash_f32(hash_f32(0x5eed0002ULL, uc), u.x), u.y), hash_v3(0ULL, wo));
    Spec f = bs.f * abs_cos_theta(bs.wi);
    Float pdf = bs.pdf;
    Float z = entered_top ? l.thickness : 0.0f;
    for (int depth = 0; depth < l.max_depth; ++depth) {
        Float rr_beta = max_component_value(f) / pdf;
        if (depth > 3 && rr_beta < 0.25f) {
            Float q = max(0.0f, 1.0f - rr_beta);
            if (layered_r(rng) < q) return false;
            pdf *= 1.0f - q;
        }
        if (w.z == 0.0f) return false;
        if (!is_zero(l.albedo)) {
            const Float sigma_t = 1.0f;
            Float dz = sample_exponential(layered_r(rng), sigma_t / abs_cos_theta(w));
            Float zp = (w.z > 0.0f) ? (z + dz) : (z - dz);
            if (zp == z) return false;
            if (0.0f < zp && zp < l.thickness) {
                V2 up = layered_r2(rng);
                Float ps_pdf;
                V3 ps_wi = sample_henyey_greenstein(-w, l.g, up, ps_pdf);
                if (ps_pdf == 0.0f || ps_wi.z == 0.0f) return false;
                f = f * (l.albedo * ps_pdf);
                pdf *= ps_pdf;
                specular_path = false;
                w = ps_wi;
                z = zp;
                continue;
            }
            z = clamp(zp, 0.0f, l.thickness);
        } else {
            z = (z == l.thickness) ? 0.0f : l.thickness;
            f = f * layered_tr(l.thickness, w);
        }
        const BaseBxDF& iface = (z == 0.0f) ? bottom : top;
        Float uc2 = layered_r(rng);
        V2 u2 = layered_r2(rng);
        BSDFSample s2;
        if (!base_sample_f(iface, -w, uc2, u2, REFLTRANS_ALL, s2, mode)) return false;
        if (sample_unusable(s2)) return false;
        f = f * s2.f;
        pdf *= s2.pdf;
        specular_path = specular_path && flags_is_specular(s2.flags);
        w = s2.wi;
        if (flags_is_transmissive(s2.flags)) {
            uint32_t flags = same_hemisphere(wo, w) ? BXDF_REFLECTION : BXDF_TRANSMISSION;
            flags |= specular_path ? BXDF_SPECULAR : BXDF_GLOSSY;
            if (flip_wi) w = -w;
            out = bsdf_sample(f, w, pdf, flags, 1.0f);
            out.pdf_is_proportional = true;
            return true;
        }
        f = f * abs_cos_theta(s2.wi);
    }
    return false;
}

// LayeredBxDF::sample_f as a RESUMABLE walk: `layered_sample_begin` is everything layered_sample_f does before its loop, `layered_sample_step` one iteration of
// the loop — the same statements in the same order, with the loop's locals in a struct. The staged layered kernel (k_scatter_layered.inl) advances 64 deposited
// walks per wave two steps at a time instead of letting every lane wait for the longest walk of its wave; tests/test_layered.py holds begin + steps bitwise equal
// to layered_sample_f (which stays the oracle's text).
struct LayeredWalk {
    V3 w;
    Spec f;
    Float pdf, z;
    Float wo_z;  // of the flipped wo (> 0 unless wo.z is NaN): all that same_hemisphere(wo, w) reads of it
    int depth;
    bool specular_path, flip_wi;
    Rng rng;
};
enum : int { WALK_FAILED = 0, WALK_DONE = 1, WALK_CONTINUES = 2 };
SHM_HD int layered_sample_begin(const BxDF& l, V3 wo, Float uc, V2 u, int mode, BSDFSample& out, LayeredWalk& k) {
    bool flip_wi = false;
    if (wo.z < 0.0f) { wo = -wo; flip_wi = true; }  // TWO_SIDED
    const BaseBxDF top = layered_top(l);             // (entered_top is always true)
    BSDFSample bs;
    if (!base_sample_f(top, wo, uc, u, REFLTRANS_ALL, bs, mode)) return WALK_FAILED;
    if (sample_unusable(bs)) return WALK_FAILED;
    if (flags_is_reflective(bs.flags)) {
        if (flip_wi) bs.wi = -bs.wi;
        bs.pdf_is_proportional = true;
        out = bs;
        return WALK_DONE;
    }
    k.w = bs.wi;
    k.specular_path = flags_is_specular(bs.flags);
    k.rng = layered_rng(hash_f32(hash_f32(hash_f32(0x5eed0002ULL, uc), u.x), u.y), hash_v3(0ULL, wo));
    k.f = bs.f * abs_cos_theta(bs.wi);
    k.pdf = bs.pdf;
    k.z = l.thickness;
    k.wo_z = wo.z;
    k.depth = 0;
    k.flip_wi = flip_wi;
    return WALK_CONTINUES;
}
SHM_HD int layered_sample_step(const BxDF& l, int mode, LayeredWalk& k, BSDFSample& out) {
    if (!(k.depth < l.max_depth)) return WALK_FAILED;  // (the loop's condition; the function's last statement)
    const int depth = k.depth++;
    Float rr_beta = max_component_value(k.f) / k.pdf;
    if (depth > 3 && rr_beta < 0.25f) {
        Float q = max(0.0f, 1.0f - rr_beta);
        if (layered_r(k.rng) < q) return WALK_FAILED;
        k.pdf *= 1.0f - q;
    }
    if (k.w.z == 0.0f) return WALK_FAILED;
    if (!is_zero(l.albedo)) {
        const Float sigma_t = 1.0f;
        Float dz = sample_exponential(layered_r(k.rng), sigma_t / abs_cos_theta(k.w));
        Float zp = (k.w.z > 0.0f) ? (k.z + dz) : (k.z - dz);
        if (zp == k.z) return WALK_FAILED;
        if (0.0f < zp && zp < l.thickness) {
            V2 up = layered_r2(k.rng);
            Float ps_pdf;
            V3 ps_wi = sample_henyey_greenstein(-k.w, l.g, up, ps_pdf);
            if (ps_pdf == 0.0f || ps_wi.z == 0.0f) return WALK_FAILED;
            k.f = k.f * (l.albedo * ps_pdf);
            k.pdf *= ps_pdf;
            k.specular_path = false;
            k.w = ps_wi;
            k.z = zp;
            return WALK_CONTINUES;
        }
        k.z = clamp(zp, 0.0f, l.thickness);
    } else {
        k.z = (k.z == l.thickness) ? 0.0f : l.thickness;
        k.f = k.f * layered_tr(l.thickness, k.w);
    }
    const BaseBxDF iface = (k.z == 0.0f) ? layered_bottom(l) : layered_top(l);
    Float uc2 = layered_r(k.rng);
    V2 u2 = layered_r2(k.rng);
    BSDFSample s2;
    if (!base_sample_f(iface, -k.w, uc2, u2, REFLTRANS_ALL, s2, mode)) return WALK_FAILED;
    if (sample_unusable(s2)) return WALK_FAILED;
    k.f = k.f * s2.f;
    k.pdf *= s2.pdf;
    k.specular_path = k.specular_path && flags_is_specular(s2.flags);
    k.w = s2.wi;
    if (flags_is_transmissive(s2.flags)) {
        uint32_t flags = (k.wo_z * k.w.z > 0.0f) ? BXDF_REFLECTION : BXDF_TRANSMISSION;  // same_hemisphere(wo, w)
        flags |= k.specular_path ? BXDF_SPECULAR : BXDF_GLOSSY;
        V3 w = k.w;
        if (k.flip_wi) w = -w;
        out = bsdf_sample(k.f, w, k.pdf, flags, 1.0f);
        out.pdf_is_proportional = true;
        return WALK_DONE;
    }
    k.f = k.f * abs_cos_theta(s2.wi);
    return WALK_CONTINUES;
}

// LayeredBxDF::pdf, bxdf.rs:1406-1575 (sample_flags must be ALL there: assert)
SHM_HD Float layered_pdf(const BxDF& l, V3 wo, V3 wi, int mode) {
    if (wo.z < 0.0f) { wo = -wo; wi = -wi; }  // TWO_SIDED
    Rng rng = layered_rng(hash_v3(0x5eed0003ULL, wo), hash_v3(0ULL, wi));
    const bool entered_top = true;
    const BaseBxDF top = layered_top(l), bottom = layered_bottom(l);
    const Float n_samples_f = (Float)l.n_samples;
    const int wis_mode = (mode == MODE_RADIANCE) ? MODE_IMPORTANCE : MODE_RADIANCE;
    Float pdf_sum = 0.0f;
    if (same_hemisphere(wo, wi)) pdf_sum += n_samples_f * base_pdf(entered_top ? top : bottom, wo, wi, REFLTRANS_REFLECTION);
    for (int s = 0; s < l.n_samples; ++s) {
        if (same_hemisphere(wo, wi)) {
            const BaseBxDF& r_i = entered_top ? bottom : top;
            const BaseBxDF& t_i = entered_top ? top : bottom;
            const uint32_t r_flags = base_flags(r_i), t_flags = base_flags(t_i);
            Float uc = layered_r(rng);
            V2 u = layered_r2(rng);
            BSDFSample wos, wis;
            bool have_wos = base_sample_f(t_i, wo, uc, u, REFLTRANS_TRANSMISSION, wos, mode);
            uc = layered_r(rng);
            u = layered_r2(rng);
            bool have_wis = base_sample_f(t_i, wi, uc, u, REFLTRANS_TRANSMISSION, wis, wis_mode);
            if (have_wos && have_wis && !is_zero(wos.f) && wos.pdf > 0.0f && !is_zero(wis.f) && wis.pdf > 0.0f) {
                if (!flags_is_non_specular(t_flags)) {
                    pdf_sum += base_pdf(r_i, -wos.wi, -wis.wi, REFLTRANS_ALL);
                } else {
                    uc = layered_r(rng);
                    u = layered_r2(rng);
                    BSDFSample rs;
                    // bxdf.rs:1491-1506 uses `rs` as soon as it exists; PBRT-v4 also requires rs.f != 0 && rs.pdf > 0 — without that a
                    // cosine sample exactly on the horizon (rs.pdf = 0, then t_pdf = 0) makes power_heuristic 0 / 0
                    if (base_sample_f(r_i, -wos.wi, uc, u, REFLTRANS_ALL, rs, mode) && (!l.strict || (!is_zero(rs.f) && rs.pdf > 0.0f))) {
                        if (!flags_is_non_specular(r_flags)) {
                            pdf_sum += base_pdf(t_i, -rs.wi, wi, REFLTRANS_ALL);
                        } else {
                            Float r_pdf = base_pdf(r_i, -wos.wi, -wis.wi, REFLTRANS_ALL);
                            Float wt = power_heuristic(1, wis.pdf, 1, r_pdf);
                            pdf_sum += wt * r_pdf;
                            Float t_pdf = base_pdf(t_i, -rs.wi, wi, REFLTRANS_ALL);
                            wt = power_heuristic(1, rs.pdf, 1, t_pdf);
                            pdf_sum += wt * t_pdf;
                        }
                    }
                }
            }
        } else {
            const BaseBxDF& to_i = entered_top ? top : bottom;
            const BaseBxDF& ti_i = entered_top ? bottom : top;
            Float uc = layered_r(rng);
            V2 u = layered_r2(rng);
            BSDFSample wos;
            if (!base_sample_f(to_i, wo, uc, u, REFLTRANS_ALL, wos, mode)) continue;
            if (sample_unusable(wos) || flags_is_reflective(wos.flags)) continue;
            uc = layered_r(rng);
            u = layered_r2(rng);
            BSDFSample wis;
            if (!base_sample_f(ti_i, wi, uc, u, REFLTRANS_ALL, wis, wis_mode)) continue;
            if (sample_unusable(wis) || flags_is_reflective(wis.flags)) continue;
            if (flags_is_specular(base_flags(to_i))) pdf_sum += base_pdf(ti_i, -wos.wi, wi, REFLTRANS_ALL);
            else if (flags_is_specular(base_flags(ti_i))) pdf_sum += base_pdf(to_i, wo, -wis.wi, REFLTRANS_ALL);
            else pdf_sum += (base_pdf(to_i, wo, -wis.wi, REFLTRANS_ALL) + base_pdf(ti_i, -wos.wi, wi, REFLTRANS_ALL)) / 2.0f;
        }
    }
    return lerp(0.9f, 1.0f / (4.0f * PI_F), pdf_sum / n_samples_f);
}
// LayeredBxDF::flags, bxdf.rs:1577-1606
SHM_HD uint32_t layered_flags(const BxDF& l) {
    uint32_t tf = base_flags(layered_top(l)), bf = base_flags(layered_bottom(l));
    uint32_t flags = BXDF_REFLECTION;
    if (flags_is_specular(tf)) flags |= BXDF_SPECULAR;
    if ((tf & BXDF_DIFFUSE) || (bf & BXDF_DIFFUSE) || !is_zero(l.albedo)) flags |= BXDF_DIFFUSE;
    else if ((tf & BXDF_GLOSSY) || (bf & BXDF_GLOSSY)) flags |= BXDF_GLOSSY;
    if (flags_is_transmissive(tf) && flags_is_transmissive(bf)) flags |= BXDF_TRANSMISSION;
    return flags;
}

// ---- enum dispatch with the coated BxDFs, bxdf.rs:105-182 ----
SHM_HD bool bxdf_is_layered(const BxDF& b) { return b.kind == SHM_MATERIAL_COATED_DIFFUSE || b.kind == SHM_MATERIAL_COATED_CONDUCTOR; }
SHM_HD uint32_t bxdf_flags(const BxDF& b) { return bxdf_is_layered(b) ? layered_flags(b) : base_flags(b); }
SHM_HD Spec bxdf_f(const BxDF& b, V3 wo, V3 wi) { return bxdf_is_layered(b) ? layered_f(b, wo, wi, MODE_RADIANCE) : base_f(b, wo, wi); }
SHM_HD bool bxdf_sample_f(const BxDF& b, V3 wo, Float uc, V2 u, uint32_t sample_flags, BSDFSample& out) {
    if (bxdf_is_layered(b)) return layered_sample_f(b, wo, uc, u, MODE_RADIANCE, out);
    return base_sample_f(b, wo, uc, u, sample_flags, out);
}
SHM_HD Float bxdf_pdf(const BxDF& b, V3 wo, V3 wi, uint32_t sample_flags) {
    return bxdf_is_layered(b) ? layered_pdf(b, wo, wi, MODE_RADIANCE) : base_pdf(b, wo, wi, sample_flags);
}
SHM_HD void bxdf_regularize(BxDF& b) {
    if (b.kind == SHM_MATERIAL_CONDUCTOR || b.kind == SHM_MATERIAL_DIELECTRIC || bxdf_is_layered(b)) b.mf.regularize();
    if (b.kind == SHM_MATERIAL_COATED_CONDUCTOR) b.mf2.regularize();
}

// ---- BSDF, bsdf.rs:9-111 ----
struct BSDF {
    BxDF bxdf;
    Frame shading_frame;
};
SHM_HD BSDF bsdf_new(V3 ns, V3 dpdus, const BxDF& bxdf) {  // bsdf.rs:22-28
    BSDF b;
    b.bxdf = bxdf;
    b.shading_frame = frame_from_xz(normalize(dpdus), ns);
    return b;
}
SHM_HD uint32_t bsdf_flags(const BSDF& b) { return bxdf_flags(b.bxdf); }
SHM_HD Spec bsdf_f(const BSDF& b, V3 wo_render, V3 wi_render) {  // bsdf.rs:44-58
    V3 wi = b.shading_frame.to_local(wi_render);
    V3 wo = b.shading_frame.to_local(wo_render);
    if (wo.z == 0.0f) return spec_const(0.0f);
    return bxdf_f(b.bxdf, wo, wi);
}
SHM_HD bool bsdf_sample_f(const BSDF& b, V3 wo_render, Float u, V2 u2, uint32_t sample_flags, BSDFSample& bs) {  // bsdf.rs:60-82
    V3 wo = b.shading_frame.to_local(wo_render);
    if (wo.z == 0.0f || !((bxdf_flags(b.bxdf) & sample_flags) != 0)) return false;
    if (!bxdf_sample_f(b.bxdf, wo, u, u2, sample_flags, bs)) return false;
    if (is_zero(bs.f) || bs.pdf == 0.0f || bs.wi.z == 0.0f) return false;
    bs.wi = b.shading_frame.from_local(bs.wi);
    return true;
}
// options.force_diffuse (interaction.rs:256-275): the BSDF is replaced by a DiffuseBxDF whose reflectance is the one-sample
// estimate BSDF::rho_hd(wo, [uc], [u2]) (bsdf.rs:99-102 -> bxdf.rs:49-71), in the same shading frame.
SHM_HD void bsdf_force_diffuse(BSDF& b, V3 wo_render, Float uc, V2 u2) {
    V3 wo = b.shading_frame.to_local(wo_render);
    Spec r = spec_const(0.0f);
    if (wo.z != 0.0f) {
        BSDFSample bs;
        // BxDF::sample_f directly (bxdf.rs:57-63): none of the rejections BSDF::sample_f adds (bsdf.rs:60-82)
        if (bxdf_sample_f(b.bxdf, wo, uc, u2, REFLTRANS_ALL, bs) && bs.pdf > 0.0f) r = r + bs.f * abs_cos_theta(bs.wi) / bs.pdf;
        r = r / 1.0f;
    }
    BxDF d = b.bxdf;
    d.kind = SHM_MATERIAL_DIFFUSE;
    d.r = r;
    b.bxdf = d;
}
SHM_HD Float bsdf_pdf(const BSDF& b, V3 wo_render, V3 wi_render, uint32_t sample_flags) {  // bsdf.rs:84-97
    V3 wo = b.shading_frame.to_local(wo_render);
    V3 wi = b.shading_frame.to_local(wi_render);
    if (wo.z == 0.0f) return 0.0f;
    return bxdf_pdf(b.bxdf, wo, wi, sample_flags);
}

}  // namespace shm
