// shm/spectrum.h — SampledSpectrum / SampledWavelengths / Spectrum lookups.
//
// Restates (paths relative to /root/reference/src):
//   spectra/mod.rs:17                       NUM_SPECTRUM_SAMPLES = 4
//   spectra/sampled_spectrum.rs:24-118      from_const, is_zero, safe_div, clamp, average, max_component_value
//   spectra/sampled_spectrum.rs:193-300     component-wise operators (s / v is a division)
//   spectra/sampled_wavelengths.rs:57-96    sample_visible, pdf, terminate_secondary, secondary_terminated
//   sampling.rs:268-278                     sample_visible_wavelengths, visible_wavelengths_pdf
//   spectra/spectrum.rs:143-166             ConstantSpectrum
//   spectra/spectrum.rs:264-291             DenselySampledSpectrum::get (truncating) / ::sample (round(), quirk 11)
//   spectra/spectrum.rs:398-428             PiecewiseLinearSpectrum::get / ::sample; math.rs:299-311 find_interval
#pragma once
#include "vec.h"
#include "../../../include/shimmer_hip.h"

namespace shm {

constexpr int NSPEC = 4;
constexpr Float LAMBDA_MIN = 360.0f;
constexpr Float LAMBDA_MAX = 830.0f;

struct Spec {
    Float v[NSPEC];
    SHM_HD Float operator[](int i) const { return v[i]; }
};
SHM_HD Spec spec_const(Float c) { Spec s; for (int i = 0; i < NSPEC; ++i) s.v[i] = c; return s; }
SHM_HD bool spec_is_finite(const Spec& s) { return is_finite(s.v[0]) && is_finite(s.v[1]) && is_finite(s.v[2]) && is_finite(s.v[3]); }
SHM_HD bool is_zero(const Spec& s) {
    for (int i = 0; i < NSPEC; ++i) if (s.v[i] != 0.0f) return false;
    return true;
}
SHM_HD Spec operator+(const Spec& a, const Spec& b) { Spec r; for (int i = 0; i < NSPEC; ++i) r.v[i] = a.v[i] + b.v[i]; return r; }
SHM_HD Spec operator-(const Spec& a, const Spec& b) { Spec r; for (int i = 0; i < NSPEC; ++i) r.v[i] = a.v[i] - b.v[i]; return r; }
SHM_HD Spec operator*(const Spec& a, const Spec& b) { Spec r; for (int i = 0; i < NSPEC; ++i) r.v[i] = a.v[i] * b.v[i]; return r; }
SHM_HD Spec operator/(const Spec& a, const Spec& b) { Spec r; for (int i = 0; i < NSPEC; ++i) r.v[i] = a.v[i] / b.v[i]; return r; }
SHM_HD Spec operator*(const Spec& a, Float f) { Spec r; for (int i = 0; i < NSPEC; ++i) r.v[i] = a.v[i] * f; return r; }
SHM_HD Spec operator*(Float f, const Spec& a) { Spec r; for (int i = 0; i < NSPEC; ++i) r.v[i] = a.v[i] * f; return r; }
SHM_HD Spec operator/(const Spec& a, Float f) { Spec r; for (int i = 0; i < NSPEC; ++i) r.v[i] = a.v[i] / f; return r; }
// sampled_spectrum.rs:39-50
SHM_HD Spec safe_div(const Spec& a, const Spec& b) {
    Spec r;
    for (int i = 0; i < NSPEC; ++i) r.v[i] = (b.v[i] != 0.0f) ? a.v[i] / b.v[i] : 0.0f;
    return r;
}
SHM_HD Spec clamp_zero(const Spec& a) { Spec r; for (int i = 0; i < NSPEC; ++i) r.v[i] = max(0.0f, a.v[i]); return r; }  // sampled_spectrum.rs:61-68
SHM_HD Spec spec_sqrt(const Spec& a) { Spec r; for (int i = 0; i < NSPEC; ++i) r.v[i] = sqrt(a.v[i]); return r; }          // sampled_spectrum.rs:312-319
SHM_HD Spec clamp(const Spec& a, Float lo, Float hi) { Spec r; for (int i = 0; i < NSPEC; ++i) r.v[i] = clamp(a.v[i], lo, hi); return r; }
// sampled_spectrum.rs:102-104: iter().sum() / len
SHM_HD Float average(const Spec& a) {
    Float s = 0.0f;
    for (int i = 0; i < NSPEC; ++i) s = s + a.v[i];
    return s / (Float)NSPEC;
}
// sampled_spectrum.rs:113-118
SHM_HD Float max_component_value(const Spec& a) {
    Float m = a.v[0];
    for (int i = 1; i < NSPEC; ++i) m = max(m, a.v[i]);
    return m;
}

struct Wavelengths {
    Float lambda[NSPEC];
    Float pdf[NSPEC];
};
// sampling.rs:268-270
SHM_HD Float sample_visible_wavelengths(Float u) { return 538.0f - 138.888889f * atanh(0.85691062f - 1.82750197f * u); }
// sampling.rs:272-278
SHM_HD Float visible_wavelengths_pdf(Float lambda) {
    if (lambda < 360.0f || lambda > 830.0f) return 0.0f;
    Float x = cosh(0.0072f * (lambda - 538.0f));
    return 0.0039398042f / (x * x);
}
// sampled_wavelengths.rs:57-71
SHM_HD Wavelengths sample_visible(Float u) {
    Wavelengths w;
    for (int i = 0; i < NSPEC; ++i) {
        Float up = u + (Float)i / (Float)NSPEC;
        if (up > 1.0f) up -= 1.0f;
        w.lambda[i] = sample_visible_wavelengths(up);
        w.pdf[i] = visible_wavelengths_pdf(w.lambda[i]);
    }
    return w;
}
// sampled_wavelengths.rs:89-96
SHM_HD bool secondary_terminated(const Wavelengths& w) {
    for (int i = 1; i < NSPEC; ++i) if (w.pdf[i] != 0.0f) return false;
    return true;
}
// sampled_wavelengths.rs:79-87
SHM_HD void terminate_secondary(Wavelengths& w) {
    if (secondary_terminated(w)) return;
    for (int i = 1; i < NSPEC; ++i) w.pdf[i] = 0.0f;
    w.pdf[0] /= (Float)NSPEC;
}
SHM_HD Spec pdf_spec(const Wavelengths& w) { Spec s; for (int i = 0; i < NSPEC; ++i) s.v[i] = w.pdf[i]; return s; }

// math.rs:299-311
template <typename Pred>
SHM_HD int find_interval(int size, Pred pred) {
    int first = 1;
    int last = size - 2;
    while (last > 0) {
        int half = last >> 1;
        int middle = first + half;
        bool pr = pred(middle);
        first = pr ? middle + 1 : first;
        last = pr ? last - (half + 1) : half;
    }
    int r = first - 1;
    if (r < 0) r = 0;
    if (r > size - 2) r = size - 2;
    return r;
}

// RgbSigmoidPolynomial::get, color.rs:353-383: s(poly(lambda, [c2, c1, c0])). `poly` is the fast_polynomial 0.1.0 crate (source
// not vendored: parity unpinned at that boundary); a three-coefficient Estrin / Horner evaluation with FMAs is
// fma(x^2, c0, fma(x, c1, c2)), which is what is done here.
// fast_polynomial::poly(x, [k0, k1, k2]) = k0 + k1 x + k2 x^2 (math.rs:563-570 pins poly(2, [1, 2, 3]) = 17), in the order defined above
SHM_HD Float poly3(Float x, Float k0, Float k1, Float k2) { return fma(x * x, k2, fma(x, k1, k0)); }
SHM_HD Float rgb_sigmoid(const Float c[3], Float lambda) {
    Float x = poly3(lambda, c[2], c[1], c[0]);
    if (is_inf(x)) return x > 0.0f ? 1.0f : 0.0f;
    return 0.5f + x / (2.0f * sqrt(1.0f + x * x));
}
SHM_HD Float dense_get(const ShmSpectrum& s, const Float* data, Float lambda) {  // spectrum.rs:265-271
    int offset = (int)lambda - s.lambda_min;
    if (offset < 0 || offset >= (int)s.n) return 0.0f;
    return data[s.offset + offset];
}

// Spectrum::get(lambda)
SHM_HD Float spectrum_get(const ShmSpectrum& s, const Float* data, Float lambda) {
    if (s.kind == SHM_SPECTRUM_CONSTANT) return s.c;
    if (s.kind == SHM_SPECTRUM_RGB_ALBEDO) return rgb_sigmoid(s.rgb_c, lambda);                                   // spectrum.rs:513-515
    if (s.kind == SHM_SPECTRUM_RGB_UNBOUNDED) return s.c * rgb_sigmoid(s.rgb_c, lambda);                           // :550-552
    if (s.kind == SHM_SPECTRUM_RGB_ILLUMINANT) return s.c * rgb_sigmoid(s.rgb_c, lambda) * dense_get(s, data, lambda);  // :592-594
    if (s.kind == SHM_SPECTRUM_DENSE) {
        // spectrum.rs:265-271: `lambda as i32` truncates toward zero (saturating)
        int offset = (int)lambda - s.lambda_min;
        if (offset < 0 || offset >= (int)s.n) return 0.0f;
        return data[s.offset + offset];
    }
    // PiecewiseLinear, spectrum.rs:399-415
    const Float* lambdas = data + s.offset;
    const Float* values = data + s.offset + s.n;
    if (s.n == 0 || lambda < lambdas[0] || lambda > lambdas[s.n - 1]) return 0.0f;
    int o = find_interval((int)s.n, [&](int i) { return lambdas[i] <= lambda; });
    Float t = (lambda - lambdas[o]) / (lambdas[o + 1] - lambdas[o]);
    return lerp(t, values[o], values[o + 1]);
}
// Spectrum::sample(lambda)
SHM_HD Spec spectrum_sample(const ShmSpectrum& s, const Float* data, const Wavelengths& w) {
    Spec r;
    if (s.kind == SHM_SPECTRUM_DENSE || s.kind == SHM_SPECTRUM_RGB_ILLUMINANT) {
        // spectrum.rs:280-291: nearest-nm lookup through round() (reference quirk 11)
        for (int i = 0; i < NSPEC; ++i) {
            int offset = (int)round(w.lambda[i]) - s.lambda_min;
            r.v[i] = (offset < 0 || offset >= (int)s.n) ? 0.0f : data[s.offset + offset];
        }
        if (s.kind == SHM_SPECTRUM_DENSE) return r;
        // RgbIlluminantSpectrum::sample, spectrum.rs:600-606: (scale * rsp) per wavelength, times the illuminant's sample
        Spec q;
        for (int i = 0; i < NSPEC; ++i) q.v[i] = s.c * rgb_sigmoid(s.rgb_c, w.lambda[i]);
        return q * r;
    }
    for (int i = 0; i < NSPEC; ++i) r.v[i] = spectrum_get(s, data, w.lambda[i]);
    return r;
}
// Raw 471-entry table sample (PixelSensor bars, film.rs:907-914 via DenselySampledSpectrum::sample)
SHM_HD Spec dense_table_sample(const Float* table, const Wavelengths& w) {
    Spec r;
    for (int i = 0; i < NSPEC; ++i) {
        int offset = (int)round(w.lambda[i]) - 360;
        r.v[i] = (offset < 0 || offset >= 471) ? 0.0f : table[offset];
    }
    return r;
}

}  // namespace shm
