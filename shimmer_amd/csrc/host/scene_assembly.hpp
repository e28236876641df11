// scene_assembly.hpp — C++ scene assembly for the PBRT-v4 loader (host/pbrt_loader.cpp): what the reference does between its parsed
// entities and the objects create_integrator receives, flattened into the ABI's ShmSceneDesc. Restates (paths relative to
// /root/reference/src):
//   spectra/spectrum.rs:179-196, 313-372, 430-489, 617-631  DenselySampledSpectrum::new, PiecewiseLinearSpectrum::{new, from_interleaved},
//                                                            BlackbodySpectrum, spectrum_to_photometric
//   spectra/named_spectrum.rs:30-46                          the named metal / glass / illuminant spectra
//   shape/mesh.rs:22-70, shape/triangle.rs:507-510, shape/sphere.rs:40-92, 275-280, shape/bilinear_patch.rs:430-433   meshes in render space, bounds
//   light.rs:560-612, 424-452, 112-160                       DiffuseAreaLight / PointLight / infinite light creation (scale / photometric)
//   loading/scene.rs:609-624, 721-886                        one area light per emissive shape; BvhAggregate per object definition + instances
//   aggregate.rs:207-467 (shm_bvh_build), camera.rs (shm_camera_perspective / _orthographic), film.rs:225-330, 767-800
// Host-side only. The same numerics are reachable from Python through shm_blackbody_dense / shm_look_at, so that the generators of
// shimmer_amd/scenes.py and this loader hand bit-identical inputs to the library.
#pragma once
#include <cmath>
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/shimmer_hip.h"
#include "../shm/path.h"  // (texture.h) rgb2spec_fetch: RgbColorSpace::to_rgb_coeffs on the host, the same code the device runs for image textures
#include "pbrt_math.hpp"

namespace pbrt {

#include "spectral_tables.inc"

struct LoadError : std::runtime_error {
    int code;
    LoadError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
[[noreturn]] inline void fail(const std::string& m, int code = SHM_ERR_INVALID_ARGUMENT) { throw LoadError(code, m); }

inline std::vector<float> table(const uint32_t* bits, size_t n) {
    std::vector<float> v(n);
    memcpy(v.data(), bits, n * sizeof(float));
    return v;
}
#define PBRT_TABLE(NAME) pbrt::table(pbrt::TBL_##NAME, sizeof(pbrt::TBL_##NAME) / sizeof(uint32_t))

// PiecewiseLinearSpectrum::from_interleaved (spectrum.rs:324-372) without the normalisation -> (lambda[], value[])
inline void from_interleaved(const std::vector<float>& s, std::vector<float>& lam, std::vector<float>& val) {
    lam.clear(); val.clear();
    if (s[0] > 360.0f) { lam.push_back(359.0f); val.push_back(s[1]); }
    for (size_t i = 0; i + 1 < s.size(); i += 2) { lam.push_back(s[i]); val.push_back(s[i + 1]); }
    if (lam.back() < 830.0f) { lam.push_back(831.0f); val.push_back(val.back()); }
}
inline float piecewise_get(const std::vector<float>& lam, const std::vector<float>& val, float x) {  // spectrum.rs:374-392
    if (x < lam.front() || x > lam.back()) return 0.0f;
    // find_interval: the last i with lam[i] <= x, clamped to [0, n - 2]
    size_t lo = 0, hi = lam.size();
    while (lo < hi) { size_t mid = (lo + hi) / 2; if (lam[mid] <= x) lo = mid + 1; else hi = mid; }
    long o = (long)lo - 1;
    if (o < 0) o = 0;
    if (o > (long)lam.size() - 2) o = (long)lam.size() - 2;
    const float t = (x - lam[o]) / (lam[o + 1] - lam[o]);
    return val[o] * (1.0f - t) + val[o + 1] * t;  // lerp(t, a, b) = a (1 - t) + b t (math.rs:246-252)
}
inline std::vector<float> piecewise_to_dense(const std::vector<float>& lam, const std::vector<float>& val) {
    std::vector<float> d(471);
    for (int l = 360; l <= 830; ++l) d[l - 360] = piecewise_get(lam, val, (float)l);
    return d;
}
// DenselySampledSpectrum::new(BlackbodySpectrum::new(T)) at 360..=830 nm, f32 as the reference (spectrum.rs:430-489)
inline std::vector<float> blackbody_dense(float t) {
    auto bb = [t](float lam_nm) {
        const float c = 299792458.0f, h = 6.62606957e-34f, kb = 1.3806488e-23f;
        const float l = lam_nm * 1e-9f;
        const float l5 = l * l * l * l * l;
        const float e = expf((h * c) / (l * kb * t));
        return (2.0f * h * c * c) / (l5 * (e - 1.0f));
    };
    const float lambda_max = 2.8977721e-3f / t;
    const float norm = 1.0f / bb(lambda_max * 1e9f);
    std::vector<float> d(471);
    for (int l = 360; l <= 830; ++l) d[l - 360] = bb((float)l) * norm;
    return d;
}
inline float spectrum_to_photometric(const std::vector<float>& dense) {  // spectrum.rs:617-631 over a 360..=830 table
    const std::vector<float> y = PBRT_TABLE(CIE_Y);
    float acc = 0.0f;
    for (int i = 0; i < 471; ++i) acc += y[i] * dense[i];
    return acc;
}
// Spectrum::get_named_spectrum(StdIllumD65): from_interleaved(CIE_ILLUM_D6500, normalize = true), densely sampled
inline std::vector<float> illuminant_d65_dense() {
    std::vector<float> lam, val;
    from_interleaved(PBRT_TABLE(CIE_ILLUM_D6500), lam, val);
    std::vector<float> dense = piecewise_to_dense(lam, val);
    const std::vector<float> y = PBRT_TABLE(CIE_Y);
    float integral = 0.0f;
    for (int i = 0; i < 471; ++i) integral += dense[i] * y[i];
    const std::vector<float> yi = PBRT_TABLE(CIE_Y_INTEGRAL);
    const float k = yi[0] / integral;
    for (float& v : val) v *= k;
    return piecewise_to_dense(lam, val);
}

// Spectrum::get_named_spectrum(IllumAcesD60): from_interleaved(ACES_ILLUM_D60, normalize = true) (named_spectrum.rs:58-62), densely sampled
inline std::vector<float> illuminant_aces_d60_dense() {
    std::vector<float> lam, val;
    from_interleaved(PBRT_TABLE(ACES_ILLUM_D60), lam, val);
    std::vector<float> dense = piecewise_to_dense(lam, val);
    const std::vector<float> y = PBRT_TABLE(CIE_Y);
    float integral = 0.0f;
    for (int i = 0; i < 471; ++i) integral += dense[i] * y[i];
    const std::vector<float> yi = PBRT_TABLE(CIE_Y_INTEGRAL);
    const float k = yi[0] / integral;
    for (float& v : val) v *= k;
    return piecewise_to_dense(lam, val);
}
// The named colour spaces (colorspace.rs:117-164: NamedColorSpace, SRGB / REC_2020 / ACES2065_1 with their primaries, illuminant and Gamut) and
// where each one's rgb2spec coefficient table comes from (rgb_to_spectra.rs:27-45: rgbtospec/srgb.spec, rec2020.spec, aces2065_1.spec)
enum ColorSpaceId : int { CS_SRGB = 0, CS_REC2020 = 1, CS_ACES2065_1 = 2, N_COLOR_SPACES = 3 };
struct ColorSpaceDef {
    const char* name;        // the ColorSpace directive's argument (matched case-insensitively, colorspace.rs:125-131)
    const char* spec_file;   // the reference's file name under rgbtospec/
    const char* env;         // explicit override
    const char* generated;   // what tools/gen_rgb2spec.py writes into shimmer_amd/data/
    float xy[3][2];          // r, g, b primaries
    bool aces_d60;           // illuminant: ACES D60 instead of D65
};
inline const ColorSpaceDef& color_space_def(int cs) {
    static const ColorSpaceDef k[N_COLOR_SPACES] = {
        {"srgb", "srgb.spec", "SHM_RGB2SPEC_SRGB", "rgb2spec_srgb_res64.spec", {{0.64f, 0.33f}, {0.3f, 0.6f}, {0.15f, 0.06f}}, false},
        {"rec2020", "rec2020.spec", "SHM_RGB2SPEC_REC2020", "rgb2spec_rec2020_res64.spec", {{0.708f, 0.292f}, {0.170f, 0.797f}, {0.131f, 0.046f}}, false},
        {"aces2065-1", "aces2065_1.spec", "SHM_RGB2SPEC_ACES2065_1", "rgb2spec_aces2065_1_res64.spec", {{0.7347f, 0.2653f}, {0.0f, 1.0f}, {0.0001f, -0.077f}}, true}};
    return k[cs];
}
inline int color_space_from_name(std::string n) {
    for (char& c : n) c = (char)tolower((unsigned char)c);
    for (int cs = 0; cs < N_COLOR_SPACES; ++cs) if (n == color_space_def(cs).name) return cs;
    return -1;
}
inline std::vector<float> illuminant_dense(int cs) { return color_space_def(cs).aces_d60 ? illuminant_aces_d60_dense() : illuminant_d65_dense(); }

// A spectrum value as parsed, before it is bound to a use (material slot: kept in its own kind; light: densely sampled)
struct SpectrumValue {
    enum Kind { NONE, CONSTANT, PIECEWISE, DENSE, RGB, TEXTURE } kind = NONE;
    float c = 0.0f;
    std::vector<float> lam, val;   // PIECEWISE
    std::vector<float> dense;      // DENSE (blackbody)
    float rgb[3] = {0, 0, 0};
    int color_space = CS_SRGB;     // RGB: the colour space the value is given in (the parameter's: paramdict.rs:615-620)
    std::string texture;           // TEXTURE: name of a spectrum texture
    std::string key;               // identity for pooling (one table per distinct emission spectrum)
};
// paramdict.rs SpectrumType: what an "rgb" value (or an RGB image texture) becomes at the slot that reads it
enum SpectrumType { SPECTRUM_ALBEDO = SHM_SPECTRUM_TYPE_ALBEDO, SPECTRUM_UNBOUNDED = SHM_SPECTRUM_TYPE_UNBOUNDED, SPECTRUM_ILLUMINANT = SHM_SPECTRUM_TYPE_ILLUMINANT };

// the rgb2spec coefficient table of a colour space: the `.spec` file of Jakob & Hanika's rgb2spec (crate rgb2spec 0.1.1 RGB2Spec::load:
// "SPEC" magic, u32 resolution, f32 scale[res], f32 data[3 * res^3 * 3], little-endian; rgb_to_spectra.rs:27-31 loads rgbtospec/srgb.spec)
struct Rgb2SpecTable {
    uint32_t res = 0;
    std::vector<float> scale, data;
};
inline bool rgb2spec_load(const std::string& path, Rgb2SpecTable& t) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char magic[4];
    uint32_t res = 0;
    bool ok = fread(magic, 1, 4, f) == 4 && fread(&res, 4, 1, f) == 1 && memcmp(magic, "SPEC", 4) == 0 && res >= 2 && res <= 1024;
    if (ok) {
        t.res = res;
        t.scale.resize(res);
        t.data.resize((size_t)3 * res * res * res * 3);
        ok = fread(t.scale.data(), sizeof(float), t.scale.size(), f) == t.scale.size() && fread(t.data.data(), sizeof(float), t.data.size(), f) == t.data.size();
    }
    fclose(f);
    if (!ok) fail(path + ": not an rgb2spec coefficient file (\"SPEC\", resolution, scale[res], data[3][res][res][res][3])");
    return true;
}

class Assembly {
public:
    struct Mesh {
        std::vector<float> p, n, s, uv;
        std::vector<uint32_t> vi;
        bool reverse = false, swaps = false;
    };
    struct Prim {
        uint32_t kind, index, material;
        int32_t light;
        uint32_t owner;  // 0 = the scene, k = object definition k
        float b[6];
    };
    std::vector<float> spec;
    std::vector<ShmMaterial> materials;
    std::vector<ShmLight> lights;
    std::vector<Mesh> meshes, patch_meshes;
    std::vector<ShmSphere> spheres;
    std::vector<Prim> prims;
    std::vector<ShmFloatTexture> float_textures;
    std::vector<ShmSpectrumTexture> spectrum_textures;
    std::map<std::string, uint32_t> objects;
    struct Inst { uint32_t object; M4 m; };
    std::vector<Inst> instances;
    uint32_t n_tris = 0, n_patches = 0;
    ShmCamera camera;
    ShmFilm film;
    bool have_camera = false;
    std::vector<float> sensor_x = PBRT_TABLE(CIE_X), sensor_y = PBRT_TABLE(CIE_Y), sensor_z = PBRT_TABLE(CIE_Z);
    std::map<std::string, std::pair<ShmSpectrum, float>> emission_cache;  // key -> (pooled dense table, photometric integral)
    // image textures / lights and the colour space they (and "rgb" parameters) need
    std::vector<ShmImageTexture> image_textures;
    std::vector<ShmImageLevel> image_levels;
    std::vector<float> texels;
    std::vector<ShmImageInfiniteLight> image_lights;
    std::vector<float> ewa_lut = PBRT_TABLE(MIP_FILTER_LUT);
    Rgb2SpecTable rgb2spec_tables[N_COLOR_SPACES];    // per named colour space, loaded the first time something needs it
    std::vector<float> cs_illuminants[N_COLOR_SPACES];  // RgbColorSpace::illuminant, densely sampled (colorspace.rs:63-65): StdIllum-D65 / ACES D60
    uint32_t cs_illuminant_offsets[N_COLOR_SPACES] = {~0u, ~0u, ~0u};
    // what the DEVICE gets (ShmSceneDesc::color_space): the table and illuminant of RGB IMAGES — always sRGB, the colour space read_png gives
    // every RGB file (image.rs:1262-1270); "rgb" PARAMETERS are converted on the host with their own colour space's table
    const Rgb2SpecTable& rgb2spec() const { return rgb2spec_tables[CS_SRGB]; }
    const std::vector<float>& cs_illuminant() const { return cs_illuminants[CS_SRGB]; }
    std::string scene_dir, lib_data_dir;    // where the tables are looked for: <scene dir>/rgbtospec, ./rgbtospec (the reference's place), then beside this library
    void need_color_space(int cs = CS_SRGB) {
        Rgb2SpecTable& t = rgb2spec_tables[cs];
        if (t.res) return;
        const ColorSpaceDef& def = color_space_def(cs);
        std::vector<std::string> search;
        if (const char* e = getenv(def.env)) search.push_back(e);
        if (!scene_dir.empty()) search.push_back(scene_dir + "/rgbtospec/" + def.spec_file);
        search.push_back(std::string("rgbtospec/") + def.spec_file);
        if (!lib_data_dir.empty()) search.push_back(lib_data_dir + "/" + def.generated);
        std::string tried;
        for (const std::string& p : search) {
            if (p.empty()) continue;
            if (rgb2spec_load(p, t)) break;
            tried += (tried.empty() ? "" : ", ") + p;
        }
        if (!t.res)
            fail(std::string("\"rgb\" values and RGB images need the ") + def.name + " rgb2spec coefficient table (the reference loads rgbtospec/" + def.spec_file +
                     ", rgb_to_spectra.rs:27-45); not found in: " + tried + " — set " + def.env + " or run tools/gen_rgb2spec.py", SHM_ERR_UNSUPPORTED);
        cs_illuminants[cs] = illuminant_dense(cs);
    }
    // RgbColorSpace::to_rgb_coeffs (colorspace.rs:95-98 -> rgb_to_spectra.rs:16-25 -> RGB2Spec::fetch)
    void rgb_coeffs(const float rgb[3], float out[3], int cs = CS_SRGB) {
        need_color_space(cs);
        shm::SceneView sv;
        memset(&sv, 0, sizeof(sv));
        sv.rgb2spec_res = rgb2spec_tables[cs].res;
        sv.rgb2spec_scale = rgb2spec_tables[cs].scale.data();
        sv.rgb2spec_data = rgb2spec_tables[cs].data.data();
        shm::rgb2spec_fetch(sv, shm::rgb3(rgb[0], rgb[1], rgb[2]), out);
    }
    // RgbAlbedoSpectrum / RgbUnboundedSpectrum / RgbIlluminantSpectrum::new (spectrum.rs:502-509, 536-547, 574-588)
    ShmSpectrum spec_rgb(const float rgb[3], SpectrumType type, int cs = CS_SRGB) {
        if (rgb[0] < 0.0f || rgb[1] < 0.0f || rgb[2] < 0.0f) fail("RGB parameter has negative component");  // paramdict.rs:628-633
        ShmSpectrum s;
        memset(&s, 0, sizeof(s));
        if (type == SPECTRUM_ALBEDO) {
            if (rgb[0] > 1.0f || rgb[1] > 1.0f || rgb[2] > 1.0f) fail("RGB parameter has component value > 1.0");  // paramdict.rs:641-646
            s.kind = SHM_SPECTRUM_RGB_ALBEDO;
            rgb_coeffs(rgb, s.rgb_c, cs);
            return s;
        }
        const float m = std::max(std::max(rgb[0], rgb[1]), rgb[2]);
        const float scale = 2.0f * m;
        const float scaled[3] = {scale != 0.0f ? rgb[0] / scale : 0.0f, scale != 0.0f ? rgb[1] / scale : 0.0f, scale != 0.0f ? rgb[2] / scale : 0.0f};
        rgb_coeffs(scaled, s.rgb_c, cs);
        s.c = scale;
        if (type == SPECTRUM_UNBOUNDED) { s.kind = SHM_SPECTRUM_RGB_UNBOUNDED; return s; }
        s.kind = SHM_SPECTRUM_RGB_ILLUMINANT;  // times the colour space's illuminant, which the spectrum carries (pooled once per colour space)
        if (cs_illuminant_offsets[cs] == ~0u) cs_illuminant_offsets[cs] = pool(cs_illuminants[cs]);
        s.offset = cs_illuminant_offsets[cs];
        s.n = 471;
        s.lambda_min = 360;
        return s;
    }
    // ---- RgbFilm::new's output matrix (film.rs:482-545, 767-850; colorspace.rs:38-72; color.rs:392-416; spectrum.rs:215-262) ----
    // xyz_from_rgb of sRGB as RgbColorSpace::new builds it from the primaries and W = XYZ::from_spectrum(D65): rgb * diag(rgb^-1 W). The
    // reference's 3x3 algebra is compensated f32; here f64, rounded once at the end.
    static void cs_matrices(int cs, const std::vector<float>& d65_dense, double xyz_from_rgb[3][3], double rgb_from_xyz[3][3], double white_xy[2]) {
        const std::vector<float> bx = PBRT_TABLE(CIE_X), by = PBRT_TABLE(CIE_Y), bz = PBRT_TABLE(CIE_Z), yi = PBRT_TABLE(CIE_Y_INTEGRAL);
        const std::vector<float>* bars[3] = {&bx, &by, &bz};
        double w[3];
        for (int k = 0; k < 3; ++k) {
            float acc = 0.0f;
            for (int i = 0; i < 471; ++i) acc += (*bars[k])[i] * d65_dense[i];
            w[k] = (double)(acc / yi[0]);
        }
        white_xy[0] = w[0] / (w[0] + w[1] + w[2]);
        white_xy[1] = w[1] / (w[0] + w[1] + w[2]);
        const float (*xy)[2] = color_space_def(cs).xy;  // XYZ::from_xy_y_default of each primary (color.rs:242-252): (x / y, 1, (1 - x - y) / y)
        double m[3][3], inv[3][3];
        for (int j = 0; j < 3; ++j) { m[0][j] = xy[j][0] * 1.0f / xy[j][1]; m[1][j] = 1.0; m[2][j] = (1.0f - xy[j][0] - xy[j][1]) * 1.0f / xy[j][1]; }
        invert3(m, inv);
        double c[3];
        for (int k = 0; k < 3; ++k) c[k] = inv[k][0] * w[0] + inv[k][1] * w[1] + inv[k][2] * w[2];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) xyz_from_rgb[i][j] = m[i][j] * c[j];
        invert3(xyz_from_rgb, rgb_from_xyz);
    }
    static void invert3(const double m[3][3], double inv[3][3]) {
        const double det = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) + m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
        inv[0][0] = (m[1][1] * m[2][2] - m[1][2] * m[2][1]) / det; inv[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) / det; inv[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) / det;
        inv[1][0] = (m[1][2] * m[2][0] - m[1][0] * m[2][2]) / det; inv[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) / det; inv[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) / det;
        inv[2][0] = (m[1][0] * m[2][1] - m[1][1] * m[2][0]) / det; inv[2][1] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) / det; inv[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) / det;
    }
    // DenselySampledSpectrum::d(temperature) (spectrum.rs:215-262), with the reference's `2.9678e6 / cct * cct` as written
    static std::vector<float> d_illuminant_dense(float temperature) {
        const float cct = temperature * 1.4388f / 1.4380f;
        if (cct < 4000.0f) {  // "CIE D ill-defined, use blackbody": BlackbodySpectrum::new(cct).get at the integer wavelengths
            return blackbody_dense(cct);
        }
        const float x = cct <= 7000.0f ? -4.607f * 1e9f / (cct * cct * cct) + 2.9678f * 1e6f / cct * cct + 0.09911f * 1e3f / cct + 0.244063f
                                      : -2.0064f * 1e9f / (cct * cct * cct) + 1.9018f * 1e6f / cct * cct + 0.24748f * 1e3f / cct + 0.23704f;
        const float y = -3.0f * x * x + 2.870f * x - 0.275f;
        const float m = 0.0241f + 0.2562f * x - 0.7341f * y;
        const float m1 = (-1.3515f - 1.7703f * x + 5.9114f * y) / m;
        const float m2 = (0.0300f - 31.4424f * x + 30.0717f * y) / m;
        const std::vector<float> lam = PBRT_TABLE(CIE_S_LAMBDA), s0 = PBRT_TABLE(CIE_S0), s1 = PBRT_TABLE(CIE_S1), s2 = PBRT_TABLE(CIE_S2);
        std::vector<float> val(lam.size());
        for (size_t i = 0; i < lam.size(); ++i) val[i] = (s0[i] + s1[i] * m1 + s2[i] * m2) * 0.01f;
        return piecewise_to_dense(lam, val);
    }
    // film.rs:524: output_rgb_from_sensor_rgb = color_space.rgb_from_xyz * sensor.xyz_from_sensor_rgb, the sensor's matrix being the von Kries
    // white balance (color.rs:404-416) from the "whitebalance" illuminant's white to the colour space's, or the identity without one
    static void film_output_matrix(float white_balance_temp, float out9[9], int cs = CS_SRGB) {
        const std::vector<float> d65 = illuminant_dense(cs);  // the film colour space's illuminant (D65, or ACES D60)
        double xyz_from_rgb[3][3], rgb_from_xyz[3][3], target_xy[2];
        cs_matrices(cs, d65, xyz_from_rgb, rgb_from_xyz, target_xy);
        double sensor[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
        if (white_balance_temp != 0.0f) {
            const std::vector<float> illum = d_illuminant_dense(white_balance_temp);
            double dummy1[3][3], dummy2[3][3], src_xy[2];
            cs_matrices(cs, illum, dummy1, dummy2, src_xy);  // (only its XYZ::from_spectrum(illum).xy() part is used)
            static const double lms_from_xyz[3][3] = {{0.8951, 0.2664, -0.1614}, {-0.7502, 1.7135, 0.0367}, {0.0389, -0.0685, 1.0296}};
            static const double xyz_from_lms[3][3] = {{0.986993, -0.147054, 0.159963}, {0.432305, 0.51836, 0.0492912}, {-0.00852866, 0.0400428, 0.968487}};
            auto from_xy = [](const double xy[2], double xyz[3]) { xyz[0] = xy[0] / xy[1]; xyz[1] = 1.0; xyz[2] = (1.0 - xy[0] - xy[1]) / xy[1]; };
            double sx[3], dx[3], sl[3], dl[3];
            from_xy(src_xy, sx);
            from_xy(target_xy, dx);
            for (int k = 0; k < 3; ++k) { sl[k] = lms_from_xyz[k][0] * sx[0] + lms_from_xyz[k][1] * sx[1] + lms_from_xyz[k][2] * sx[2]; dl[k] = lms_from_xyz[k][0] * dx[0] + lms_from_xyz[k][1] * dx[1] + lms_from_xyz[k][2] * dx[2]; }
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
                double acc = 0.0;
                for (int k = 0; k < 3; ++k) acc += xyz_from_lms[i][k] * (dl[k] / sl[k]) * lms_from_xyz[k][j];
                sensor[i][j] = acc;
            }
        }
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            double acc = 0.0;
            for (int k = 0; k < 3; ++k) acc += rgb_from_xyz[i][k] * sensor[k][j];
            out9[3 * i + j] = (float)acc;
        }
    }
    // RgbColorSpace::SRGB.luminance_vector(): row 1 of xyz_from_rgb (colorspace.rs:38-72, 107-114). With Y = 1 for the three primaries that row
    // is c = rgb^-1 W itself, W = XYZ::from_spectrum(D65) (f32 running sums as inner_product, spectrum.rs:609-615). The 3x3 inverse of the
    // reference is built from compensated difference_of_products; here it is evaluated in f64 and rounded once.
    void srgb_luminance_vector(float lum[3]) {
        need_color_space();
        float w[3];
        const std::vector<float>* bars[3] = {&sensor_x, &sensor_y, &sensor_z};
        const std::vector<float> yi = PBRT_TABLE(CIE_Y_INTEGRAL);
        for (int k = 0; k < 3; ++k) {
            float acc = 0.0f;
            for (int i = 0; i < 471; ++i) acc += (*bars[k])[i] * cs_illuminants[CS_SRGB][i];
            w[k] = acc / yi[0];
        }
        const float xy[3][2] = {{0.64f, 0.33f}, {0.3f, 0.6f}, {0.15f, 0.06f}};
        double m[3][3];
        for (int j = 0; j < 3; ++j) { m[0][j] = xy[j][0] * 1.0f / xy[j][1]; m[1][j] = 1.0; m[2][j] = (1.0f - xy[j][0] - xy[j][1]) * 1.0f / xy[j][1]; }
        const double det = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) + m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
        double inv[3][3];
        inv[0][0] = (m[1][1] * m[2][2] - m[1][2] * m[2][1]) / det; inv[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) / det; inv[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) / det;
        inv[1][0] = (m[1][2] * m[2][0] - m[1][0] * m[2][2]) / det; inv[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) / det; inv[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) / det;
        inv[2][0] = (m[1][0] * m[2][1] - m[1][1] * m[2][0]) / det; inv[2][1] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) / det; inv[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) / det;
        for (int k = 0; k < 3; ++k) lum[k] = (float)(inv[k][0] * w[0] + inv[k][1] * w[1] + inv[k][2] * w[2]);
    }
    // Spectrum::get of an RGB-derived spectrum at the 471 integer wavelengths (DenselySampledSpectrum::new of it: lights)
    std::vector<float> rgb_dense(const float rgb[3], SpectrumType type, int cs = CS_SRGB) {
        ShmSpectrum s = spec_rgb(rgb, type, cs);
        std::vector<float> d(471);
        for (int l = 360; l <= 830; ++l) d[l - 360] = shm::spectrum_get(s, spec.data(), (float)l);
        return d;
    }

    // ---- spectra ----
    static ShmSpectrum spec_constant(float c) {
        ShmSpectrum s;
        memset(&s, 0, sizeof(s));
        s.kind = SHM_SPECTRUM_CONSTANT;
        s.c = c;
        return s;
    }
    uint32_t pool(const std::vector<float>& v) {
        const uint32_t off = (uint32_t)spec.size();
        spec.insert(spec.end(), v.begin(), v.end());
        return off;
    }
    ShmSpectrum spec_dense(const std::vector<float>& v) {
        ShmSpectrum s;
        memset(&s, 0, sizeof(s));
        s.kind = SHM_SPECTRUM_DENSE;
        s.offset = pool(v);
        s.n = 471;
        s.lambda_min = 360;
        return s;
    }
    ShmSpectrum spec_piecewise(const std::vector<float>& lam, const std::vector<float>& val) {
        ShmSpectrum s;
        memset(&s, 0, sizeof(s));
        s.kind = SHM_SPECTRUM_PIECEWISE_LINEAR;
        std::vector<float> both(lam);
        both.insert(both.end(), val.begin(), val.end());
        s.offset = pool(both);
        s.n = (uint32_t)lam.size();
        return s;
    }
    // a material slot: constants and piecewise-linear spectra keep their kind; a blackbody becomes its dense table
    ShmSpectrum bind(const SpectrumValue& v, const std::map<std::string, ShmSpectrum>& spectrum_texture_names, SpectrumType type = SPECTRUM_ALBEDO) {
        switch (v.kind) {
            case SpectrumValue::CONSTANT: return spec_constant(v.c);
            case SpectrumValue::PIECEWISE: return spec_piecewise(v.lam, v.val);
            case SpectrumValue::DENSE: return spec_dense(v.dense);
            case SpectrumValue::TEXTURE: {
                auto it = spectrum_texture_names.find(v.texture);
                if (it == spectrum_texture_names.end()) fail("Couldn't find spectrum texture named \"" + v.texture + "\"");
                return it->second;
            }
            case SpectrumValue::RGB: return spec_rgb(v.rgb, type, v.color_space);
            default: fail("missing spectrum");
        }
    }
    // a light's emission: DenselySampledSpectrum::new(spectrum) (light.rs:525-527, 416-419)
    std::vector<float> dense_of(const SpectrumValue& v, SpectrumType type = SPECTRUM_ILLUMINANT) {
        switch (v.kind) {
            case SpectrumValue::CONSTANT: return std::vector<float>(471, v.c);
            case SpectrumValue::PIECEWISE: return piecewise_to_dense(v.lam, v.val);
            case SpectrumValue::DENSE: return v.dense;
            case SpectrumValue::RGB: return rgb_dense(v.rgb, type, v.color_space);
            default: fail("a light's spectrum cannot be a texture");
        }
    }

    // ---- shapes (vertices already in render space) ----
    void tri_bounds(const Mesh& m, uint32_t t, float b[6]) const {
        for (int k = 0; k < 3; ++k) { b[k] = INFINITY; b[3 + k] = -INFINITY; }
        for (int c = 0; c < 3; ++c) {
            const float* q = &m.p[3 * m.vi[3 * t + c]];
            for (int k = 0; k < 3; ++k) { b[k] = std::min(b[k], q[k]); b[3 + k] = std::max(b[3 + k], q[k]); }
        }
    }
    int area_light(uint32_t prim_index, float area, const SpectrumValue& L, float scale, bool two_sided) {
        ShmLight l;
        memset(&l, 0, sizeof(l));
        l.kind = SHM_LIGHT_DIFFUSE_AREA;
        l.primitive = prim_index;
        l.two_sided = two_sided ? 1u : 0u;
        l.area = area;
        auto it = emission_cache.find(L.key);
        if (it == emission_cache.end()) {
            const std::vector<float> dense = dense_of(L);
            it = emission_cache.emplace(L.key, std::make_pair(spec_dense(dense), spectrum_to_photometric(dense))).first;
        }
        l.scale = scale / it->second.second;  // light.rs:598: scale /= spectrum_to_photometric(L)
        l.spectrum = it->second.first;
        lights.push_back(l);
        return (int)lights.size() - 1;
    }
    struct Emission {
        bool on = false;
        SpectrumValue L;
        float scale = 1.0f, power = -1.0f;
        bool two_sided = false;
    };
    void add_mesh(Mesh&& mesh, uint32_t material, const Emission& em, uint32_t owner) {
        const uint32_t ntri = (uint32_t)(mesh.vi.size() / 3);
        const uint32_t base = n_tris;
        n_tris += ntri;
        meshes.push_back(std::move(mesh));
        const Mesh& m = meshes.back();
        for (uint32_t t = 0; t < ntri; ++t) {
            Prim pr{SHM_SHAPE_TRIANGLE, base + t, material, -1, owner, {0}};
            tri_bounds(m, t, pr.b);
            if (em.on) {
                const float* a = &m.p[3 * m.vi[3 * t]];
                const float* b = &m.p[3 * m.vi[3 * t + 1]];
                const float* c = &m.p[3 * m.vi[3 * t + 2]];
                const double e1[3] = {(double)b[0] - a[0], (double)b[1] - a[1], (double)b[2] - a[2]}, e2[3] = {(double)c[0] - a[0], (double)c[1] - a[1], (double)c[2] - a[2]};
                const double cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
                const float area = (float)(0.5 * std::sqrt(cx * cx + cy * cy + cz * cz));  // Triangle::area (only the light's `area` field: phi())
                float scale = em.scale;
                pr.light = area_light((uint32_t)prims.size(), area, em.L, scale, em.two_sided);
                if (em.power > 0.0f) lights.back().scale *= em.power / ((em.two_sided ? 2.0f : 1.0f) * area * 3.14159265358979323846f);  // light.rs:600-611
            }
            prims.push_back(pr);
        }
    }
    void add_patch_mesh(Mesh&& mesh, uint32_t material, const Emission& em, uint32_t owner) {
        const uint32_t n = (uint32_t)(mesh.vi.size() / 4);
        const uint32_t base = n_patches;
        n_patches += n;
        patch_meshes.push_back(std::move(mesh));
        const Mesh& m = patch_meshes.back();
        for (uint32_t t = 0; t < n; ++t) {
            Prim pr{SHM_SHAPE_BILINEAR_PATCH, base + t, material, -1, owner, {0}};
            for (int k = 0; k < 3; ++k) { pr.b[k] = INFINITY; pr.b[3 + k] = -INFINITY; }
            double q[4][3];
            for (int c = 0; c < 4; ++c) {
                const float* v = &m.p[3 * m.vi[4 * t + c]];
                for (int k = 0; k < 3; ++k) { pr.b[k] = std::min(pr.b[k], v[k]); pr.b[3 + k] = std::max(pr.b[3 + k], v[k]); q[c][k] = v[k]; }
            }
            if (em.on) {
                auto tri_area = [](const double* a, const double* b, const double* c) {
                    const double e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
                    const double cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
                    return 0.5 * std::sqrt(cx * cx + cy * cy + cz * cz);
                };
                const float area = (float)(tri_area(q[0], q[1], q[2]) + tri_area(q[3], q[1], q[2]));
                pr.light = area_light((uint32_t)prims.size(), area, em.L, em.scale, em.two_sided);
                if (em.power > 0.0f) lights.back().scale *= em.power / ((em.two_sided ? 2.0f : 1.0f) * area * 3.14159265358979323846f);
            }
            prims.push_back(pr);
        }
    }
    void add_sphere(float radius, float z_min_in, float z_max_in, float phi_max_deg, const Xf& render_from_object, bool reverse, uint32_t material, const Emission& em,
                    uint32_t owner) {
        ShmSphere s;
        memset(&s, 0, sizeof(s));
        auto clampf = [](float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); };
        const float zmin = std::min(z_min_in, z_max_in), zmax = std::max(z_min_in, z_max_in);
        s.radius = radius;
        s.z_min = clampf(zmin, -radius, radius);                                // sphere.rs:40-53
        s.z_max = clampf(zmax, -radius, radius);
        s.theta_z_min = acosf(clampf(zmin / radius, -1.0f, 1.0f));
        s.theta_z_max = acosf(clampf(zmax / radius, -1.0f, 1.0f));
        s.phi_max = (3.14159265358979323846f / 180.0f) * clampf(phi_max_deg, 0.0f, 360.0f);
        memcpy(s.render_from_object, render_from_object.m.m, sizeof(float) * 16);
        memcpy(s.object_from_render, render_from_object.inv.m, sizeof(float) * 16);
        s.reverse_orientation = reverse ? 1 : 0;
        s.transform_swaps_handedness = swaps_handedness(render_from_object.m) ? 1 : 0;
        spheres.push_back(s);
        Prim pr{SHM_SHAPE_SPHERE, (uint32_t)spheres.size() - 1, material, -1, owner, {0}};
        // Sphere::bounds = render_from_object.apply(Bounds3f{(-r, -r, z_min), (r, r, z_max)}) (sphere.rs:275-280; transform.rs:557-571: the eight corners)
        for (int k = 0; k < 3; ++k) { pr.b[k] = INFINITY; pr.b[3 + k] = -INFINITY; }
        for (int c = 0; c < 8; ++c) {
            const V3 q = shm::v3(c & 1 ? radius : -radius, c & 2 ? radius : -radius, c & 4 ? s.z_max : s.z_min);
            const M4& m = render_from_object.m;
            const float p[3] = {m.m[0][0] * q.x + m.m[0][1] * q.y + m.m[0][2] * q.z + m.m[0][3], m.m[1][0] * q.x + m.m[1][1] * q.y + m.m[1][2] * q.z + m.m[1][3],
                                m.m[2][0] * q.x + m.m[2][1] * q.y + m.m[2][2] * q.z + m.m[2][3]};
            for (int k = 0; k < 3; ++k) { pr.b[k] = std::min(pr.b[k], p[k]); pr.b[3 + k] = std::max(pr.b[3 + k], p[k]); }
        }
        if (em.on) {
            const float area = s.phi_max * s.radius * (s.z_max - s.z_min);  // sphere.rs:282-284
            pr.light = area_light((uint32_t)prims.size(), area, em.L, em.scale, em.two_sided);
            if (em.power > 0.0f) lights.back().scale *= em.power / ((em.two_sided ? 2.0f : 1.0f) * area * 3.14159265358979323846f);
        }
        prims.push_back(pr);
    }

    // ---- the flattened description (owns every array desc points to) ----
    struct Built {
        ShmSceneDesc desc;
        std::vector<ShmBvhNode> nodes;
        std::vector<ShmPrimitive> primitives;
        std::vector<ShmTriangleMesh> meshes;
        std::vector<ShmBilinearPatchMesh> patch_meshes;
        std::vector<ShmInstance> instances;
        std::unique_ptr<Assembly> owner;
    };
    // BvhAggregate::new per object definition and for the scene (loading/scene.rs:721-886), then the ABI's leaf-ordered arrays
    static std::unique_ptr<Built> build(std::unique_ptr<Assembly> self_owned) {
        Assembly& a = *self_owned;
        if (!a.have_camera) fail("the scene has no Camera");
        if (a.prims.empty()) fail("the scene has no shapes");
        std::unique_ptr<Built> out(new Built());
        const uint32_t n = (uint32_t)a.prims.size();
        const uint32_t n_objects = (uint32_t)a.objects.size();
        struct Tree { std::vector<ShmBvhNode> nodes; std::vector<uint32_t> order; };
        auto build_tree = [&](const std::vector<uint32_t>& idx) {
            Tree t;
            std::vector<float> b(6 * idx.size());
            for (size_t i = 0; i < idx.size(); ++i) memcpy(&b[6 * i], a.prims[idx[i]].b, sizeof(float) * 6);
            t.nodes.resize(2 * idx.size());
            std::vector<uint32_t> ord(idx.size());
            uint32_t cnt = 0;
            if (shm_bvh_build(b.data(), (uint32_t)idx.size(), 0, t.nodes.data(), &cnt, ord.data()) != SHM_OK) fail(std::string("BVH build failed: ") + shm_last_error());
            t.nodes.resize(cnt);
            t.order.resize(idx.size());
            for (size_t i = 0; i < idx.size(); ++i) t.order[i] = idx[ord[i]];
            return t;
        };
        std::vector<Tree> obj_trees(n_objects + 1);
        for (uint32_t k = 1; k <= n_objects; ++k) {
            std::vector<uint32_t> idx;
            for (uint32_t i = 0; i < n; ++i) if (a.prims[i].owner == k) idx.push_back(i);
            if (idx.empty()) fail("empty object definition");
            obj_trees[k] = build_tree(idx);
        }
        // an instance's bounds: the object's root bounds through render_from_instance (transform.rs:557-571)
        for (uint32_t ii = 0; ii < a.instances.size(); ++ii) {
            const ShmBvhNode& root = obj_trees[a.instances[ii].object].nodes[0];
            const M4& m = a.instances[ii].m;
            for (Prim& pr : a.prims) {
                if (pr.kind != SHM_SHAPE_INSTANCE || pr.index != ii) continue;
                for (int k = 0; k < 3; ++k) { pr.b[k] = INFINITY; pr.b[3 + k] = -INFINITY; }
                for (int c = 0; c < 8; ++c) {
                    const float q[3] = {c & 1 ? root.bmax[0] : root.bmin[0], c & 2 ? root.bmax[1] : root.bmin[1], c & 4 ? root.bmax[2] : root.bmin[2]};
                    for (int r = 0; r < 3; ++r) {
                        const float v = m.m[r][0] * q[0] + m.m[r][1] * q[1] + m.m[r][2] * q[2] + m.m[r][3];
                        pr.b[r] = std::min(pr.b[r], v);
                        pr.b[3 + r] = std::max(pr.b[3 + r], v);
                    }
                }
            }
        }
        std::vector<uint32_t> top;
        for (uint32_t i = 0; i < n; ++i) if (a.prims[i].owner == 0) top.push_back(i);
        if (top.empty()) fail("the scene has no shapes outside object definitions");
        obj_trees[0] = build_tree(top);
        std::vector<uint32_t> node_base(n_objects + 1), order;
        uint32_t nb = 0, pb = 0;
        for (uint32_t k = 0; k <= n_objects; ++k) {
            node_base[k] = nb;
            for (ShmBvhNode nd : obj_trees[k].nodes) {
                nd.offset += nd.n_prims > 0 ? pb : nb;
                out->nodes.push_back(nd);
            }
            nb += (uint32_t)obj_trees[k].nodes.size();
            pb += (uint32_t)obj_trees[k].order.size();
            order.insert(order.end(), obj_trees[k].order.begin(), obj_trees[k].order.end());
        }
        std::vector<uint32_t> slot_of_input(n);
        for (uint32_t s = 0; s < n; ++s) slot_of_input[order[s]] = s;
        out->primitives.resize(n);
        for (uint32_t s = 0; s < n; ++s) {
            const Prim& pr = a.prims[order[s]];
            out->primitives[s] = ShmPrimitive{pr.kind, pr.index, pr.material, pr.light};
        }
        for (ShmLight& l : a.lights) if (l.kind == SHM_LIGHT_DIFFUSE_AREA) l.primitive = slot_of_input[l.primitive];
        for (const Inst& in : a.instances) {
            ShmInstance si;
            memset(&si, 0, sizeof(si));
            memcpy(si.render_from_primitive, in.m.m, sizeof(float) * 16);
            M4 inv;
            if (!m4_inverse(in.m, inv)) fail("singular instance transform");
            memcpy(si.primitive_from_render, inv.m, sizeof(float) * 16);
            si.root_node = node_base[in.object];
            out->instances.push_back(si);
        }
        for (const Mesh& m : a.meshes) {
            ShmTriangleMesh mm;
            memset(&mm, 0, sizeof(mm));
            mm.n_triangles = (uint32_t)(m.vi.size() / 3);
            mm.n_vertices = (uint32_t)(m.p.size() / 3);
            mm.vertex_indices = m.vi.data();
            mm.p = m.p.data();
            mm.n = m.n.empty() ? nullptr : m.n.data();
            mm.s = m.s.empty() ? nullptr : m.s.data();
            mm.uv = m.uv.empty() ? nullptr : m.uv.data();
            mm.reverse_orientation = m.reverse;
            mm.transform_swaps_handedness = m.swaps;
            out->meshes.push_back(mm);
        }
        for (const Mesh& m : a.patch_meshes) {
            ShmBilinearPatchMesh pm;
            memset(&pm, 0, sizeof(pm));
            pm.n_patches = (uint32_t)(m.vi.size() / 4);
            pm.n_vertices = (uint32_t)(m.p.size() / 3);
            pm.vertex_indices = m.vi.data();
            pm.p = m.p.data();
            pm.n = m.n.empty() ? nullptr : m.n.data();
            pm.uv = m.uv.empty() ? nullptr : m.uv.data();
            pm.reverse_orientation = m.reverse;
            pm.transform_swaps_handedness = m.swaps;
            out->patch_meshes.push_back(pm);
        }
        if (a.spec.empty()) a.spec.push_back(0.0f);
        ShmSceneDesc& d = out->desc;
        memset(&d, 0, sizeof(d));
        d.abi_version = SHM_ABI_VERSION;
        d.n_nodes = (uint32_t)out->nodes.size(); d.nodes = out->nodes.data();
        d.n_primitives = n; d.primitives = out->primitives.data();
        d.n_meshes = (uint32_t)out->meshes.size(); d.meshes = out->meshes.data();
        d.n_spheres = (uint32_t)a.spheres.size(); d.spheres = a.spheres.data();
        d.n_materials = (uint32_t)a.materials.size(); d.materials = a.materials.data();
        d.n_lights = (uint32_t)a.lights.size(); d.lights = a.lights.data();
        d.n_spectrum_floats = (uint32_t)a.spec.size(); d.spectrum_data = a.spec.data();
        a.film.sensor_r_bar = a.sensor_x.data(); a.film.sensor_g_bar = a.sensor_y.data(); a.film.sensor_b_bar = a.sensor_z.data();  // PixelSensor::new cie1931 (film.rs:823-837)
        d.camera = a.camera;
        d.film = a.film;
        d.n_patch_meshes = (uint32_t)out->patch_meshes.size(); d.patch_meshes = out->patch_meshes.data();
        d.n_instances = (uint32_t)out->instances.size(); d.instances = out->instances.data();
        d.n_float_textures = (uint32_t)a.float_textures.size(); d.float_textures = a.float_textures.data();
        d.n_spectrum_textures = (uint32_t)a.spectrum_textures.size(); d.spectrum_textures = a.spectrum_textures.data();
        d.n_image_textures = (uint32_t)a.image_textures.size(); d.image_textures = a.image_textures.data();
        d.n_image_levels = (uint32_t)a.image_levels.size(); d.image_levels = a.image_levels.data();
        d.n_texel_floats = a.texels.size(); d.texel_data = a.texels.data();
        d.n_image_lights = (uint32_t)a.image_lights.size(); d.image_lights = a.image_lights.data();
        if (!a.image_textures.empty() || !a.image_lights.empty()) d.ewa_filter_lut = a.ewa_lut.data();
        if (a.rgb2spec().res) {  // ShmColorSpace (only when something asked for it: RGB images, image lights — the sRGB table, see rgb2spec())
            d.color_space.rgb2spec_res = a.rgb2spec().res;
            d.color_space.rgb2spec_scale = a.rgb2spec().scale.data();
            d.color_space.rgb2spec_data = a.rgb2spec().data.data();
            d.color_space.illuminant = a.cs_illuminant().data();
        }
        out->owner = std::move(self_owned);
        return out;
    }
};

}  // namespace pbrt
