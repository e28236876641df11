// shm/path.h — per-vertex building blocks of PathIntegrator::li: lights, material -> BSDF, camera rays, film.
//
// Restates (paths relative to /root/reference/src):
//   light.rs:984-1043             LightSampleContext
//   light.rs:392-497              PointLight::{sample_li,pdf_li}
//   light.rs:632-684              DiffuseAreaLight::{sample_li,pdf_li,l}
//   light.rs:747-803              UniformInfiniteLight::{sample_li,pdf_li,le}
//   light.rs:848-904              ImageInfinitelight::{sample_li,pdf_li,le} (arithmetic in texture.h)
//   light_sampler.rs:83-111       UniformLightSampler::{sample_light,pmf_light}
//   interaction.rs:187-278        SurfaceInteraction::get_bsdf (ray differentials are carried only in scenes with image
//                                 textures, texture.h: elsewhere they are dead values; bump_map with a constant
//                                 displacement is executed literally so that signed zeros match, material.rs:1477-1508)
//   material.rs:301-311,456-499,603-635,723-742   get_bxdf for Diffuse/Conductor/Dielectric/ThinDielectric
//   sampling.rs:347-371, filter.rs:99-105         get_camera_sample, BoxFilter::sample
//   camera.rs:1003-1079, 769-792  {Perspective,Orthographic}Camera::generate_ray_differential
//   film.rs:548-574, 907-914      RgbFilm::add_sample, PixelSensor::to_sensor_rgb
#pragma once
#include "bxdf.h"
#include "scene.h"
#include "texture.h"

namespace shm {

struct LightSampleContext {  // light.rs:984-999
    P3i pi;
    V3 n, ns;
    SHM_HD V3 p() const { return pi.mid(); }
};
SHM_HD LightSampleContext light_ctx_from(const SurfaceInteraction& si) {  // light.rs:1001-1009
    LightSampleContext c;
    c.pi = si.pi; c.n = si.n; c.ns = si.shading.n;
    return c;
}
struct LightLiSample {  // light.rs:1045-1065
    Spec l;
    V3 wi;
    Float pdf;
    P3i p_light_pi;  // p_light: Interaction {pi, n}
    V3 p_light_n;
};

SHM_HD bool light_is_delta(const ShmLight& l) { return l.kind == SHM_LIGHT_POINT; }  // light.rs:1098-1102

// DiffuseAreaLight::l, light.rs:668-684
SHM_HD Spec area_light_l(const SceneView& sv, const ShmLight& light, V3 n, V3 w, const Wavelengths& lambda) {
    if (!light.two_sided && dot(n, w) < 0.0f) return spec_const(0.0f);
    return light.scale * spectrum_sample(light.spectrum, sv.spectrum_data, lambda);
}

// Light::sample_li. PathIntegrator calls it with allow_incomplete_pdf = true (integrator.rs:927), SimplePathIntegrator with false
// (integrator.rs:651-655): the flag only matters for the uniform infinite light.
// HAS_TEX (scenes with image textures or an image infinite light: both need the colour-space tables) compiles the
// ImageInfinitelight branches in.
template <bool TRI_ONLY = false, bool HAS_TEX = false>
SHM_HD bool light_sample_li(const SceneView& sv, const ShmLight& light, const LightSampleContext& ctx, V2 u,
                            const Wavelengths& lambda, LightLiSample& out, bool allow_incomplete_pdf = true) {
    if (HAS_TEX && light.kind == SHM_LIGHT_IMAGE_INFINITE) {  // light.rs:848-880
        const ImageLightRec& il = sv.image_lights[light.primitive];
        Float map_pdf;
        V2 uv = pc2d_sample(sv.dist_data, allow_incomplete_pdf ? il.compensated : il.distribution, (int)il.n, u, map_pdf);
        if (map_pdf == 0.0f) return false;
        V3 w_light = equal_area_square_to_sphere(uv);
        V3 wi = xf_vector(il.render_from_light, w_light);
        out.l = light.scale * image_light_le_uv(sv, il, uv, lambda);
        out.wi = wi;
        out.pdf = map_pdf / (4.0f * PI_F);
        out.p_light_pi = p3i_exact(ctx.p() + wi * (2.0f * sv.scene_radius));
        out.p_light_n = v3s(0.0f);
        return true;
    }
    if (light.kind == SHM_LIGHT_POINT) {  // light.rs:452-468
        V3 p = ld3(light.position);
        V3 wi = normalize(p - ctx.p());
        Spec li = light.scale * spectrum_sample(light.spectrum, sv.spectrum_data, lambda) / distance_squared(p, ctx.p());
        out.l = li; out.wi = wi; out.pdf = 1.0f;
        out.p_light_pi = p3i_exact(p);
        out.p_light_n = v3s(0.0f);
        return true;
    }
    if (light.kind == SHM_LIGHT_UNIFORM_INFINITE) {
        if (allow_incomplete_pdf) return false;  // light.rs:747-749
        // light.rs:750-766: a uniform spherical sample, its pdf from uniform_hemisphere_pdf() (= 1/4pi there: quirk 2)
        V3 wi = sample_uniform_sphere(u);
        out.l = light.scale * spectrum_sample(light.spectrum, sv.spectrum_data, lambda);
        out.wi = wi;
        out.pdf = uniform_hemisphere_pdf(sv.quirks_off != 0);
        out.p_light_pi = p3i_exact(ctx.p() + wi * (2.0f * sv.scene_radius));
        out.p_light_n = v3s(0.0f);
        return true;
    }
    // DiffuseAreaLight::sample_li, light.rs:632-661
    ShapeSampleContext sctx;
    sctx.pi = ctx.pi; sctx.n = ctx.n; sctx.ns = ctx.ns;
    ShapeSample ss;
    const PrimRec& pr = light_prim_rec(sv, light);
    bool ok;
    if (!TRI_ONLY && (pr.kind_index & PRIM_SPHERE_BIT)) ok = sphere_sample_with_context(sv.spheres[pr.kind_index & PRIM_INDEX_MASK], sctx, u, ss);
    else if (!TRI_ONLY && (pr.kind_index & PRIM_PATCH_BIT)) ok = blp_sample_with_context(load_patch_rec(sv, pr), sctx, u, ss, sv.quirks_off != 0);  // (the emitter's record: the per-light copy, as for triangles)
    else ok = triangle_sample_with_context(load_triangle_rec(sv, pr), sctx, u, ss, sv.quirks_off != 0);
    if (!ok) return false;
    if (ss.pdf == 0.0f || length_squared(ss.pi.mid() - ctx.p()) == 0.0f) return false;
    V3 wi = normalize(ss.pi.mid() - ctx.p());
    Spec le = area_light_l(sv, light, ss.n, -wi, lambda);
    if (is_zero(le)) return false;
    out.l = le; out.wi = wi; out.pdf = ss.pdf;
    out.p_light_pi = ss.pi;
    out.p_light_n = ss.n;
    return true;
}
// Light::pdf_li with allow_incomplete_pdf = true
template <bool TRI_ONLY = false, bool HAS_TEX = false>
SHM_HD Float light_pdf_li(const SceneView& sv, const ShmLight& light, const LightSampleContext& ctx, V3 wi) {
    if (HAS_TEX && light.kind == SHM_LIGHT_IMAGE_INFINITE) {  // light.rs:882-892, the compensated distribution
        const ImageLightRec& il = sv.image_lights[light.primitive];
        V3 w_light = xf_vector(il.light_from_render, wi);
        V2 uv = equal_area_sphere_to_square(w_light);
        return pc2d_pdf(sv.dist_data, il.compensated, (int)il.n, uv) / (4.0f * PI_F);
    }
    if (light.kind != SHM_LIGHT_DIFFUSE_AREA) return 0.0f;  // light.rs:470-477, 774-775
    ShapeSampleContext sctx;
    sctx.pi = ctx.pi; sctx.n = ctx.n; sctx.ns = ctx.ns;
    const PrimRec& pr = light_prim_rec(sv, light);
    if (!TRI_ONLY && (pr.kind_index & PRIM_SPHERE_BIT)) return sphere_pdf_with_context(sv.spheres[pr.kind_index & PRIM_INDEX_MASK], sctx, wi, sv.quirks_off != 0);
    if (!TRI_ONLY && (pr.kind_index & PRIM_PATCH_BIT)) return blp_pdf_with_context(load_patch_rec(sv, pr), sctx, wi, sv.quirks_off != 0);
    return triangle_pdf_with_context(load_triangle_rec(sv, pr), sctx, wi);
}
// Light::le of an infinite light for an escaped ray: UniformInfiniteLight (light.rs:795-797) or ImageInfinitelight (:900-904)
template <bool HAS_TEX = false>
SHM_HD Spec infinite_light_le(const SceneView& sv, const ShmLight& light, V3 ray_d, const Wavelengths& lambda) {
    if (HAS_TEX && light.kind == SHM_LIGHT_IMAGE_INFINITE) return light.scale * image_light_le(sv, sv.image_lights[light.primitive], ray_d, lambda);
    return light.scale * spectrum_sample(light.spectrum, sv.spectrum_data, lambda);
}
// UniformLightSampler, light_sampler.rs:91-111. Returns light index or -1; p = 1/n.
SHM_HD int light_sampler_sample(const SceneView& sv, Float u, Float& p) {
    if (sv.n_lights == 0) return -1;
    uint32_t idx = (uint32_t)(u * (Float)sv.n_lights);
    if (idx > sv.n_lights - 1) idx = sv.n_lights - 1;
    p = 1.0f / (Float)sv.n_lights;
    return (int)idx;
}
SHM_HD Float light_sampler_pmf(const SceneView& sv) { return sv.n_lights == 0 ? 0.0f : 1.0f / (Float)sv.n_lights; }

// SurfaceInteraction::get_bsdf (interaction.rs:187-278).
// lambda is mutable: DielectricMaterial terminates secondary wavelengths for dispersive eta.
// HAS_TEX: the scene binds image textures; `df` is what compute_differentials (the first statement of the reference's get_bsdf,
// interaction.rs:197) left in the interaction. Without textures the differentials are dead values and df may be null.
template <bool HAS_TEX = false>
SHM_HD BSDF get_bsdf(const SceneView& sv, SurfaceInteraction& si, const ShmMaterial& m_in, Wavelengths& lambda, const Differentials* df = nullptr) {
    // Resolve mixed materials (interaction.rs:205-220, MixMaterial::choose_material material.rs:1308-1329). The reference takes
    // u from the tile's entropy-seeded rng; defined here as a hash of (wo, p, nesting level): reproducible, parity unpinned.
    // MaterialEvalContext::from(&*self) (interaction.rs:211): its TextureEvalContext part never changes below (p, n, uv, differentials)
    TextureEvalContext tctx;
    if (HAS_TEX) tctx = tex_ctx_from(si, *df);
    // a float parameter: the constant field, or the FloatTexture bound to the slot
    auto fval = [&](const ShmMaterial& mm, int slot, Float constant) {
        if (HAS_TEX && mm.float_tex[slot] != 0u) return float_texture_evaluate(sv, mm.float_tex[slot] - 1u, tctx);
        return constant;
    };
    const ShmMaterial* mp = &m_in;
    for (int level = 0; mp->kind == SHM_MATERIAL_MIX && level < 16; ++level) {
        Float amt = fval(*mp, SHM_FLOATSLOT_MIX_AMOUNT, mp->mix_amount);
        uint32_t pick;
        if (amt <= 0.0f) pick = 0;
        else if (amt >= 1.0f) pick = 1;
        else {
            uint64_t h = hash_v3(hash_v3(0x4d495800ULL + (uint64_t)level, si.wo), si.p());
            Float u = (Float)(uint32_t)(h >> 40) * 5.9604644775390625e-8f;  // 24 bits -> [0, 1)
            pick = (amt < u) ? 0u : 1u;
        }
        mp = &sv.materials[mp->mix_material[pick]];
    }
    const ShmMaterial& m = *mp;
    // interaction.rs:223-245: displacement (bump map) takes precedence over a normal map
    if (m.has_displacement) {
        V3 dpdu, dpdv;
        if (HAS_TEX && m.float_tex[SHM_FLOATSLOT_DISPLACEMENT] != 0u) {
            bump_map_texture(sv, m.float_tex[SHM_FLOATSLOT_DISPLACEMENT] - 1u, si, *df, dpdu, dpdv);
        } else {
            // bump_map, material.rs:1477-1508, with FloatConstantTexture: u_displace == v_displace == displace.
            // du, dv are finite and > 0 (interaction.rs:316-339 clamps; 0 -> 0.0005), so (x - x) / du == +0.
            Float displace = m.displacement;
            Float du = 0.0005f, dv = 0.0005f;
            dpdu = si.shading.dpdu + (displace - displace) / du * si.shading.n + displace * si.shading.dndu;
            dpdv = si.shading.dpdv + (displace - displace) / dv * si.shading.n + displace * si.shading.dndv;
        }
        V3 ns = normalize(cross(dpdu, dpdv));
        set_shading_geometry(si, ns, dpdu, dpdv, si.shading.dndu, si.shading.dndv, false);
    } else if (HAS_TEX && m.normal_map != 0u) {
        V3 dpdu, dpdv;
        normal_map_texture(sv, m.normal_map - 1u, si, dpdu, dpdv);
        V3 ns = normalize(cross(dpdu, dpdv));
        set_shading_geometry(si, ns, dpdu, dpdv, si.shading.dndu, si.shading.dndv, false);
    }
    auto tex = [&](const ShmSpectrum& s) { return spectrum_texture_evaluate<HAS_TEX>(sv, s, &tctx, lambda); };
    BxDF b;
    b.kind = m.kind;
    b.r = spec_const(0.0f);
    b.k = spec_const(0.0f);
    b.eta = 1.0f;
    b.mf.alpha_x = 0.0f;
    b.mf.alpha_y = 0.0f;
    b.mf2.alpha_x = 0.0f;
    b.mf2.alpha_y = 0.0f;
    b.thickness = 0.0f;
    b.g = 0.0f;
    b.albedo = spec_const(0.0f);
    b.max_depth = 0;
    b.n_samples = 1;
    b.strict = (int)sv.quirks_off;
    if (m.kind == SHM_MATERIAL_DIFFUSE) {
        b.r = clamp(tex(m.a), 0.0f, 1.0f);  // material.rs:301-311
    } else if (m.kind == SHM_MATERIAL_CONDUCTOR) {  // material.rs:456-499
        Float ur = fval(m, SHM_FLOATSLOT_U_ROUGHNESS, m.u_roughness), vr = fval(m, SHM_FLOATSLOT_V_ROUGHNESS, m.v_roughness);
        if (m.remap_roughness) { ur = roughness_to_alpha(ur); vr = roughness_to_alpha(vr); }
        b.r = tex(m.a);
        b.k = tex(m.b);
        b.mf = trowbridge_reitz_new(ur, vr);
    } else if (m.kind == SHM_MATERIAL_DIELECTRIC) {  // material.rs:603-635
        Float sampled_eta = spectrum_get(m.a, sv.spectrum_data, lambda.lambda[0]);
        if (m.a.kind != SHM_SPECTRUM_CONSTANT) terminate_secondary(lambda);
        if (sampled_eta == 0.0f) sampled_eta = 1.0f;
        Float ur = fval(m, SHM_FLOATSLOT_U_ROUGHNESS, m.u_roughness), vr = fval(m, SHM_FLOATSLOT_V_ROUGHNESS, m.v_roughness);
        if (m.remap_roughness) { ur = roughness_to_alpha(ur); vr = roughness_to_alpha(vr); }
        b.eta = sampled_eta;
        b.mf = trowbridge_reitz_new(ur, vr);
    } else if (m.kind == SHM_MATERIAL_THIN_DIELECTRIC) {  // material.rs:723-742
        Float sampled_eta = spectrum_get(m.a, sv.spectrum_data, lambda.lambda[0]);
        if (m.a.kind != SHM_SPECTRUM_CONSTANT) terminate_secondary(lambda);
        if (sampled_eta == 0.0f) sampled_eta = 1.0f;
        b.eta = sampled_eta;
    } else if (m.kind == SHM_MATERIAL_COATED_DIFFUSE) {  // material.rs:917-963
        b.r = clamp(tex(m.a), 0.0f, 1.0f);
        Float ur = fval(m, SHM_FLOATSLOT_U_ROUGHNESS, m.u_roughness), vr = fval(m, SHM_FLOATSLOT_V_ROUGHNESS, m.v_roughness);
        if (m.remap_roughness) { ur = roughness_to_alpha(ur); vr = roughness_to_alpha(vr); }
        b.mf = trowbridge_reitz_new(ur, vr);
        b.thickness = fval(m, SHM_FLOATSLOT_THICKNESS, m.thickness);
        Float sampled_eta = spectrum_get(m.d, sv.spectrum_data, lambda.lambda[0]);
        if (m.d.kind != SHM_SPECTRUM_CONSTANT) terminate_secondary(lambda);
        if (sampled_eta == 0.0f) sampled_eta = 1.0f;
        b.eta = sampled_eta;
        b.albedo = clamp(tex(m.c), 0.0f, 1.0f);
        b.g = clamp(fval(m, SHM_FLOATSLOT_G, m.g), -1.0f, 1.0f);
        b.max_depth = m.max_depth;
        b.n_samples = m.n_samples;
    } else {  // CoatedConductor, material.rs:1189-1256
        Float iur = fval(m, SHM_FLOATSLOT_U_ROUGHNESS, m.u_roughness), ivr = fval(m, SHM_FLOATSLOT_V_ROUGHNESS, m.v_roughness);
        if (m.remap_roughness) { iur = roughness_to_alpha(iur); ivr = roughness_to_alpha(ivr); }
        b.mf = trowbridge_reitz_new(iur, ivr);
        b.thickness = fval(m, SHM_FLOATSLOT_THICKNESS, m.thickness);
        Float ieta = spectrum_get(m.d, sv.spectrum_data, lambda.lambda[0]);
        if (m.d.kind != SHM_SPECTRUM_CONSTANT) terminate_secondary(lambda);
        if (ieta == 0.0f) ieta = 1.0f;
        Spec ce, ck;
        if (!m.conductor_from_reflectance) {
            ce = tex(m.a);
            ck = tex(m.b);
        } else {
            Spec r = clamp(tex(m.a), 0.0f, 0.9999f);  // "avoid r == 1 NaN case"
            ce = spec_const(1.0f);
            ck = 2.0f * spec_sqrt(r) / spec_sqrt(clamp_zero(spec_const(1.0f) - r));
        }
        ce = ce / ieta;
        ck = ck / ieta;
        Float cur = fval(m, SHM_FLOATSLOT_U2_ROUGHNESS, m.u2_roughness), cvr = fval(m, SHM_FLOATSLOT_V2_ROUGHNESS, m.v2_roughness);
        // material.rs:1239-1243: with remap_roughness the conductor's alphas are derived from the (already remapped)
        // INTERFACE roughness, not from its own: reference behaviour preserved
        if (m.remap_roughness) { cur = roughness_to_alpha(iur); cvr = roughness_to_alpha(ivr); }
        b.mf2 = trowbridge_reitz_new(cur, cvr);
        b.r = ce;
        b.k = ck;
        b.eta = ieta;
        b.albedo = clamp(tex(m.c), 0.0f, 1.0f);
        b.g = clamp(fval(m, SHM_FLOATSLOT_G, m.g), -1.0f, 1.0f);
        b.max_depth = m.max_depth;
        b.n_samples = m.n_samples;
    }
    return bsdf_new(si.shading.n, si.shading.dpdu, b);
}

// {Perspective,Orthographic}Camera::generate_ray_differential (camera.rs:1003-1079, 769-792) for one CameraSample; `aux`
// (may be null) receives the auxiliary rays.
SHM_HD Ray camera_generate_ray_differential(const ShmCamera& cam, V2 p_film, V2 p_lens, AuxRays* aux) {
    V3 p_camera = xf_point(cam.camera_from_raster, v3(p_film.x, p_film.y, 0.0f));
    Ray base;
    if (cam.kind == SHM_CAMERA_ORTHOGRAPHIC) {
        // camera.rs:769-792 returns this ray in CAMERA space: unlike generate_ray (:747-767) it never applies render_from_camera.
        // The integrators call generate_ray_differential (integrator.rs:351), so that is what is rendered; kept as written (with
        // CameraWorld render space the two spaces differ by the camera's rotation only). No depth of field there either.
        base.o = p_camera;
        base.d = v3(0.0f, 0.0f, 1.0f);
        if (aux) {
            aux->has = true;
            aux->rx_o = base.o + ld3(cam.dx_camera);
            aux->ry_o = base.o + ld3(cam.dy_camera);
            aux->rx_d = base.d;
            aux->ry_d = base.d;
        }
        return base;
    }
    base.o = v3s(0.0f);
    base.d = normalize(p_camera);
    if (cam.lens_radius > 0.0f) {
        V2 pl = cam.lens_radius * sample_uniform_disk_concentric(p_lens);
        Float ft = cam.focal_distance / base.d.z;
        V3 p_focus = base.o + base.d * ft;
        base.o = v3(pl.x, pl.y, 0.0f);
        base.d = normalize(p_focus - base.o);
    }
    if (aux) {
        AuxRays a;
        a.has = true;
        if (cam.lens_radius > 0.0f) {  // camera.rs:1036-1056
            V2 pl = cam.lens_radius * sample_uniform_disk_concentric(p_lens);
            V3 dx = normalize(p_camera + ld3(cam.dx_camera));
            Float ft = cam.focal_distance / dx.z;
            V3 p_focus = v3s(0.0f) + ft * dx;
            a.rx_o = v3(pl.x, pl.y, 0.0f);
            a.rx_d = normalize(p_focus - a.rx_o);
            V3 dy = normalize(p_camera + ld3(cam.dy_camera));
            ft = cam.focal_distance / dy.z;
            p_focus = v3s(0.0f) + ft * dy;
            a.ry_o = v3(pl.x, pl.y, 0.0f);
            a.ry_d = normalize(p_focus - a.ry_o);
        } else {  // camera.rs:1057-1068
            a.rx_o = base.o;
            a.ry_o = base.o;
            a.rx_d = normalize(p_camera + ld3(cam.dx_camera));
            a.ry_d = normalize(p_camera + ld3(cam.dy_camera));
        }
        // Transform::apply_ray for a RayDifferential, transform.rs:534-555: plain point / vector applies for the auxiliary rays
        aux->has = true;
        aux->rx_o = xf_point(cam.render_from_camera, a.rx_o);
        aux->rx_d = xf_vector(cam.render_from_camera, a.rx_d);
        aux->ry_o = xf_point(cam.render_from_camera, a.ry_o);
        aux->ry_d = xf_vector(cam.render_from_camera, a.ry_d);
    }
    return xf_ray(cam.render_from_camera, base);
}

// evaluate_pixel_sample head (integrator.rs:338-362): wavelength sample, camera sample, camera ray (+ its scaled differentials).
// The sampler dimension order (1d lambda | 2d filter | 2d lens | 1d time) is the reference's.
SHM_HD Ray generate_camera_ray(const SceneView& sv, int px, int py, Rng& rng, bool disable_wavelength_jitter,
                               bool disable_pixel_jitter, Wavelengths& lambda, Float& filter_weight, AuxRays* aux = nullptr,
                               int samples_per_pixel = 1) {
    Float lu = disable_wavelength_jitter ? 0.5f : sampler_get_1d(rng);
    lambda = sample_visible(lu);
    // get_camera_sample, sampling.rs:347-371 (filter.sample(get_pixel_2d()) is drawn in both branches)
    V2 uf = sampler_get_2d(rng);
    V2 fp = v2(lerp(uf.x, -sv.filter_radius[0], sv.filter_radius[0]), lerp(uf.y, -sv.filter_radius[1], sv.filter_radius[1]));
    V2 p_film, p_lens;
    if (disable_pixel_jitter) {
        p_film = v2((Float)px, (Float)py) + v2(0.5f, 0.5f);
        p_lens = v2(0.5f, 0.5f);
    } else {
        p_film = v2((Float)px, (Float)py) + fp + v2(0.5f, 0.5f);
        p_lens = sampler_get_2d(rng);
        (void)sampler_get_1d(rng);  // time
    }
    filter_weight = 1.0f;
    Ray r = camera_generate_ray_differential(sv.camera, p_film, p_lens, aux);
    if (aux) {  // integrator.rs:356-362
        Float ray_diff_scale = max(0.125f, 1.0f / sqrt((Float)samples_per_pixel));
        if (!disable_pixel_jitter) scale_differentials(r, *aux, ray_diff_scale);
    }
    return r;
}

// PixelSensor::to_sensor_rgb + the clamp of RgbFilm::add_sample (film.rs:907-914, 556-565).
SHM_HD V3 film_sample_rgb(const SceneView& sv, const Spec& l_in, const Wavelengths& lambda) {
    Spec l = safe_div(l_in, pdf_spec(lambda));
    V3 rgb = v3(average(dense_table_sample(sv.sensor_r_bar, lambda) * l),
                average(dense_table_sample(sv.sensor_g_bar, lambda) * l),
                average(dense_table_sample(sv.sensor_b_bar, lambda) * l))
             * sv.imaging_ratio;
    Float m = max(max(rgb.x, rgb.y), rgb.z);
    if (m > sv.max_component_value) rgb = rgb * sv.max_component_value / m;
    return rgb;
}

}  // namespace shm
