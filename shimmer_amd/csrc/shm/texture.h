// shm/texture.h — image textures on the path: ray differentials at a hit, texture coordinate mappings, MIP-pyramid filtering,
// RGB -> sigmoid-polynomial lookup, and the spectrum an image texture evaluates to.  
//
// Restates (paths relative to /root/reference/src):
//   interaction.rs:280-366        SurfaceInteraction::compute_differentials
//   camera.rs:307-354             CameraBase::approximate_dp_dxy
//   interaction.rs:430-514        SurfaceInteraction::spawn_ray_with_differentials (Igehy's specular differentials)
//   ray.rs:137-145                RayDifferential::scale_differentials
//   texture.rs:896-1046           UVMapping / SphericalMapping / CylindricalMapping / PlanarMapping ::map
//   texture.rs:777-808            SpectrumImageTexture::evaluate
//   mipmap.rs:121-199             MIPMap::filter (point / bilinear / trilinear / EWA level selection)
//   mipmap.rs:201-231, 301-331    texel_rgb, TexelType for RGB
//   mipmap.rs:233-295             TexelType::ewa
//   image.rs:134-180, 452-476, 619-646   remap_pixel_coords, get_channel_wrapped, bilerp_channel_wrapped
//   math.rs:439-452               modulo
//   spectra/spectrum.rs:498-607   Rgb{Albedo,Unbounded,Illuminant}Spectrum::new
//   colorspace.rs:95-98, rgb_to_spectra.rs:16-25    RgbColorSpace::to_rgb_coeffs -> rgb2spec::RGB2Spec::fetch
//
// Third-party arithmetic on this path, source not vendored (parity unpinned at the boundary, restated from the published
// algorithm): crate rgb2spec 0.1.1 `RGB2Spec::fetch` = Jakob & Hanika's rgb2spec_fetch (clamp to [0,1], largest component picks
// the table, scale-table interval search, trilinear interpolation of the three coefficients); black (largest component 0) is
// outside that algorithm's domain and is DEFINED here as the zero spectrum, see rgb2spec_fetch.  `f32::log2` is the platform's
// in the reference; here it is shm::log2 (fp.h), exact for powers of two.
//
// Reference behaviour kept as written: SphericalMapping::map returns spherical_theta for BOTH coordinates (texture.rs:966-969);
// CylindricalMapping::map's s is PI + atan2(y, x) / 2pi (texture.rs:996).
#pragma once
#include "scene.h"

namespace shm {

// ray.rs AuxiliaryRays + the Option around it
struct AuxRays {
    bool has;
    V3 rx_o, rx_d, ry_o, ry_d;
};
SHM_HD AuxRays aux_none() {
    AuxRays a;
    a.has = false;
    a.rx_o = a.rx_d = a.ry_o = a.ry_d = v3s(0.0f);
    return a;
}
// ray.rs:137-145
SHM_HD void scale_differentials(const Ray& ray, AuxRays& aux, Float s) {
    if (!aux.has) return;
    aux.rx_o = ray.o + (aux.rx_o - ray.o) * s;
    aux.ry_o = ray.o + (aux.ry_o - ray.o) * s;
    aux.rx_d = ray.d + (aux.rx_d - ray.d) * s;
    aux.ry_d = ray.d + (aux.ry_d - ray.d) * s;
}

// The differential fields of SurfaceInteraction (interaction.rs:37-44), kept beside it.
struct Differentials {
    V3 dpdx, dpdy;
    Float dudx, dvdx, dudy, dvdy;
};
SHM_HD Differentials differentials_zero() {
    Differentials d;
    d.dpdx = d.dpdy = v3s(0.0f);
    d.dudx = d.dvdx = d.dudy = d.dvdy = 0.0f;
    return d;
}

// Rot3 applied as Transform::apply_inverse(Vector) does with m_inv = transpose (transform.rs:249-252, 617-621)
SHM_HD V3 rot3_apply_transpose(const Rot3& r, V3 a) {
    return v3(r.m[0][0] * a.x + r.m[1][0] * a.y + r.m[2][0] * a.z, r.m[0][1] * a.x + r.m[1][1] * a.y + r.m[2][1] * a.z,
              r.m[0][2] * a.x + r.m[1][2] * a.y + r.m[2][2] * a.z);
}
// Transform::apply(Point3f) of a pure rotation: apply_point_helper with m[i][3] = 0, m[3] = (0,0,0,1) (transform.rs:753-767)
SHM_HD V3 rot3_apply_point(const Rot3& r, V3 p) {
    Float xp = r.m[0][0] * p.x + r.m[0][1] * p.y + r.m[0][2] * p.z + 0.0f;
    Float yp = r.m[1][0] * p.x + r.m[1][1] * p.y + r.m[1][2] * p.z + 0.0f;
    Float zp = r.m[2][0] * p.x + r.m[2][1] * p.y + r.m[2][2] * p.z + 0.0f;
    return v3(xp, yp, zp);  // wp = 0*x + 0*y + 0*z + 1 == 1 for finite p: no division
}

// camera.rs:307-354
SHM_HD void approximate_dp_dxy(const ShmCamera& cam, V3 p, V3 n, int samples_per_pixel, bool disable_pixel_jitter, V3& dpdx, V3& dpdy) {
    V3 p_camera = xf_point(cam.camera_from_render, p);                      // render_from_camera.apply_inverse(p)
    Rot3 down_z_from_camera = rotate_from_to(normalize(p_camera), v3(0.0f, 0.0f, 1.0f));
    V3 p_down_z = rot3_apply_point(down_z_from_camera, p_camera);
    // camera_from_render_n = apply_normal_helper(&render_from_camera.m, n) (transform.rs:623-629); then apply(Normal) of the
    // rotation = apply_normal_helper(&m_inv = transpose) = the plain matrix-vector product
    V3 n_camera = xf_normal(cam.render_from_camera, n);
    V3 n_down_z = rot3_apply(down_z_from_camera, n_camera);
    Float d = n_down_z.z * p_down_z.z;
    V3 xo = v3s(0.0f) + ld3(cam.min_pos_differential_x), xd = v3(0.0f, 0.0f, 1.0f) + ld3(cam.min_dir_differential_x);
    Float tx = -(dot(n_down_z, xo) - d) / dot(n_down_z, xd);
    V3 yo = v3s(0.0f) + ld3(cam.min_pos_differential_y), yd = v3(0.0f, 0.0f, 1.0f) + ld3(cam.min_dir_differential_y);
    Float ty = -(dot(n_down_z, yo) - d) / dot(n_down_z, yd);
    V3 px = xo + xd * tx;  // Ray::get, ray.rs:29-31
    V3 py = yo + yd * ty;
    Float spp_scale = disable_pixel_jitter ? 1.0f : max(0.125f, 1.0f / sqrt((Float)samples_per_pixel));
    dpdx = spp_scale * xf_vector(cam.render_from_camera, rot3_apply_transpose(down_z_from_camera, px - p_down_z));
    dpdy = spp_scale * xf_vector(cam.render_from_camera, rot3_apply_transpose(down_z_from_camera, py - p_down_z));
}

// interaction.rs:280-366
SHM_HD Differentials compute_differentials(const SceneView& sv, const SurfaceInteraction& si, const AuxRays& aux, int samples_per_pixel,
                                           bool disable_pixel_jitter, bool disable_texture_filtering = false) {
    if (disable_texture_filtering) return differentials_zero();  // :287-295
    Differentials r;
    V3 p = si.p();
    if (aux.has && dot(si.n, aux.rx_d) != 0.0f && dot(si.n, aux.ry_d) != 0.0f) {
        Float d = -dot(si.n, p);
        Float tx = (-dot(si.n, aux.rx_o) - d) / dot(si.n, aux.rx_d);
        V3 px = aux.rx_o + tx * aux.rx_d;
        Float ty = (-dot(si.n, aux.ry_o) - d) / dot(si.n, aux.ry_d);
        V3 py = aux.ry_o + ty * aux.ry_d;
        r.dpdx = px - p;
        r.dpdy = py - p;
    } else {
        approximate_dp_dxy(sv.camera, p, si.n, samples_per_pixel, disable_pixel_jitter, r.dpdx, r.dpdy);
    }
    Float ata00 = dot(si.dpdu, si.dpdu), ata01 = dot(si.dpdu, si.dpdv), ata11 = dot(si.dpdv, si.dpdv);
    Float inv_det = 1.0f / difference_of_products(ata00, ata11, ata01, ata01);
    if (!is_finite(inv_det)) inv_det = 0.0f;
    Float atb0x = dot(si.dpdu, r.dpdx), atb1x = dot(si.dpdv, r.dpdx);
    Float atb0y = dot(si.dpdu, r.dpdy), atb1y = dot(si.dpdv, r.dpdy);
    r.dudx = difference_of_products(ata11, atb0x, ata01, atb1x) * inv_det;
    r.dvdx = difference_of_products(ata00, atb1x, ata01, atb0x) * inv_det;
    r.dudy = difference_of_products(ata11, atb0y, ata01, atb1y) * inv_det;
    r.dvdy = difference_of_products(ata00, atb1y, ata01, atb0y) * inv_det;
    r.dudx = is_finite(r.dudx) ? clamp(r.dudx, -1e8f, 1e8f) : 0.0f;
    r.dvdx = is_finite(r.dvdx) ? clamp(r.dvdx, -1e8f, 1e8f) : 0.0f;
    r.dudy = is_finite(r.dudy) ? clamp(r.dudy, -1e8f, 1e8f) : 0.0f;
    r.dvdy = is_finite(r.dvdy) ? clamp(r.dvdy, -1e8f, 1e8f) : 0.0f;
    return r;
}

// interaction.rs:430-514: the auxiliary rays of the ray spawned towards wi (the main ray is interaction.spawn_ray(wi) as before).
// _pre takes the surface part (p, wo, shading normal, dp/dx, dp/dy and the normal derivatives dn/dx = dndu du/dx + dndv dv/dx,
// interaction.rs:436-440) as values: the staged shading computes it where the interaction lives and uses it one kernel later.
SHM_HD AuxRays spawn_ray_differentials_pre(V3 p, V3 wo, V3 n, V3 dpdx, V3 dpdy, V3 dndx, V3 dndy, const AuxRays& aux_i, V3 wi, uint32_t flags, Float eta) {
    AuxRays rd = aux_none();
    if (aux_i.has) {
        V3 dwodx = -aux_i.rx_d - wo;
        V3 dwody = -aux_i.ry_d - wo;
        if (flags == BXDF_SPECULAR_REFLECTION) {
            rd.has = true;
            rd.rx_o = p + dpdx;
            rd.ry_o = p + dpdy;
            Float dwo_dotn_dx = dot(dwodx, n) + dot(wo, dndx);
            Float dwo_dotn_dy = dot(dwody, n) + dot(wo, dndy);
            rd.rx_d = wi - dwodx + 2.0f * (dot(wo, n) * dndx + dwo_dotn_dx * n);
            rd.ry_d = wi - dwody + 2.0f * (dot(wo, n) * dndy + dwo_dotn_dy * n);
        } else if (flags == BXDF_SPECULAR_TRANSMISSION) {
            rd.has = true;
            rd.rx_o = p + dpdx;
            rd.ry_o = p + dpdy;
            if (dot(wo, n) < 0.0f) {
                n = -n;
                dndx = -dndx;
                dndy = -dndy;
            }
            Float dwo_dotn_dx = dot(dwodx, n) + dot(wo, dndx);
            Float dwo_dotn_dy = dot(dwody, n) + dot(wo, dndy);
            Float mu = dot(wo, n) / eta - abs_dot(wi, n);
            Float dmudx = dwo_dotn_dx * (1.0f / eta + 1.0f / sqr(eta) * dot(wo, n) / dot(wi, n));
            Float dmudy = dwo_dotn_dy * (1.0f / eta + 1.0f / sqr(eta) * dot(wo, n) / dot(wi, n));
            rd.rx_d = wi - eta * dwodx + (mu * dndx + dmudx * n);
            rd.ry_d = wi - eta * dwody + (mu * dndy + dmudy * n);
        }
    }
    if (rd.has && (length_squared(rd.rx_d) > 1e16f || length_squared(rd.ry_d) > 1e16f || length_squared(rd.rx_o) > 1e16f ||
                   length_squared(rd.ry_o) > 1e16f))
        rd.has = false;
    return rd;
}
SHM_HD AuxRays spawn_ray_differentials(const SurfaceInteraction& si, const Differentials& df, const AuxRays& aux_i, V3 wi, uint32_t flags, Float eta) {
    V3 dndx = si.shading.dndu * df.dudx + si.shading.dndv * df.dvdx;
    V3 dndy = si.shading.dndu * df.dudy + si.shading.dndv * df.dvdy;
    return spawn_ray_differentials_pre(si.p(), si.wo, si.shading.n, df.dpdx, df.dpdy, dndx, dndy, aux_i, wi, flags, eta);
}

// ---------------------------------------------------------------------------------------------
// texture coordinates
// ---------------------------------------------------------------------------------------------
struct TextureEvalContext {  // texture.rs:1057-1067
    V3 p, dpdx, dpdy, n;
    V2 uv;
    Float dudx, dudy, dvdx, dvdy;
};
SHM_HD TextureEvalContext tex_ctx_from(const SurfaceInteraction& si, const Differentials& df) {  // texture.rs:1111-1125
    TextureEvalContext c;
    c.p = si.p(); c.dpdx = df.dpdx; c.dpdy = df.dpdy; c.n = si.n; c.uv = si.uv;
    c.dudx = df.dudx; c.dudy = df.dudy; c.dvdx = df.dvdx; c.dvdy = df.dvdy;
    return c;
}
// ---- the calling form of the four texture evaluators that are REAL calls on the device (SHM_HD_NOINLINE below) ----
// A `const SceneView&` / `const TextureEvalContext&` / `const Wavelengths&` handed to a real call has to live in memory: the kernel's SceneView (≈ 90
// registers' worth of pointers) became a per-lane stack object, every pointer of it — in the callees AND in the kernel's own inlined code, since an escaped
// object is not promoted to registers at all — was read back from scratch in front of the gather it addresses, and the context (18 floats) and the wavelengths
// (8) were stored and re-read around every call (round 4: 1 240 B of scratch per lane in k_vertex<textured> with no register spilled). The `_v` forms take
// the scene as a POINTER TO THE WORKGROUP'S COPY IN LDS (SceneView::call_copy, set by stage_scene_tables_tex; the host passes the object's own address) and
// the context and the wavelengths as scalars (clang passes scalars in registers however many there are; an aggregate beyond the 16th argument register
// goes through the stack). Same statements, same values.
#if defined(__HIP_DEVICE_COMPILE__)
#define SHM_SV_FOR_CALL(sv) ((sv).call_copy)
#else
#define SHM_SV_FOR_CALL(sv) (&(sv))
#endif
#define SHM_TEXCTX_PARAMS Float c_px, Float c_py, Float c_pz, Float c_dpdx_x, Float c_dpdx_y, Float c_dpdx_z, Float c_dpdy_x, Float c_dpdy_y, Float c_dpdy_z, \
                          Float c_nx, Float c_ny, Float c_nz, Float c_u, Float c_v, Float c_dudx, Float c_dudy, Float c_dvdx, Float c_dvdy
#define SHM_TEXCTX_ARGS(c) (c).p.x, (c).p.y, (c).p.z, (c).dpdx.x, (c).dpdx.y, (c).dpdx.z, (c).dpdy.x, (c).dpdy.y, (c).dpdy.z, \
                           (c).n.x, (c).n.y, (c).n.z, (c).uv.x, (c).uv.y, (c).dudx, (c).dudy, (c).dvdx, (c).dvdy
#define SHM_TEXCTX_FROM_PARAMS(name)                                                                                                         \
    TextureEvalContext name;                                                                                                                 \
    name.p = v3(c_px, c_py, c_pz); name.dpdx = v3(c_dpdx_x, c_dpdx_y, c_dpdx_z); name.dpdy = v3(c_dpdy_x, c_dpdy_y, c_dpdy_z);               \
    name.n = v3(c_nx, c_ny, c_nz); name.uv = v2(c_u, c_v); name.dudx = c_dudx; name.dudy = c_dudy; name.dvdx = c_dvdx; name.dvdy = c_dvdy
#define SHM_LAMBDA_PARAMS Float l_0, Float l_1, Float l_2, Float l_3, Float l_p0, Float l_p1, Float l_p2, Float l_p3
#define SHM_LAMBDA_ARGS(l) (l).lambda[0], (l).lambda[1], (l).lambda[2], (l).lambda[3], (l).pdf[0], (l).pdf[1], (l).pdf[2], (l).pdf[3]
#define SHM_LAMBDA_FROM_PARAMS(name)                                                                     \
    Wavelengths name;                                                                                    \
    name.lambda[0] = l_0; name.lambda[1] = l_1; name.lambda[2] = l_2; name.lambda[3] = l_3;              \
    name.pdf[0] = l_p0; name.pdf[1] = l_p1; name.pdf[2] = l_p2; name.pdf[3] = l_p3
static_assert(NSPEC == 4, "the scalar calling form of the texture evaluators spells out four wavelengths");

struct TexCoord2D {  // texture.rs:1048-1054
    V2 st;
    Float dsdx, dsdy, dtdx, dtdy;
};
SHM_HD Float spherical_theta(V3 v) { return safe_acos(v.z); }  // vecmath/spherical.rs:16-18

SHM_HD TexCoord2D texture_map(const ShmImageTexture& t, const TextureEvalContext& ctx, bool strict = false) {
    TexCoord2D c;
    if (t.mapping == SHM_TEXMAP_UV) {  // texture.rs:918-935
        c.dsdx = t.su * ctx.dudx;
        c.dsdy = t.su * ctx.dudy;
        c.dtdx = t.sv * ctx.dvdx;
        c.dtdy = t.sv * ctx.dvdy;
        c.st = v2(t.su * ctx.uv.x + t.du, t.sv * ctx.uv.y + t.dv);
        return c;
    }
    const Float* m = t.texture_from_render;
    if (t.mapping == SHM_TEXMAP_SPHERICAL) {  // texture.rs:943-975
        V3 pt = xf_point(m, ctx.p);
        Float x2y2 = sqr(pt.x) + sqr(pt.y);
        Float sqrtx2y2 = sqrt(x2y2);
        V3 dsdp = v3(-pt.y, pt.x, 0.0f) / (2.0f * PI_F * x2y2);
        V3 dtdp = 1.0f / (PI_F * (x2y2 + sqr(pt.z))) * v3(pt.x * pt.z / sqrtx2y2, pt.y * pt.z / sqrtx2y2, -sqrtx2y2);
        V3 dpdx = xf_vector(m, ctx.dpdx), dpdy = xf_vector(m, ctx.dpdy);
        c.dsdx = dot(dsdp, dpdx);
        c.dsdy = dot(dsdp, dpdy);
        c.dtdx = dot(dtdp, dpdx);
        c.dtdy = dot(dtdp, dpdy);
        V3 vec = normalize(pt - v3s(0.0f));
        if (strict) {  // PBRT-v4's SphericalMapping: (theta / pi, phi / 2 pi) with a real acos
            Float phi = atan2(vec.y, vec.x);
            c.st = v2(acos(clamp(vec.z, -1.0f, 1.0f)) * INV_PI, (phi < 0.0f ? phi + 2.0f * PI_F : phi) * INV_2PI);
        } else c.st = v2(spherical_theta(vec) * INV_PI, spherical_theta(vec) * INV_2PI);
        return c;
    }
    if (t.mapping == SHM_TEXMAP_CYLINDRICAL) {  // texture.rs:983-1009
        V3 pt = xf_point(m, ctx.p);
        Float x2y2 = sqr(pt.x) + sqr(pt.y);
        V3 dsdp = v3(-pt.y, pt.x, 0.0f) / (2.0f * PI_F * x2y2);
        V3 dtdp = v3(0.0f, 0.0f, 1.0f);
        V3 dpdx = xf_vector(m, ctx.dpdx), dpdy = xf_vector(m, ctx.dpdy);
        c.dsdx = dot(dsdp, dpdx);
        c.dsdy = dot(dsdp, dpdy);
        c.dtdx = dot(dtdp, dpdx);
        c.dtdy = dot(dtdp, dpdy);
        c.st = v2(PI_F + atan2(pt.y, pt.x) * INV_2PI, pt.z);
        return c;
    }
    // PlanarMapping, texture.rs:1021-1044
    V3 vec = xf_point(m, ctx.p);
    V3 dpdx = xf_vector(m, ctx.dpdx), dpdy = xf_vector(m, ctx.dpdy);
    V3 vs = ld3(t.vs), vt = ld3(t.vt);
    c.dsdx = dot(vs, dpdx);
    c.dsdy = dot(vs, dpdy);
    c.dtdx = dot(vt, dpdx);
    c.dtdy = dot(vt, dpdy);
    c.st = v2(t.du + dot(vec, vs), t.dv + dot(vec, vt));
    return c;
}

// ---------------------------------------------------------------------------------------------
// image access
// ---------------------------------------------------------------------------------------------
SHM_HD int modulo_i(int a, int b) {  // math.rs:439-452
    int result = a - (a / b) * b;
    return result < 0 ? result + b : result;
}
// image.rs:134-180; returns false for an out-of-bounds coordinate under WrapMode::Black
SHM_HD bool remap_pixel_coords(int& x, int& y, int rx, int ry, uint32_t wrap) {
    if (wrap == SHM_WRAP_OCTAHEDRAL_SPHERE) {
        if (x < 0) { x = -x; y = ry - 1 - y; }
        else if (x >= rx) { x = 2 * rx - 1 - x; y = ry - 1 - y; }
        if (y < 0) { x = rx - 1 - x; y = -y; }
        else if (y >= ry) { x = rx - 1 - x; y = 2 * ry - 1 - y; }
        if (rx == 1) x = 0;
        if (ry == 1) y = 0;
        // One mirror per axis is all the reference does: a coordinate further out than one image size stays out of range and
        // the reference panics on the slice index (image.rs:441-443). No panic exists on the device: clamp (memory safety only).
        x = x < 0 ? 0 : (x > rx - 1 ? rx - 1 : x);
        y = y < 0 ? 0 : (y > ry - 1 ? ry - 1 : y);
        return true;
    }
    if (!(x >= 0 && x < rx)) {
        if (wrap == SHM_WRAP_BLACK) return false;
        if (wrap == SHM_WRAP_CLAMP) x = x < 0 ? 0 : (x > rx - 1 ? rx - 1 : x);
        else x = modulo_i(x, rx);
    }
    if (!(y >= 0 && y < ry)) {
        if (wrap == SHM_WRAP_BLACK) return false;
        if (wrap == SHM_WRAP_CLAMP) y = y < 0 ? 0 : (y > ry - 1 ? ry - 1 : y);
        else y = modulo_i(y, ry);
    }
    return true;
}
// Rust `f as i32`: saturating, NaN -> 0
SHM_HD int float_to_i32(Float f) {
    if (is_nan(f)) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return -2147483647 - 1;
    return (int)f;
}

struct RGB3 {  // color.rs:192-196; every operator there is component-wise
    Float r, g, b;
};
SHM_HD RGB3 rgb3(Float r, Float g, Float b) { RGB3 c; c.r = r; c.g = g; c.b = b; return c; }
SHM_HD RGB3 operator+(RGB3 a, RGB3 b) { return rgb3(a.r + b.r, a.g + b.g, a.b + b.b); }
SHM_HD RGB3 operator-(RGB3 a, RGB3 b) { return rgb3(a.r - b.r, a.g - b.g, a.b - b.b); }
SHM_HD RGB3 operator*(RGB3 a, Float s) { return rgb3(a.r * s, a.g * s, a.b * s); }
SHM_HD RGB3 operator/(RGB3 a, Float s) { return rgb3(a.r / s, a.g / s, a.b / s); }
SHM_HD RGB3 lerp_rgb(Float t, RGB3 a, RGB3 b) { return a * (1.0f - t) + b * t; }  // math.rs:246-252

struct TextureView {
    const ShmImageTexture* t;
    const ShmImageLevel* levels;  // already offset to the texture's first level
    const Float* texels;
};
// Image::get_channel_wrapped, image.rs:452-476 (every level is handed over as f32)
SHM_HD Float tex_channel(const TextureView& tv, int level, int x, int y, int c) {
    const ShmImageLevel& l = tv.levels[level];
    if (!remap_pixel_coords(x, y, l.width, l.height, tv.t->wrap)) return 0.0f;
    return tv.texels[l.texel_offset + (uint32_t)((int)tv.t->n_channels * (y * l.width + x)) + (uint32_t)c];
}
// MIPMap::texel_rgb, mipmap.rs:201-219
SHM_HD RGB3 tex_texel(const TextureView& tv, int level, int x, int y) {
    if (tv.t->n_channels == 3) return rgb3(tex_channel(tv, level, x, y, 0), tex_channel(tv, level, x, y, 1), tex_channel(tv, level, x, y, 2));
    Float v = tex_channel(tv, level, x, y, 0);
    return rgb3(v, v, v);
}
// Image::bilerp_channel_wrapped, image.rs:619-646
SHM_HD Float tex_bilerp_channel(const TextureView& tv, int level, V2 p, int c) {
    const ShmImageLevel& l = tv.levels[level];
    Float x = p.x * (Float)l.width - 0.5f;
    Float y = p.y * (Float)l.height - 0.5f;
    int xi = float_to_i32(floor(x)), yi = float_to_i32(floor(y));
    Float dx = x - (Float)xi, dy = y - (Float)yi;
    Float v0 = tex_channel(tv, level, xi, yi, c), v1 = tex_channel(tv, level, xi + 1, yi, c);
    Float v2_ = tex_channel(tv, level, xi, yi + 1, c), v3_ = tex_channel(tv, level, xi + 1, yi + 1, c);
    return (1.0f - dx) * (1.0f - dy) * v0 + dx * (1.0f - dy) * v1 + (1.0f - dx) * dy * v2_ + dx * dy * v3_;
}
// TexelType for RGB :: bilerp, mipmap.rs:314-330
SHM_HD RGB3 tex_bilerp(const TextureView& tv, int level, V2 st) {
    if (tv.t->n_channels == 3) return rgb3(tex_bilerp_channel(tv, level, st, 0), tex_bilerp_channel(tv, level, st, 1), tex_bilerp_channel(tv, level, st, 2));
    Float v = tex_bilerp_channel(tv, level, st, 0);
    return rgb3(v, v, v);
}

constexpr int MIP_FILTER_LUT_SIZE = 128;  // mipmap.rs:390

// TexelType (mipmap.rs:233-331): what MIPMap::filter is generic over. RGB: texel_rgb / three bilerps (:201-219, :314-330).
// Float: texel = channel 0; bilerp = channel 0 of a one-channel image, the AVERAGE of the three bilerps of an RGB one (:297-312;
// an RGBA image would use its alpha: the ABI hands over 1 or 3 channels).
template <typename T> struct TexelOps;
template <> struct TexelOps<RGB3> {
    static SHM_HD RGB3 zero() { return rgb3(0.0f, 0.0f, 0.0f); }
    static SHM_HD RGB3 texel(const TextureView& tv, int level, int x, int y) { return tex_texel(tv, level, x, y); }
    // the same value from a texel's address (tex_texel / tex_channel with the coordinates already remapped; outside the image: black)
    static SHM_HD RGB3 from_channels(const Float* p, int n_channels, bool inside_image) {
        if (!inside_image) return rgb3(0.0f, 0.0f, 0.0f);
        if (n_channels == 3) return rgb3(p[0], p[1], p[2]);
        return rgb3(p[0], p[0], p[0]);
    }
    static SHM_HD RGB3 bilerp(const TextureView& tv, int level, V2 st) { return tex_bilerp(tv, level, st); }
    static SHM_HD RGB3 lerp(Float t, RGB3 a, RGB3 b) { return lerp_rgb(t, a, b); }
};
template <> struct TexelOps<Float> {
    static SHM_HD Float zero() { return 0.0f; }
    static SHM_HD Float texel(const TextureView& tv, int level, int x, int y) { return tex_channel(tv, level, x, y, 0); }
    static SHM_HD Float from_channels(const Float* p, int, bool inside_image) { return inside_image ? p[0] : 0.0f; }
    static SHM_HD Float bilerp(const TextureView& tv, int level, V2 st) {
        if (tv.t->n_channels == 1) return tex_bilerp_channel(tv, level, st, 0);
        // ImageChannelValues::average, image.rs:275-277
        return (((0.0f + tex_bilerp_channel(tv, level, st, 0)) + tex_bilerp_channel(tv, level, st, 1)) + tex_bilerp_channel(tv, level, st, 2)) / 3.0f;
    }
    static SHM_HD Float lerp(Float t, Float a, Float b) { return shm::lerp(t, a, b); }
};

// TexelType::ewa, mipmap.rs:233-295
template <typename T>
SHM_HD T tex_ewa(const TextureView& tv, const Float* lut, int level, V2 st, V2 dst0, V2 dst1) {
    int n_levels = (int)tv.t->n_levels;
    if (level >= n_levels) return TexelOps<T>::texel(tv, n_levels - 1, 0, 0);
    const ShmImageLevel& l = tv.levels[level];
    st.x = st.x * (Float)l.width - 0.5f;
    st.y = st.y * (Float)l.height - 0.5f;
    dst0.x *= (Float)l.width;
    dst0.y *= (Float)l.height;
    dst1.x *= (Float)l.width;
    dst1.y *= (Float)l.height;
    Float a = sqr(dst0.y) + sqr(dst1.y) + 1.0f;
    Float b = -2.0f * (dst0.x * dst0.y + dst1.x * dst1.y);
    Float c = sqr(dst0.x) + sqr(dst1.x) + 1.0f;
    Float inv_f = 1.0f / (a * c - sqr(b) * 0.25f);
    a *= inv_f;
    b *= inv_f;
    c *= inv_f;
    Float det = -sqr(b) + 4.0f * a * c;
    Float inv_det = 1.0f / det;
    Float u_sqrt = safe_sqrt(det * c), v_sqrt = safe_sqrt(a * det);
    int s0 = float_to_i32(ceil(st.x - 2.0f * inv_det * u_sqrt));
    int s1 = float_to_i32(floor(st.x + 2.0f * inv_det * u_sqrt));
    int t0 = float_to_i32(ceil(st.y - 2.0f * inv_det * v_sqrt));
    int t1 = float_to_i32(floor(st.y + 2.0f * inv_det * v_sqrt));
    T sum = TexelOps<T>::zero();
    Float sum_wts = 0.0f;
    // The level's and the texture's fields are read once, and a texel of the ellipse's bounding box is FETCHED whether or not it lies inside the
    // ellipse (its address is valid either way: remap_pixel_coords wraps, clamps or blacks it out): with the load inside the `r2 < 1` branch a
    // wave made one dependent memory round trip per texel. What is accumulated, and in which order, is unchanged.
    const int lw = l.width, lh = l.height, nc = (int)tv.t->n_channels;
    const uint32_t wrap = tv.t->wrap;
    const Float* const level_texels = tv.texels + l.texel_offset;
    for (int it = t0; it <= t1; ++it) {
        Float tt = (Float)it - st.y;
        for (int is = s0; is <= s1; ++is) {
            Float ss = (Float)is - st.x;
            Float r2 = a * sqr(ss) + b * ss * tt + c * sqr(tt);
            int x = is, y = it;
            const bool inside_image = remap_pixel_coords(x, y, lw, lh, wrap);
            const Float* tp = level_texels + (inside_image ? (uint32_t)(nc * (y * lw + x)) : 0u);
            const T texel = TexelOps<T>::from_channels(tp, nc, inside_image);
            int index = float_to_i32(r2 * (Float)MIP_FILTER_LUT_SIZE);  // `as usize` saturates at 0 for negatives
            if (index < 0) index = 0;
            if (index > MIP_FILTER_LUT_SIZE - 1) index = MIP_FILTER_LUT_SIZE - 1;
            const Float weight = lut[index];
            if (r2 < 1.0f) {
                sum = sum + texel * weight;
                sum_wts += weight;
            }
        }
    }
    return sum / sum_wts;
}

// MIPMap::filter::<T>, mipmap.rs:121-199
template <typename T>
SHM_HD T tex_filter(const TextureView& tv, const Float* lut, V2 st, V2 dst0, V2 dst1) {
    const ShmImageTexture& t = *tv.t;
    int n_levels = (int)t.n_levels;
    if (t.filter == SHM_TEXFILTER_EWA) {
        if (length_squared(dst0) < length_squared(dst1)) { V2 tmp = dst0; dst0 = dst1; dst1 = tmp; }
        Float longer_vec_length = sqrt(length_squared(dst0));
        Float shorter_vec_length = sqrt(length_squared(dst1));
        if (shorter_vec_length * t.max_anisotropy < longer_vec_length && shorter_vec_length > 0.0f) {
            Float scale = longer_vec_length / (shorter_vec_length * t.max_anisotropy);
            dst1 = dst1 * scale;
            shorter_vec_length *= scale;
        }
        if (shorter_vec_length == 0.0f) return TexelOps<T>::bilerp(tv, 0, st);
        Float lod = max(0.0f, (Float)n_levels - 1.0f + log2(shorter_vec_length));
        int ilod = float_to_i32(floor(lod));
        return TexelOps<T>::lerp(lod - (Float)ilod, tex_ewa<T>(tv, lut, ilod, st, dst0, dst1), tex_ewa<T>(tv, lut, ilod + 1, st, dst0, dst1));
    }
    // reduce(|acc, e| acc.max(e)) over [|dst0.x|, |dst0.y|, |dst1.x|, |dst1.y|]
    Float width = 2.0f * max(max(max(abs(dst0.x), abs(dst0.y)), abs(dst1.x)), abs(dst1.y));
    Float level = (Float)n_levels - 1.0f + log2(max(width, 1e-8f));
    if (level >= (Float)n_levels - 1.0f) return TexelOps<T>::texel(tv, n_levels - 1, 0, 0);
    int i_level = float_to_i32(floor(level));
    if (i_level < 0) i_level = 0;
    if (t.filter == SHM_TEXFILTER_POINT) {
        const ShmImageLevel& l = tv.levels[i_level];
        int sx = float_to_i32(round(st.x * (Float)l.width - 0.5f));
        int sy = float_to_i32(round(st.y * (Float)l.height - 0.5f));
        return TexelOps<T>::texel(tv, i_level, sx, sy);
    }
    if (t.filter == SHM_TEXFILTER_BILINEAR) return TexelOps<T>::bilerp(tv, i_level, st);
    // Trilinear
    if (i_level == 0) return TexelOps<T>::bilerp(tv, 0, st);
    return TexelOps<T>::lerp(level - (Float)i_level, TexelOps<T>::bilerp(tv, i_level, st), TexelOps<T>::bilerp(tv, i_level + 1, st));
}
SHM_HD TextureView texture_view(const SceneView& sv, uint32_t texture_index) {
    TextureView tv;
    tv.t = &sv.image_textures[texture_index];
    tv.levels = sv.image_levels + tv.t->first_level;
    tv.texels = sv.texel_data;
    return tv;
}
// FloatImageTexture::evaluate, texture.rs:393-403
SHM_HD_NOINLINE Float float_image_texture_evaluate_v(const SceneView* svp, uint32_t texture_index, SHM_TEXCTX_PARAMS) {
    const SceneView& sv = *svp;
    SHM_TEXCTX_FROM_PARAMS(ctx);
    TextureView tv = texture_view(sv, texture_index);
    TexCoord2D c = texture_map(*tv.t, ctx, sv.quirks_off != 0);
    c.st.y = 1.0f - c.st.y;
    Float v = tex_filter<Float>(tv, sv.ewa_lut, c.st, v2(c.dsdx, c.dtdx), v2(c.dsdy, c.dtdy)) * tv.t->scale;
    return tv.t->invert ? max(0.0f, 1.0f - v) : v;
}
SHM_HD Float float_image_texture_evaluate(const SceneView& sv, uint32_t texture_index, const TextureEvalContext& ctx) {
    return float_image_texture_evaluate_v(SHM_SV_FOR_CALL(sv), texture_index, SHM_TEXCTX_ARGS(ctx));
}
// FloatTexture::evaluate (texture.rs:142-152): the node's post-order program (scene.h FloatTexOp), children before parents
SHM_HD_NOINLINE Float float_texture_evaluate_v(const SceneView* svp, uint32_t index, SHM_TEXCTX_PARAMS) {
    const SceneView& sv = *svp;
    SHM_TEXCTX_FROM_PARAMS(ctx);
    const FloatTexRange r = sv.ftex_ranges[index];
    if (r.count == 1u) {
        // a program of one op is a leaf (the combining kinds have children before them): evaluated without the value array, which — indexed at run time —
        // lives in scratch memory on the device (a store and a dependent load around the one value)
        const ShmFloatTexture& t = sv.float_textures[sv.ftex_ops[r.first].node];
        if (t.kind == SHM_FLOATTEX_CONSTANT) return t.value;
        if (t.kind == SHM_FLOATTEX_IMAGE) return float_image_texture_evaluate(sv, t.image, ctx);
    }
    Float vals[FTEX_MAX_OPS];
    for (uint32_t k = 0; k < r.count; ++k) {
        const FloatTexOp op = sv.ftex_ops[r.first + k];
        const ShmFloatTexture& t = sv.float_textures[op.node];
        Float v;
        if (t.kind == SHM_FLOATTEX_CONSTANT) {            // texture.rs:175-179
            v = t.value;
        } else if (t.kind == SHM_FLOATTEX_IMAGE) {
            v = float_image_texture_evaluate(sv, t.image, ctx);
        } else if (t.kind == SHM_FLOATTEX_SCALED) {       // texture.rs:206-214: tex = a, scale = b
            Float sc = vals[op.b];
            v = (sc == 0.0f) ? 0.0f : vals[op.a] * sc;
        } else if (t.kind == SHM_FLOATTEX_MIX) {          // texture.rs:244-260: tex1 = a, tex2 = b, amount = c
            Float amt = vals[op.c];
            Float t1 = (amt != 1.0f) ? vals[op.a] : 0.0f;
            Float t2 = (amt != 0.0f) ? vals[op.b] : 0.0f;
            v = t1 * (1.0f - amt) + t2 * amt;
        } else {                                          // FloatDirectionMixTexture, texture.rs:290-305
            Float amt = dot(ctx.n, ld3(t.dir));
            Float t1 = (amt != 0.0f) ? vals[op.a] : 0.0f;
            Float t2 = (amt != 1.0f) ? vals[op.b] : 0.0f;
            v = amt * t1 + (1.0f - amt) * t2;
        }
        vals[k] = v;
    }
    return vals[r.count - 1];
}
SHM_HD Float float_texture_evaluate(const SceneView& sv, uint32_t index, const TextureEvalContext& ctx) {
    return float_texture_evaluate_v(SHM_SV_FOR_CALL(sv), index, SHM_TEXCTX_ARGS(ctx));
}

// ---------------------------------------------------------------------------------------------
// RGB -> sigmoid polynomial coefficients: rgb2spec 0.1.1 RGB2Spec::fetch (un-vendored; Jakob & Hanika's rgb2spec_fetch)
// ---------------------------------------------------------------------------------------------
SHM_HD void rgb2spec_fetch(const SceneView& sv, RGB3 rgb_in, Float out[3]) {
    const int res = (int)sv.rgb2spec_res;
    Float rgb[3] = {max(min(rgb_in.r, 1.0f), 0.0f), max(min(rgb_in.g, 1.0f), 0.0f), max(min(rgb_in.b, 1.0f), 0.0f)};
    int i = 0;
    for (int j = 1; j < 3; ++j)
        if (rgb[j] >= rgb[i]) i = j;
    Float z = rgb[i];
    if (z == 0.0f) {
        // Black is outside the published algorithm's domain (scale = inf, 0 * inf = NaN coefficients), yet WrapMode::Black and
        // RgbUnboundedSpectrum::new (spectrum.rs:543) feed it exactly that. Defined here as the identically-zero spectrum
        // s(-inf) = 0, which is what PBRT-v4's own table returns for black; the crate's behaviour is unpinned.
        out[0] = 0.0f; out[1] = 0.0f; out[2] = -infinity();
        return;
    }
    Float scale = (Float)(res - 1) / z;
    Float x = rgb[(i + 1) % 3] * scale, y = rgb[(i + 2) % 3] * scale;
    // (uint32_t) x of a NaN (z == 0: 0 * inf) is 0 with the saturating casts of Rust
    int xi = float_to_i32(x), yi = float_to_i32(y);
    if (xi < 0) xi = 0;
    if (yi < 0) yi = 0;
    if (xi > res - 2) xi = res - 2;
    if (yi > res - 2) yi = res - 2;
    // rgb2spec_find_interval: the last index with scale[index] <= z, clamped to [0, res - 2]
    int left = 0, last_interval = res - 2, size = last_interval;
    while (size > 0) {
        int half = size >> 1, middle = left + half + 1;
        if (sv.rgb2spec_scale[middle] <= z) { left = middle; size -= half + 1; }
        else size = half;
    }
    int zi = left < last_interval ? left : last_interval;
    uint32_t offset = (uint32_t)((((i * res + zi) * res + yi) * res + xi) * 3);
    uint32_t dx = 3u, dy = 3u * (uint32_t)res, dz = 3u * (uint32_t)res * (uint32_t)res;
    Float x1 = x - (Float)xi, x0 = 1.0f - x1, y1 = y - (Float)yi, y0 = 1.0f - y1;
    Float z1 = (z - sv.rgb2spec_scale[zi]) / (sv.rgb2spec_scale[zi + 1] - sv.rgb2spec_scale[zi]), z0 = 1.0f - z1;
    const Float* d = sv.rgb2spec_data;
    for (int j = 0; j < 3; ++j) {
        out[j] = ((d[offset] * x0 + d[offset + dx] * x1) * y0 + (d[offset + dy] * x0 + d[offset + dy + dx] * x1) * y1) * z0 +
                 ((d[offset + dz] * x0 + d[offset + dz + dx] * x1) * y0 + (d[offset + dz + dy] * x0 + d[offset + dz + dy + dx] * x1) * y1) * z1;
        offset++;
    }
}

// SpectrumImageTexture::evaluate, texture.rs:777-808
SHM_HD_NOINLINE Spec image_texture_evaluate_v(const SceneView* svp, uint32_t texture_index, SHM_TEXCTX_PARAMS, SHM_LAMBDA_PARAMS) {
    const SceneView& sv = *svp;
    SHM_TEXCTX_FROM_PARAMS(ctx);
    SHM_LAMBDA_FROM_PARAMS(lambda);
    TextureView tv;
    tv.t = &sv.image_textures[texture_index];
    tv.levels = sv.image_levels + tv.t->first_level;
    tv.texels = sv.texel_data;
    TexCoord2D c = texture_map(*tv.t, ctx, sv.quirks_off != 0);
    c.st.y = 1.0f - c.st.y;
    RGB3 rgb = tex_filter<RGB3>(tv, sv.ewa_lut, c.st, v2(c.dsdx, c.dtdx), v2(c.dsdy, c.dtdy)) * tv.t->scale;
    if (tv.t->invert) rgb = rgb3(1.0f, 1.0f, 1.0f) - rgb;
    rgb = rgb3(max(0.0f, rgb.r), max(0.0f, rgb.g), max(0.0f, rgb.b));  // clamp_zero, color.rs:204-210
    if (!tv.t->has_color_space) return spec_const(rgb.r);
    Float coeff[3];
    Float scale = 1.0f;
    if (tv.t->spectrum_type == SHM_SPECTRUM_TYPE_ALBEDO) {  // spectrum.rs:502-509
        rgb2spec_fetch(sv, rgb, coeff);
    } else {  // spectrum.rs:537-547, 573-588
        Float m = max(max(rgb.r, rgb.g), rgb.b);
        scale = 2.0f * m;
        if (scale != 0.0f) rgb2spec_fetch(sv, rgb / scale, coeff);
        else rgb2spec_fetch(sv, rgb3(0.0f, 0.0f, 0.0f), coeff);
    }
    Spec s;
    if (tv.t->spectrum_type == SHM_SPECTRUM_TYPE_ALBEDO) {
        for (int i = 0; i < NSPEC; ++i) s.v[i] = rgb_sigmoid(coeff, lambda.lambda[i]);
        return s;
    }
    for (int i = 0; i < NSPEC; ++i) s.v[i] = scale * rgb_sigmoid(coeff, lambda.lambda[i]);
    if (tv.t->spectrum_type == SHM_SPECTRUM_TYPE_UNBOUNDED) return s;
    return s * dense_table_sample(sv.cs_illuminant, lambda);  // spectrum.rs:600-606
}
SHM_HD Spec image_texture_evaluate(const SceneView& sv, uint32_t texture_index, const TextureEvalContext& ctx, const Wavelengths& lambda) {
    return image_texture_evaluate_v(SHM_SV_FOR_CALL(sv), texture_index, SHM_TEXCTX_ARGS(ctx), SHM_LAMBDA_ARGS(lambda));
}

// ---------------------------------------------------------------------------------------------
// ImageInfinitelight (light.rs:805-981): equal-area octahedral environment map, importance-sampled through a
// PiecewiseConstant2D (sampling.rs:20-179) — the compensated one under allow_incomplete_pdf (PathIntegrator).
// ---------------------------------------------------------------------------------------------
// fast_polynomial 0.1.0 `poly_array` with seven coefficients (math.rs:514). The crate is not vendored: parity unpinned; its
// documented scheme is Estrin's with FMAs, defined here as ((c6 x^2 + (c5 x + c4)) x^4 + ((c3 x + c2) x^2 + (c1 x + c0))).
SHM_HD Float poly7_estrin(Float x, const Float c[7]) {
    Float x2 = x * x, x4 = x2 * x2;
    Float p01 = fma(c[1], x, c[0]), p23 = fma(c[3], x, c[2]), p45 = fma(c[5], x, c[4]);
    Float lo = fma(p23, x2, p01), hi = fma(c[6], x2, p45);
    return fma(hi, x4, lo);
}
// math.rs:456-485. Kept as written: phi = (vp - up / r + 1) pi / 4 (PBRT-v4 has (vp - up) / r + 1).
SHM_HD V3 equal_area_square_to_sphere(V2 p) {
    Float u = 2.0f * p.x - 1.0f, v = 2.0f * p.y - 1.0f;
    Float up = abs(u), vp = abs(v);
    Float signed_distance = 1.0f - (up + vp);
    Float d = abs(signed_distance);
    Float r = 1.0f - d;
    Float phi = ((r == 0.0f) ? 1.0f : (vp - up / r + 1.0f)) * PI_F / 4.0f;
    Float z = copysign(1.0f - sqr(r), signed_distance);
    Float cos_phi = copysign(cos(phi), u), sin_phi = copysign(sin(phi), v);
    return v3(cos_phi * r * safe_sqrt(2.0f - sqr(r)), sin_phi * r * safe_sqrt(2.0f - sqr(r)), z);
}
// math.rs:488-540
SHM_HD V2 equal_area_sphere_to_square(V3 d) {
    Float x = abs(d.x), y = abs(d.y), z = abs(d.z);
    Float r = safe_sqrt(1.0f - z);
    Float a = max(x, y), b = min(x, y);
    b = (a == 0.0f) ? 0.0f : b / a;
    const Float t[7] = {0.406758566246788489601959989e-5f, 0.636226545274016134946890922156f, 0.61572017898280213493197203466e-2f,
                        -0.247333733281268944196501420480f, 0.881770664775316294736387951347e-1f, 0.419038818029165735901852432784e-1f,
                        -0.251390972343483509333252996350e-1f};
    Float phi = poly7_estrin(b, t);
    if (x < y) phi = 1.0f - phi;
    Float v = phi * r;
    Float u = r - v;
    if (d.z < 0.0f) {
        Float tmp = u; u = v; v = tmp;
        u = 1.0f - u;
        v = 1.0f - v;
    }
    u = copysign(u, d.x);
    v = copysign(v, d.y);
    return v2(0.5f * (u + 1.0f), 0.5f * (v + 1.0f));
}

// PiecewiseConstant1D::sample, sampling.rs:69-90 (min = 0, max = 1: lerp(t, 0, 1) = 0 * (1 - t) + 1 * t)
SHM_HD Float pc1d_sample(const Float* func, const Float* cdf, Float func_int, int n, Float u, Float& pdf, int& offset) {
    offset = find_interval(n + 1, [&](int i) { return cdf[i] <= u; });
    Float du = u - cdf[offset];
    if (cdf[offset + 1] - cdf[offset] > 0.0f) du /= cdf[offset + 1] - cdf[offset];
    pdf = (func_int > 0.0f) ? func[offset] / func_int : 0.0f;
    return lerp(((Float)offset + du) / (Float)n, 0.0f, 1.0f);
}
// PiecewiseConstant2D::sample, sampling.rs:160-168
SHM_HD V2 pc2d_sample(const Float* data, const Dist2DRec& d, int n, V2 u, Float& pdf) {
    Float pdf1, pdf0;
    int iv, iu;
    Float d1 = pc1d_sample(data + d.marginal_func, data + d.marginal_cdf, d.marginal_int, n, u.y, pdf1, iv);
    // the row's integral is the marginal's func value (sampling.rs:140-144)
    Float d0 = pc1d_sample(data + d.func + (uint32_t)(iv * n), data + d.cdf + (uint32_t)(iv * (n + 1)), data[d.marginal_func + (uint32_t)iv], n, u.x, pdf0, iu);
    pdf = pdf0 * pdf1;
    return v2(d0, d1);
}
// PiecewiseConstant2D::pdf, sampling.rs:170-177 (domain [0,1]^2: Bounds2f::offset is the identity)
SHM_HD Float pc2d_pdf(const Float* data, const Dist2DRec& d, int n, V2 p) {
    int iu = float_to_i32(p.x * (Float)n), iv = float_to_i32(p.y * (Float)n);  // `as usize`: saturating, negatives -> 0
    iu = iu < 0 ? 0 : (iu > n - 1 ? n - 1 : iu);
    iv = iv < 0 ? 0 : (iv > n - 1 ? n - 1 : iv);
    return data[d.func + (uint32_t)(iv * n + iu)] / d.marginal_int;
}
// ImageInfinitelight::image_le, light.rs:968-977 (lookup_nearest_channel_wrapped, image.rs:590-601) — without the light's scale
SHM_HD Spec image_light_le_uv(const SceneView& sv, const ImageLightRec& il, V2 uv, const Wavelengths& lambda) {
    const ShmImageLevel& l = sv.image_levels[il.image_level];
    int x = float_to_i32(uv.x * (Float)l.width), y = float_to_i32(uv.y * (Float)l.height);
    RGB3 rgb = rgb3(0.0f, 0.0f, 0.0f);
    if (remap_pixel_coords(x, y, l.width, l.height, SHM_WRAP_OCTAHEDRAL_SPHERE)) {
        const Float* t = sv.texel_data + l.texel_offset + (uint32_t)(3 * (y * l.width + x));
        rgb = rgb3(max(0.0f, t[0]), max(0.0f, t[1]), max(0.0f, t[2]));  // clamp_zero
    }
    // RgbIlluminantSpectrum::new + sample, spectrum.rs:573-606
    Float m = max(max(rgb.r, rgb.g), rgb.b);
    Float scale = 2.0f * m;
    Float coeff[3];
    if (scale != 0.0f) rgb2spec_fetch(sv, rgb / scale, coeff);
    else rgb2spec_fetch(sv, rgb3(0.0f, 0.0f, 0.0f), coeff);
    Spec s;
    for (int i = 0; i < NSPEC; ++i) s.v[i] = scale * rgb_sigmoid(coeff, lambda.lambda[i]);
    return s * dense_table_sample(sv.cs_illuminant, lambda);
}
// ImageInfinitelight::le, light.rs:900-904 (without the light's scale)
SHM_HD Spec image_light_le(const SceneView& sv, const ImageLightRec& il, V3 ray_d, const Wavelengths& lambda) {
    V3 w_light = xf_vector(il.light_from_render, ray_d);
    return image_light_le_uv(sv, il, equal_area_sphere_to_square(w_light), lambda);
}

// material::bump_map, material.rs:1477-1508, with a FloatTexture displacement (the constant-displacement form stays in get_bsdf).
// The shifted contexts are TextureEvalContext::from(&NormalBumpEvalContext): their n is the SHADING normal (material.rs:1428-1432).
SHM_HD void bump_map_texture(const SceneView& sv, uint32_t displacement, const SurfaceInteraction& si, const Differentials& df, V3& dpdu_out, V3& dpdv_out) {
    TextureEvalContext ctx;
    ctx.p = si.p(); ctx.dpdx = df.dpdx; ctx.dpdy = df.dpdy; ctx.n = si.shading.n; ctx.uv = si.uv;
    ctx.dudx = df.dudx; ctx.dudy = df.dudy; ctx.dvdx = df.dvdx; ctx.dvdy = df.dvdy;
    TextureEvalContext shifted = ctx;
    Float du = 0.5f * (abs(df.dudx) + abs(df.dudy));
    if (du == 0.0f) du = 0.0005f;
    shifted.p = ctx.p + du * si.shading.dpdu;
    shifted.uv = ctx.uv + v2(du, 0.0f);
    Float u_displace = float_texture_evaluate(sv, displacement, shifted);
    Float dv = 0.5f * (abs(df.dvdx) + abs(df.dvdy));
    if (dv == 0.0f) dv = 0.0005f;
    shifted.p = ctx.p + dv * si.shading.dpdv;
    shifted.uv = ctx.uv + v2(0.0f, dv);
    Float v_displace = float_texture_evaluate(sv, displacement, shifted);
    Float displace = float_texture_evaluate(sv, displacement, ctx);
    dpdu_out = si.shading.dpdu + (u_displace - displace) / du * si.shading.n + displace * si.shading.dndu;
    dpdv_out = si.shading.dpdv + (v_displace - displace) / dv * si.shading.n + displace * si.shading.dndv;
}
// material::normal_map, material.rs:1453-1475: the finest level of the image, repeat wrap, bilinear
SHM_HD void normal_map_texture(const SceneView& sv, uint32_t texture_index, const SurfaceInteraction& si, V3& dpdu_out, V3& dpdv_out) {
    TextureView tv = texture_view(sv, texture_index);
    ShmImageTexture repeat = *tv.t;
    repeat.wrap = SHM_WRAP_REPEAT;
    tv.t = &repeat;
    V2 uv = v2(si.uv.x, 1.0f - si.uv.y);
    V3 ns = v3(2.0f * tex_bilerp_channel(tv, 0, uv, 0) - 1.0f, 2.0f * tex_bilerp_channel(tv, 0, uv, 1) - 1.0f, 2.0f * tex_bilerp_channel(tv, 0, uv, 2) - 1.0f);
    ns = normalize(ns);
    Frame frame = frame_from_xz(normalize(si.shading.dpdu), si.shading.n);
    ns = frame.from_local(ns);
    Float ulen = length(si.shading.dpdu), vlen = length(si.shading.dpdv);
    dpdu_out = normalize(gram_schmidt(si.shading.dpdu, ns)) * ulen;
    dpdv_out = normalize(cross(ns, dpdu_out)) * vlen;
}

// A composite SpectrumTexture tree (scale / mix / directionmix over spectrum and image leaves), texture.rs:573-581, 628-645,
// 810-828, as a post-order program like the float ones
SHM_HD_NOINLINE Spec spectrum_texture_node_evaluate_v(const SceneView* svp, uint32_t index, SHM_TEXCTX_PARAMS, SHM_LAMBDA_PARAMS) {
    const SceneView& sv = *svp;
    SHM_TEXCTX_FROM_PARAMS(ctx);
    SHM_LAMBDA_FROM_PARAMS(lambda);
    const FloatTexRange r = sv.stex_ranges[index];
    if (r.count == 1u) {  // a single leaf: no value array (float_texture_evaluate_v)
        const ShmSpectrumTexture& t = sv.spectrum_textures[sv.stex_ops[r.first].node];
        if (t.kind == SHM_SPECTEX_LEAF)
            return (t.leaf.kind == SHM_SPECTRUM_IMAGE_TEXTURE) ? image_texture_evaluate(sv, t.leaf.offset, ctx, lambda) : spectrum_sample(t.leaf, sv.spectrum_data, lambda);
    }
    Spec vals[STEX_MAX_OPS];
    for (uint32_t k = 0; k < r.count; ++k) {
        const FloatTexOp op = sv.stex_ops[r.first + k];
        const ShmSpectrumTexture& t = sv.spectrum_textures[op.node];
        Spec v;
        if (t.kind == SHM_SPECTEX_LEAF) {
            v = (t.leaf.kind == SHM_SPECTRUM_IMAGE_TEXTURE) ? image_texture_evaluate(sv, t.leaf.offset, ctx, lambda)
                                                            : spectrum_sample(t.leaf, sv.spectrum_data, lambda);
        } else if (t.kind == SHM_SPECTEX_SCALED) {
            Float sc = float_texture_evaluate(sv, t.f, ctx);
            v = (sc == 0.0f) ? spec_const(0.0f) : vals[op.a] * sc;
        } else if (t.kind == SHM_SPECTEX_MIX) {
            Float amt = float_texture_evaluate(sv, t.f, ctx);
            Spec t1 = (amt != 1.0f) ? vals[op.a] : spec_const(0.0f);
            Spec t2 = (amt != 0.0f) ? vals[op.b] : spec_const(0.0f);
            v = t1 * (1.0f - amt) + t2 * amt;
        } else {
            Float amt = dot(ctx.n, ld3(t.dir));
            Spec t1 = (amt != 0.0f) ? vals[op.a] : spec_const(0.0f);
            Spec t2 = (amt != 1.0f) ? vals[op.b] : spec_const(0.0f);
            v = amt * t1 + (1.0f - amt) * t2;
        }
        vals[k] = v;
    }
    return vals[r.count - 1];
}
SHM_HD Spec spectrum_texture_node_evaluate(const SceneView& sv, uint32_t index, const TextureEvalContext& ctx, const Wavelengths& lambda) {
    return spectrum_texture_node_evaluate_v(SHM_SV_FOR_CALL(sv), index, SHM_TEXCTX_ARGS(ctx), SHM_LAMBDA_ARGS(lambda));
}

// SpectrumTexture::evaluate for a material slot: a constant spectrum texture samples its spectrum (texture.rs:509-513), an image
// texture filters its pyramid, a composite evaluates its tree. HAS_TEX = false compiles the texture branches out (scenes without
// textures: identical code as before).
template <bool HAS_TEX>
SHM_HD Spec spectrum_texture_evaluate(const SceneView& sv, const ShmSpectrum& s, const TextureEvalContext* ctx, const Wavelengths& lambda) {
    if (HAS_TEX && s.kind == SHM_SPECTRUM_IMAGE_TEXTURE) return image_texture_evaluate(sv, s.offset, *ctx, lambda);
    if (HAS_TEX && s.kind == SHM_SPECTRUM_TEXTURE_NODE) return spectrum_texture_node_evaluate(sv, s.offset, *ctx, lambda);
    return spectrum_sample(s, sv.spectrum_data, lambda);
}

}  // namespace shm
