// shm/shapes.h — ray/bounds, ray/triangle, ray/sphere, surface interactions, ray spawning, shape sampling.
//
// Restates (paths relative to /root/reference/src):
//   bounding_box.rs:520-563       Bounds3f::intersect_p_cached
//   shape/triangle.rs:162-302     solid_angle, intersect_triangle (watertight, f64 edge fallback)
//   shape/triangle.rs:305-504     interaction_from_intersection
//   shape/triangle.rs:537-745     area, sample, sample_with_context, pdf_with_context
//   shape/sphere.rs:95-271        basic_intersect (interval arithmetic), interaction_from_intersection
//   shape/sphere.rs:301-457       area, sample, sample_with_context, pdf_with_context (quirk 1)
//   transform.rs:363-800          Transform::apply for Point3f/Vector3f/Normal3f/Point3fi/Vector3fi/
//                                 SurfaceInteraction (quirk 6), apply_ray
//   interaction.rs:68-85,100-135,379-405  offset_ray_origin, SurfaceInteraction::new, set_shading_geometry
//   ray.rs:53-99                  offset_ray_origin, spawn_ray, spawn_ray_to_both_offset
#pragma once
#include "sampling.h"
#include "../../../include/shimmer_hip.h"

namespace shm {

struct Ray {
    V3 o, d;
};

// bounding_box.rs:520-563. bounds = {min, max}; dir_is_neg[i] in {0,1}.
SHM_HD bool intersect_p_cached(const Float bmin[3], const Float bmax[3], V3 o, Float ray_t_max, V3 inv_dir,
                               const int dir_is_neg[3]) {
    const Float g = 1.0f + 2.0f * gamma(3);
    Float t_min = ((dir_is_neg[0] ? bmax[0] : bmin[0]) - o.x) * inv_dir.x;
    Float t_max = ((dir_is_neg[0] ? bmin[0] : bmax[0]) - o.x) * inv_dir.x;
    Float ty_min = ((dir_is_neg[1] ? bmax[1] : bmin[1]) - o.y) * inv_dir.y;
    Float ty_max = ((dir_is_neg[1] ? bmin[1] : bmax[1]) - o.y) * inv_dir.y;
    t_max *= g;
    ty_max *= g;
    if (t_min > ty_max || ty_min > t_max) return false;
    if (ty_min > t_min) t_min = ty_min;
    if (ty_max < t_max) t_max = ty_max;
    Float tz_min = ((dir_is_neg[2] ? bmax[2] : bmin[2]) - o.z) * inv_dir.z;
    Float tz_max = ((dir_is_neg[2] ? bmin[2] : bmax[2]) - o.z) * inv_dir.z;
    tz_max *= g;
    if (t_min > tz_max || tz_min > t_max) return false;
    if (tz_min > t_min) t_min = tz_min;
    if (tz_max < t_max) t_max = tz_max;
    return t_min < ray_t_max && t_max > 0.0f;
}

struct TriangleIntersection {
    Float b0, b1, b2, t;
};

// shape/triangle.rs:197-219: the permutation and shear depend on the ray only ("it may be worth caching those with
// the ray", triangle.rs:221-222) — hoisted so that a traversal computes the three divisions once per ray. Same
// operations, same order: bit-identical to evaluating them per triangle.
struct RayShear {
    int kx, ky, kz;
    V3 d;  // permuted direction
    Float sx, sy, sz;
};
// permute((kx, ky, kz)) for the three cyclic cases kz = 0, 1, 2 -> (y,z,x), (z,x,y), (x,y,z): written as selects so
// that the GPU build uses v_cndmask instead of a branchy index switch (values are identical to V3::operator[]).
SHM_HD V3 permute_cyclic(V3 p, int kz) {
    const bool k0 = (kz == 0), k1 = (kz == 1);
    return v3(k0 ? p.y : (k1 ? p.z : p.x), k0 ? p.z : (k1 ? p.x : p.y), k0 ? p.x : (k1 ? p.y : p.z));
}
// max of three values that are |.| results (non-negative or NaN): IEEE maxNum, the semantics of Rust's f32::max that
// max_component_value uses (math.rs:112-117); the sign-of-zero ambiguity of maxNum cannot arise for |.| inputs.
SHM_HD Float max_abs3(Float a, Float b, Float c) { return __builtin_fmaxf(a, __builtin_fmaxf(b, c)); }
SHM_HD RayShear ray_shear(V3 rd) {
    RayShear r;
    r.kz = max_component_index(abs3(rd));
    r.kx = r.kz + 1;
    if (r.kx == 3) r.kx = 0;
    r.ky = r.kx + 1;
    if (r.ky == 3) r.ky = 0;
    r.d = permute_cyclic(rd, r.kz);
    r.sx = -r.d.x / r.d.z;
    r.sy = -r.d.y / r.d.z;
    r.sz = 1.0f / r.d.z;
    return r;
}

// triangle.rs:182-185: "return no intersection if the triangle is degenerate" — a property of the triangle alone
SHM_HD bool triangle_is_degenerate(V3 p0, V3 p1, V3 p2) { return length_squared(cross(p2 - p0, p1 - p0)) == 0.0f; }
// shape/triangle.rs:186-302 with the ray constants precomputed: everything after the degeneracy check (callers that hold the
// precomputed PRIM_DEGENERATE_BIT use this one; intersect_triangle_pre below is the whole function).
SHM_HD bool intersect_triangle_nondegenerate(V3 ro, const RayShear& rs, Float t_max, V3 p0, V3 p1, V3 p2, TriangleIntersection& out);
SHM_HD bool intersect_triangle_pre(V3 ro, const RayShear& rs, Float t_max, V3 p0, V3 p1, V3 p2, TriangleIntersection& out) {
    if (triangle_is_degenerate(p0, p1, p2)) return false;
    return intersect_triangle_nondegenerate(ro, rs, t_max, p0, p1, p2, out);
}
SHM_HD bool intersect_triangle_nondegenerate(V3 ro, const RayShear& rs, Float t_max, V3 p0, V3 p1, V3 p2, TriangleIntersection& out) {
    V3 p0t = p0 - ro, p1t = p1 - ro, p2t = p2 - ro;
    p0t = permute_cyclic(p0t, rs.kz);
    p1t = permute_cyclic(p1t, rs.kz);
    p2t = permute_cyclic(p2t, rs.kz);
    const Float sx = rs.sx, sy = rs.sy, sz = rs.sz;
    p0t.x += sx * p0t.z;
    p0t.y += sy * p0t.z;
    p1t.x += sx * p1t.z;
    p1t.y += sy * p1t.z;
    p2t.x += sx * p2t.z;
    p2t.y += sy * p2t.z;
    Float e0 = difference_of_products(p1t.x, p2t.y, p1t.y, p2t.x);
    Float e1 = difference_of_products(p2t.x, p0t.y, p2t.y, p0t.x);
    Float e2 = difference_of_products(p0t.x, p1t.y, p0t.y, p1t.x);
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {
        double p2txp1ty = (double)p2t.x * (double)p1t.y;
        double p2typ1tx = (double)p2t.y * (double)p1t.x;
        e0 = (Float)(p2typ1tx - p2txp1ty);
        double p0txp2ty = (double)p0t.x * (double)p2t.y;
        double p0typ2tx = (double)p0t.y * (double)p2t.x;
        e1 = (Float)(p0typ2tx - p0txp2ty);
        double p1txp0ty = (double)p1t.x * (double)p0t.y;
        double p1typ0tx = (double)p1t.y * (double)p0t.x;
        e2 = (Float)(p1typ0tx - p1txp0ty);
    }
    if ((e0 < 0.0f || e1 < 0.0f || e2 < 0.0f) && (e0 > 0.0f || e1 > 0.0f || e2 > 0.0f)) return false;
    Float det = e0 + e1 + e2;
    if (det == 0.0f) return false;
    p0t.z *= sz;
    p1t.z *= sz;
    p2t.z *= sz;
    Float t_scaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
    if (det < 0.0f && (t_scaled >= 0.0f || t_scaled < t_max * det)) return false;
    else if (det > 0.0f && (t_scaled <= 0.0f || t_scaled > t_max * det)) return false;
    Float inv_det = 1.0f / det;
    Float b0 = e0 * inv_det;
    Float b1 = e1 * inv_det;
    Float b2 = e2 * inv_det;
    Float t = t_scaled * inv_det;
    Float max_zt = max_abs3(abs(p0t.z), abs(p1t.z), abs(p2t.z));
    Float delta_z = gamma(3) * max_zt;
    Float max_xt = max_abs3(abs(p0t.x), abs(p1t.x), abs(p2t.x));
    Float max_yt = max_abs3(abs(p0t.y), abs(p1t.y), abs(p2t.y));
    Float delta_x = gamma(5) * (max_xt + max_zt);
    Float delta_y = gamma(5) * (max_yt + max_zt);
    Float delta_e = 2.0f * (gamma(2) * max_xt * max_yt + delta_y * max_xt + delta_x * max_yt);
    Float max_e = max_abs3(abs(e0), abs(e1), abs(e2));
    Float delta_t = 3.0f * (gamma(3) * max_e * max_zt + delta_e * max_zt + delta_z * max_e) * abs(inv_det);
    if (t <= delta_t) return false;
    out.b0 = b0; out.b1 = b1; out.b2 = b2; out.t = t;
    return true;
}
SHM_HD bool intersect_triangle(V3 ro, V3 rd, Float t_max, V3 p0, V3 p1, V3 p2, TriangleIntersection& out) {
    return intersect_triangle_pre(ro, ray_shear(rd), t_max, p0, p1, p2, out);
}

// interaction.rs:24-31 + 87-98
struct Shading {
    V3 n, dpdu, dpdv, dndu, dndv;
};
struct SurfaceInteraction {
    P3i pi;
    V3 wo, n;
    V2 uv;
    V3 dpdu, dpdv, dndu, dndv;
    Shading shading;
    SHM_HD V3 p() const { return pi.mid(); }
};

// interaction.rs:100-135
SHM_HD SurfaceInteraction surface_interaction_new(P3i pi, V2 uv, V3 wo, V3 dpdu, V3 dpdv, V3 dndu, V3 dndv,
                                                  bool flip_normal) {
    Float normal_sign = flip_normal ? -1.0f : 1.0f;
    V3 normal = normal_sign * normalize(cross(dpdu, dpdv));
    SurfaceInteraction si;
    si.pi = pi; si.wo = wo; si.n = normal; si.uv = uv;
    si.dpdu = dpdu; si.dpdv = dpdv; si.dndu = dndu; si.dndv = dndv;
    si.shading.n = normal; si.shading.dpdu = dpdu; si.shading.dpdv = dpdv;
    si.shading.dndu = dndu; si.shading.dndv = dndv;
    return si;
}
// interaction.rs:379-405
SHM_HD void set_shading_geometry(SurfaceInteraction& si, V3 ns, V3 dpdus, V3 dpdvs, V3 dndus, V3 dndvs,
                                 bool orientation_is_authoritative) {
    si.shading.n = ns;
    if (orientation_is_authoritative) si.n = face_forward(si.n, si.shading.n);
    else si.shading.n = face_forward(si.shading.n, si.n);
    si.shading.dpdu = dpdus;
    si.shading.dpdv = dpdvs;
    si.shading.dndu = dndus;
    si.shading.dndv = dndvs;
    while (length_squared(si.shading.dpdu) > 1e16f || length_squared(si.shading.dpdv) > 1e16f) {
        si.shading.dpdu = si.shading.dpdu / 1e8f;
        si.shading.dpdv = si.shading.dpdv / 1e8f;
    }
}

// Per-triangle inputs gathered by the caller from the mesh arrays (shape/mesh.rs:9-20).
struct TriangleData {
    V3 p0, p1, p2;
    bool has_uv, has_n, has_s;
    bool flip;  // reverse_orientation ^ transform_swaps_handedness
    V2 uv0, uv1, uv2;
    V3 n0, n1, n2;
    V3 s0, s1, s2;
};

// shape/triangle.rs:305-504
SHM_HD SurfaceInteraction triangle_interaction(const TriangleData& tr, const TriangleIntersection& ti, V3 wo) {
    V3 p0 = tr.p0, p1 = tr.p1, p2 = tr.p2;
    V2 uv[3];
    if (!tr.has_uv) { uv[0] = v2(0.0f, 0.0f); uv[1] = v2(1.0f, 0.0f); uv[2] = v2(1.0f, 1.0f); }
    else { uv[0] = tr.uv0; uv[1] = tr.uv1; uv[2] = tr.uv2; }
    V2 duv02 = uv[0] - uv[2];
    V2 duv12 = uv[1] - uv[2];
    V3 dp02 = p0 - p2;
    V3 dp12 = p1 - p2;
    Float determinant = difference_of_products(duv02.x, duv12.y, duv02.y, duv12.x);
    bool degenerate_uv = abs(determinant) < 1e-9f;
    V3 dpdu = v3s(0.0f), dpdv = v3s(0.0f);
    if (!degenerate_uv) {
        Float inv_det = 1.0f / determinant;
        // math.rs:198-203 difference_of_products_float_vec(a, b, c, d): cd = c*d; (a*b - cd) + (-c*d + cd)
        {
            V3 cd = duv02.y * dp12;
            V3 difference = duv12.y * dp02 - cd;
            V3 error = -duv02.y * dp12 + cd;
            dpdu = (difference + error) * inv_det;
        }
        {
            V3 cd = duv12.x * dp02;
            V3 difference = duv02.x * dp12 - cd;
            V3 error = -duv12.x * dp02 + cd;
            dpdv = (difference + error) * inv_det;
        }
    }
    if (degenerate_uv || length_squared(cross(dpdu, dpdv)) == 0.0f) {
        V3 ng = cross(p2 - p0, p1 - p0);
        if (length_squared(ng) == 0.0f) {
            V3 a = p2 - p0, b = p1 - p0;
            ng = v3((Float)difference_of_products64((double)a.y, (double)b.z, (double)a.z, (double)b.y),
                    (Float)difference_of_products64((double)a.z, (double)b.x, (double)a.x, (double)b.z),
                    (Float)difference_of_products64((double)a.x, (double)b.y, (double)a.y, (double)b.x));
        }
        coordinate_system(normalize(ng), dpdu, dpdv);
    }
    V3 p_hit = ti.b0 * p0 + ti.b1 * p1 + ti.b2 * p2;
    V2 uv_hit = ti.b0 * uv[0] + ti.b1 * uv[1] + ti.b2 * uv[2];
    bool flip_normal = tr.flip;
    V3 p_abs_sum = abs3(ti.b0 * p0) + abs3(ti.b1 * p1) + abs3(ti.b2 * p2);
    V3 p_error = gamma(7) * p_abs_sum;
    SurfaceInteraction isect = surface_interaction_new(p3i_from_value_and_error(p_hit, p_error), uv_hit, wo, dpdu,
                                                       dpdv, v3s(0.0f), v3s(0.0f), flip_normal);
    isect.shading.n = normalize(cross(dp02, dp12));
    isect.n = isect.shading.n;
    if (tr.flip) {
        isect.shading.n = -isect.shading.n;
        isect.n = -isect.n;
    }
    if (tr.has_n || tr.has_s) {
        V3 ns;
        if (!tr.has_n) ns = isect.n;
        else {
            V3 n = ti.b0 * tr.n0 + ti.b1 * tr.n1 + ti.b2 * tr.n2;
            ns = (length_squared(n) > 0.0f) ? normalize(n) : isect.n;
        }
        V3 ss;
        if (!tr.has_s) ss = isect.dpdu;
        else {
            V3 s = ti.b0 * tr.s0 + ti.b1 * tr.s1 + ti.b2 * tr.s2;
            ss = (length_squared(s) == 0.0f) ? isect.dpdu : s;
        }
        V3 ts = cross(ns, ss);
        if (length_squared(ts) > 0.0f) ss = cross(ts, ns);
        else coordinate_system(ns, ss, ts);
        V3 dndu, dndv;
        if (!tr.has_n) { dndu = v3s(0.0f); dndv = v3s(0.0f); }
        else {
            V2 d02 = uv[0] - uv[2];
            V2 d12 = uv[1] - uv[2];
            Float det = difference_of_products(d02.x, d12.y, d02.y, d12.x);
            bool deg = abs(det) < 1e-9f;
            if (deg) {
                V3 dn = cross(tr.n2 - tr.n0, tr.n1 - tr.n0);
                if (length_squared(dn) == 0.0f) { dndu = v3s(0.0f); dndv = v3s(0.0f); }
                else coordinate_system(dn, dndu, dndv);
            } else {
                Float inv_det = 1.0f / det;
                V3 dn1 = tr.n0 - tr.n2;
                V3 dn2 = tr.n1 - tr.n2;
                {
                    V3 cd = d02.y * dn2;
                    V3 difference = d12.y * dn1 - cd;
                    V3 error = -d02.y * dn2 + cd;
                    dndu = (difference + error) * inv_det;
                }
                {
                    V3 cd = d12.x * dn1;
                    V3 difference = d02.x * dn2 - cd;
                    V3 error = -d12.x * dn1 + cd;
                    dndv = (difference + error) * inv_det;
                }
            }
        }
        set_shading_geometry(isect, ns, ss, ts, dndu, dndv, true);
    }
    return isect;
}

// ray.rs:53-71
SHM_HD V3 offset_ray_origin(const P3i& pi, V3 n, V3 w) {
    Float d = dot(abs3(n), pi.error());
    V3 offset = d * n;
    if (dot(w, n) < 0.0f) offset = -offset;
    V3 po = pi.mid() + offset;
    for (int i = 0; i < 3; ++i) {
        Float o = offset[i];
        if (o > 0.0f) v3_set(po, i, next_float_up(po[i]));
        else if (o < 0.0f) v3_set(po, i, next_float_down(po[i]));
    }
    return po;
}
// ray.rs:83-99
SHM_HD Ray spawn_ray_to_both_offset(const P3i& p_from, V3 n_from, const P3i& p_to, V3 n_to) {
    V3 pf = offset_ray_origin(p_from, n_from, p_to.mid() - p_from.mid());
    V3 pt = offset_ray_origin(p_to, n_to, pf - p_to.mid());
    Ray r;
    r.o = pf;
    r.d = pt - pf;
    return r;
}

// ---------------------------------------------------------------------------------------------
// Transform (transform.rs). Row-major 4x4 m[r*4+c].
// ---------------------------------------------------------------------------------------------
SHM_HD V3 xf_point(const Float* m, V3 p) {  // transform.rs apply_point_helper
    Float xp = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
    Float yp = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
    Float zp = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
    Float wp = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
    if (wp == 1.0f) return v3(xp, yp, zp);
    return v3(xp, yp, zp) / wp;
}
SHM_HD V3 xf_vector(const Float* m, V3 v) {  // apply_vector_helper
    return v3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z,
              m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
SHM_HD V3 xf_normal(const Float* minv, V3 n) {  // apply_normal_helper(&m_inv, n): transpose of the inverse
    return v3(minv[0] * n.x + minv[4] * n.y + minv[8] * n.z, minv[1] * n.x + minv[5] * n.y + minv[9] * n.z,
              minv[2] * n.x + minv[6] * n.y + minv[10] * n.z);
}
// transform.rs:384-450 TransformI<Point3fi>
SHM_HD P3i xf_point_i(const Float* m, const P3i& val) {
    Float x = val.x.midpoint(), y = val.y.midpoint(), z = val.z.midpoint();
    Float xp = (m[0] * x + m[1] * y) + (m[2] * z + m[3]);
    Float yp = (m[4] * x + m[5] * y) + (m[6] * z + m[7]);
    Float zp = (m[8] * x + m[9] * y) + (m[10] * z + m[11]);
    Float wp = (m[12] * x + m[13] * y) + (m[14] * z + m[15]);
    V3 p_error;
    if (val.is_exact()) {
        p_error.x = gamma(3) * (abs(m[0] * x) + abs(m[1] * y) + abs(m[2] * z) + abs(m[3]));
        p_error.y = gamma(3) * (abs(m[4] * x) + abs(m[5] * y) + abs(m[6] * z) + abs(m[7]));
        p_error.z = gamma(3) * (abs(m[8] * x) + abs(m[9] * y) + abs(m[10] * z) + abs(m[11]));
    } else {
        V3 e = val.error();
        p_error.x = (gamma(3) + 1.0f) * (abs(m[0]) * e.x + abs(m[1]) * e.y + abs(m[2]) * e.z)
                    + gamma(3) * (abs(m[0] * x) + abs(m[1] * y) + abs(m[2] * z) + abs(m[3]));
        p_error.y = (gamma(3) + 1.0f) * (abs(m[4]) * e.x + abs(m[5]) * e.y + abs(m[6]) * e.z)
                    + gamma(3) * (abs(m[4] * x) + abs(m[5] * y) + abs(m[6] * z) + abs(m[7]));
        p_error.z = (gamma(3) + 1.0f) * (abs(m[8]) * e.x + abs(m[9]) * e.y + abs(m[10]) * e.z)
                    + gamma(3) * (abs(m[8] * x) + abs(m[9] * y) + abs(m[10] * z) + abs(m[11]));
    }
    P3i r = p3i_from_value_and_error(v3(xp, yp, zp), p_error);
    if (wp == 1.0f) return r;
    return r / iv(wp);
}
// transform.rs:452-500 TransformI<Vector3fi>
SHM_HD P3i xf_vector_i(const Float* m, const P3i& val) {
    Float x = val.x.midpoint(), y = val.y.midpoint(), z = val.z.midpoint();
    V3 err;
    if (val.is_exact()) {
        err.x = gamma(3) * (abs(m[0] * x) + abs(m[1] * y) + abs(m[2] * z));
        err.y = gamma(3) * (abs(m[4] * x) + abs(m[5] * y) + abs(m[6] * z));
        err.z = gamma(3) * (abs(m[8] * x) + abs(m[9] * y) + abs(m[10] * z));
    } else {
        V3 e = val.error();
        err.x = (gamma(3) + 1.0f) * (abs(m[0]) * e.x + abs(m[1]) * e.y + abs(m[2]) * e.z)
                + gamma(3) * (abs(m[0] * x) + abs(m[1] * y) + abs(m[2] * z));
        err.y = (gamma(3) + 1.0f) * (abs(m[4]) * e.x + abs(m[5]) * e.y + abs(m[6]) * e.z)
                + gamma(3) * (abs(m[4] * x) + abs(m[5] * y) + abs(m[6] * z));
        err.z = (gamma(3) + 1.0f) * (abs(m[8]) * e.x + abs(m[9]) * e.y + abs(m[10]) * e.z)
                + gamma(3) * (abs(m[8] * x) + abs(m[9] * y) + abs(m[10] * z));
    }
    Float xp = m[0] * x + m[1] * y + m[2] * z;
    Float yp = m[4] * x + m[5] * y + m[6] * z;
    Float zp = m[8] * x + m[9] * y + m[10] * z;
    return p3i_from_value_and_error(v3(xp, yp, zp), err);
}
// transform.rs:502-519 TransformRayI<Ray>::apply_ray with t_max = None.
// `self.apply(val.o).into()`: the Point3f overload, then an exact Point3fi; o + (d*dt) is an interval sum.
SHM_HD Ray xf_ray(const Float* m, const Ray& val) {
    P3i o = p3i_exact(xf_point(m, val.o));
    V3 d = xf_vector(m, val.d);
    Float ls = length_squared(d);
    if (ls > 0.0f) {
        Float dt = dot(abs3(d), o.error()) / ls;
        o = o + p3i_exact(d * dt);
    }
    Ray r;
    r.o = o.mid();
    r.d = d;
    return r;
}
// transform.rs:556-608 TransformI<SurfaceInteraction> (quirk 6: t = self.inverse() is used for everything
// but the point). m = self.m, minv = self.m_inv; t.m = minv, t.m_inv = m. `strict` (ShmRenderParams::disable_reference_quirks, round 6): PBRT-v4's Transform::operator()
// (SurfaceInteraction) — vectors through m, normals through the transpose of m_inv.
SHM_HD SurfaceInteraction xf_surface_interaction(const Float* m, const Float* minv, const SurfaceInteraction& v, bool strict = false) {
    if (strict) {
        const Float* t = m;
        m = minv;     // (below, "minv" maps vectors and "m" maps normals: swapped, they are the forward maps)
        minv = t;
        SurfaceInteraction r;
        V3 n = normalize(xf_normal(m, v.n));
        r.pi = xf_point_i(minv, v.pi);
        r.wo = normalize(xf_vector(minv, v.wo));
        r.n = n;
        r.uv = v.uv;
        r.dpdu = xf_vector(minv, v.dpdu);
        r.dpdv = xf_vector(minv, v.dpdv);
        r.dndu = xf_normal(m, v.dndu);
        r.dndv = xf_normal(m, v.dndv);
        r.shading.n = face_forward(normalize(xf_normal(m, v.shading.n)), n);
        r.shading.dpdu = xf_vector(minv, v.shading.dpdu);
        r.shading.dpdv = xf_vector(minv, v.shading.dpdv);
        r.shading.dndu = xf_normal(m, v.shading.dndu);
        r.shading.dndv = xf_normal(m, v.shading.dndv);
        return r;
    }
    SurfaceInteraction r;
    V3 n = normalize(xf_normal(m, v.n));            // t.apply(Normal) uses t.m_inv = m
    r.pi = xf_point_i(m, v.pi);
    r.wo = normalize(xf_vector(minv, v.wo));        // t.apply(Vector) uses t.m = minv
    r.n = n;
    r.uv = v.uv;
    r.dpdu = xf_vector(minv, v.dpdu);
    r.dpdv = xf_vector(minv, v.dpdv);
    r.dndu = xf_normal(m, v.dndu);
    r.dndv = xf_normal(m, v.dndv);
    r.shading.n = face_forward(normalize(xf_normal(m, v.shading.n)), n);
    r.shading.dpdu = xf_vector(minv, v.shading.dpdu);
    r.shading.dpdv = xf_vector(minv, v.shading.dpdv);
    r.shading.dndu = xf_normal(m, v.shading.dndu);
    r.shading.dndv = xf_normal(m, v.shading.dndv);
    return r;
}

// ---------------------------------------------------------------------------------------------
// Sphere (shape/sphere.rs)
// ---------------------------------------------------------------------------------------------
struct QuadricIntersection {
    Float t_hit;
    V3 p_obj;
    Float phi;
};

// shape/sphere.rs:95-196
SHM_HD bool sphere_basic_intersect(const ShmSphere& s, V3 ro, V3 rd, Float t_max, QuadricIntersection& out) {
    P3i oi = xf_point_i(s.object_from_render, p3i_exact(ro));
    P3i di = xf_vector_i(s.object_from_render, p3i_exact(rd));
    Interval a = iv_sqr(di.x) + iv_sqr(di.y) + iv_sqr(di.z);
    Interval b = 2.0f * (di.x * oi.x + di.y * oi.y + di.z * oi.z);
    Interval c = iv_sqr(oi.x) + iv_sqr(oi.y) + iv_sqr(oi.z) - iv_sqr(iv(s.radius));
    P3i v = oi - (b / (2.0f * a)) * di;
    Interval len = length(v);
    Interval discrim = 4.0f * a * (iv(s.radius) + len) * (iv(s.radius) - len);
    if (discrim.low < 0.0f) return false;
    Interval root_discrim = iv_sqrt(discrim);
    Interval q;
    if (b.midpoint() < 0.0f) q = -0.5f * (b - root_discrim);
    else q = -0.5f * (b + root_discrim);
    Interval t0 = q / a;
    Interval t1 = c / q;
    if (t0.low > t1.low) { Interval tmp = t0; t0 = t1; t1 = tmp; }
    if (t0.high > t_max || t1.low <= 0.0f) return false;
    Interval t_shape_hit = t0;
    if (t_shape_hit.low <= 0.0f) {
        t_shape_hit = t1;
        if (t_shape_hit.high > t_max) return false;
    }
    V3 p_hit = oi.mid() + t_shape_hit.midpoint() * di.mid();
    p_hit = p_hit * (s.radius / distance(p_hit, v3s(0.0f)));
    if (p_hit.x == 0.0f && p_hit.y == 0.0f) p_hit.x = 1e-5f * s.radius;
    Float phi = atan2(p_hit.y, p_hit.x);
    if (phi < 0.0f) phi += 2.0f * PI_F;
    if ((s.z_min > -s.radius && p_hit.z < s.z_min) || (s.z_max < s.radius && p_hit.z > s.z_max)
        || phi > s.phi_max) {
        if (t_shape_hit == t1) return false;
        if (t1.high > t_max) return false;
        t_shape_hit = t1;
        p_hit = oi.mid() + t_shape_hit.midpoint() * di.mid();
        p_hit = p_hit * (s.radius / distance(p_hit, v3s(0.0f)));
        if (p_hit.x == 0.0f && p_hit.y == 0.0f) p_hit.x = 1e-5f * s.radius;
        phi = atan2(p_hit.y, p_hit.x);
        if (phi < 0.0f) phi += 2.0f * PI_F;
        if ((s.z_min > -s.radius && p_hit.z < s.z_min) || (s.z_max < s.radius && p_hit.z > s.z_max)
            || phi > s.phi_max)
            return false;
    }
    out.t_hit = t_shape_hit.midpoint();
    out.p_obj = p_hit;
    out.phi = phi;
    return true;
}

// shape/sphere.rs:198-271
SHM_HD SurfaceInteraction sphere_interaction(const ShmSphere& s, const QuadricIntersection& isect, V3 wo, bool strict = false) {
    V3 p_hit = isect.p_obj;
    Float phi = isect.phi;
    Float u = phi / s.phi_max;
    Float cos_theta_ = p_hit.z / s.radius;
    Float theta = strict ? acos(clamp(cos_theta_, -1.0f, 1.0f)) : safe_acos(cos_theta_);  // (quirk 5: safe_acos is asin in the reference)
    Float v = (theta - s.theta_z_min) / (s.theta_z_max - s.theta_z_min);
    Float z_radius = sqrt(p_hit.x * p_hit.x + p_hit.y * p_hit.y);
    Float cos_phi_ = p_hit.x / z_radius;
    Float sin_phi_ = p_hit.y / z_radius;
    V3 dpdu = v3(-s.phi_max * p_hit.y, s.phi_max * p_hit.x, 0.0f);
    Float sin_theta_ = safe_sqrt(1.0f - cos_theta_ * cos_theta_);
    V3 dpdv = (s.theta_z_max - s.theta_z_min) * v3(p_hit.z * cos_phi_, p_hit.z * sin_phi_, -s.radius * sin_theta_);
    V3 d2pduu = -s.phi_max * s.phi_max * v3(p_hit.x, p_hit.y, 0.0f);
    V3 d2pduv = (s.theta_z_max - s.theta_z_min) * p_hit.z * s.phi_max * v3(-sin_phi_, cos_phi_, 0.0f);
    V3 d2pdvv = -((s.theta_z_max - s.theta_z_min) * (s.theta_z_max - s.theta_z_min)) * v3(p_hit.x, p_hit.y, p_hit.z);
    Float e1 = dot(dpdu, dpdu);
    Float f1 = dot(dpdu, dpdv);
    Float g1 = dot(dpdv, dpdv);
    V3 n = normalize(cross(dpdu, dpdv));
    Float e = dot(n, d2pduu);
    Float f = dot(n, d2pduv);
    Float g = dot(n, d2pdvv);
    Float egf2 = difference_of_products(e1, g1, f1, f1);
    Float env_egf2 = (egf2 == 0.0f) ? 0.0f : 1.0f / egf2;
    V3 dndu = (f * f1 - e * g1) * env_egf2 * dpdu + (e * f1 - f * e1) * env_egf2 * dpdv;
    V3 dndv = (g * f1 - f * g1) * env_egf2 * dpdu + (f * f1 - g * e1) * env_egf2 * dpdv;
    V3 p_error = gamma(5) * abs3(p_hit);
    bool flip_normal = (s.reverse_orientation != 0) ^ (s.transform_swaps_handedness != 0);
    V3 wo_object = xf_vector(s.object_from_render, wo);
    SurfaceInteraction si = surface_interaction_new(p3i_from_value_and_error(p_hit, p_error), v2(u, v), wo_object,
                                                    dpdu, dpdv, dndu, dndv, flip_normal);
    return xf_surface_interaction(s.render_from_object, s.object_from_render, si, strict);
}

// ---------------------------------------------------------------------------------------------
// Shape sampling (light sampling)
// ---------------------------------------------------------------------------------------------
struct ShapeSampleContext {  // shape/shape.rs:240-287
    P3i pi;
    V3 n, ns;
    SHM_HD V3 p() const { return pi.mid(); }
};
struct ShapeSample {  // shape/shape.rs ShapeSample {intr: Interaction, pdf}
    P3i pi;
    V3 n;
    Float pdf;
};

SHM_HD Float triangle_area(const TriangleData& tr) { return 0.5f * length(cross(tr.p1 - tr.p0, tr.p2 - tr.p0)); }
SHM_HD Float triangle_solid_angle(const TriangleData& tr, V3 p) {  // triangle.rs:162-169
    return spherical_triangle_area(normalize(tr.p0 - p), normalize(tr.p1 - p), normalize(tr.p2 - p));
}
// triangle.rs:543-593 (the n.is_empty() branch always negates: reference behaviour preserved)
// (`strict`: ShmRenderParams::disable_reference_quirks — PBRT-v4's form of the two reference behaviours below that bias an image: a mesh without normals flips the sampled
//  normal only with reverse_orientation ^ transform_swaps_handedness, and the spherical sample is drawn from the WARPED u whose density it is given. Default: as the reference.)
SHM_HD ShapeSample triangle_sample(const TriangleData& tr, V2 u, bool strict = false) {
    Float b0, b1, b2;
    sample_uniform_triangle(u, b0, b1, b2);
    V3 p = b0 * tr.p0 + b1 * tr.p1 + b2 * tr.p2;
    V3 n = normalize(cross(tr.p1 - tr.p0, tr.p2 - tr.p0));
    if (!tr.has_n) {  // triangle.rs:558-560 negates ALWAYS (a one-sided emitter without normals then shows its area samples its dark side: no next-event estimation from it)
        if (!strict || tr.flip) n = n * -1.0f;
    } else {
        V3 ns = b0 * tr.n0 + b1 * tr.n1 + b2 * tr.n2;
        n = face_forward(n, ns);
    }
    V3 p_abs_sum = abs3(b0 * tr.p0) + abs3(b1 * tr.p1) + abs3(b2 * tr.p2);
    V3 p_error = gamma(6) * p_abs_sum;
    ShapeSample ss;
    ss.pi = p3i_from_value_and_error(p, p_error);
    ss.n = n;
    ss.pdf = 1.0f / triangle_area(tr);
    return ss;
}
constexpr Float MIN_SPHERICAL_SAMPLE_AREA = 3e-4f;
constexpr Float MAX_SPHERICAL_SAMPLE_AREA = 6.22f;
// triangle.rs:595-694
SHM_HD bool triangle_sample_with_context(const TriangleData& tr, const ShapeSampleContext& ctx, V2 u, ShapeSample& out, bool strict = false) {
    Float solid_angle = triangle_solid_angle(tr, ctx.p());
    if (solid_angle < MIN_SPHERICAL_SAMPLE_AREA || solid_angle > MAX_SPHERICAL_SAMPLE_AREA) {
        ShapeSample ss = triangle_sample(tr, u, strict);
        V3 wi = ss.pi.mid() - ctx.p();
        if (length_squared(wi) == 0.0f) return false;
        wi = normalize(wi);
        ss.pdf /= abs_dot(ss.n, -wi) / distance_squared(ctx.p(), ss.pi.mid());
        if (is_inf(ss.pdf)) return false;
        out = ss;
        return true;
    }
    Float pdf = 1.0f;
    if (ctx.ns != v3s(0.0f)) {
        V3 rp = ctx.p();
        V3 wi0 = normalize(tr.p0 - rp), wi1 = normalize(tr.p1 - rp), wi2 = normalize(tr.p2 - rp);
        Float w[4] = {max(0.01f, abs_dot(ctx.ns, wi1)), max(0.01f, abs_dot(ctx.ns, wi1)),
                      max(0.01f, abs_dot(ctx.ns, wi0)), max(0.01f, abs_dot(ctx.ns, wi2))};
        // triangle.rs:639-641: the warped u shadows only inside this block; the spherical sample below
        // uses the ORIGINAL u (reference behaviour preserved).
        V2 uw = sample_bilinear(u, w);
        pdf = bilinear_pdf(uw, w);
        if (strict) u = uw;  // (PBRT-v4: the sample the density belongs to; tests/test_direct_lighting_analytic.py measures what the shadowing does to an image)
    }
    V3 verts[3] = {tr.p0, tr.p1, tr.p2};
    Float b[3];
    Float tri_pdf = sample_spherical_triangle(verts, ctx.p(), u, b, strict);
    if (tri_pdf == 0.0f) return false;
    pdf = pdf * tri_pdf;
    V3 p_abs_sum = abs3(b[0] * tr.p0) + abs3(b[1] * tr.p1) + abs3((1.0f - b[0] - b[1]) * tr.p2);
    V3 p_error = gamma(6) * p_abs_sum;
    V3 p = b[0] * tr.p0 + b[1] * tr.p1 + b[2] * tr.p2;
    V3 n = normalize(cross(tr.p1 - tr.p0, tr.p2 - tr.p0));
    if (tr.has_n) {
        V3 ns = b[0] * tr.n0 + b[1] * tr.n1 + b[2] * tr.n2;
        n = face_forward(n, ns);
    } else if (tr.flip) {
        n = n * -1.0f;
    }
    out.pi = p3i_from_value_and_error(p, p_error);
    out.n = n;
    out.pdf = pdf;
    return true;
}
// triangle.rs:696-745
SHM_HD Float triangle_pdf_with_context(const TriangleData& tr, const ShapeSampleContext& ctx, V3 wi) {
    Float solid_angle = triangle_solid_angle(tr, ctx.p());
    if (solid_angle < MIN_SPHERICAL_SAMPLE_AREA || solid_angle > MAX_SPHERICAL_SAMPLE_AREA) {
        V3 o = offset_ray_origin(ctx.pi, ctx.n, wi);  // ctx.spawn_ray(wi)
        TriangleIntersection ti;
        if (!intersect_triangle(o, wi, infinity(), tr.p0, tr.p1, tr.p2, ti)) return 0.0f;
        SurfaceInteraction isect = triangle_interaction(tr, ti, -wi);
        Float pdf = (1.0f / triangle_area(tr)) / (abs_dot(isect.n, -wi) / distance_squared(ctx.p(), isect.p()));
        if (is_inf(pdf)) return 0.0f;
        return pdf;
    }
    Float pdf = 1.0f / solid_angle;
    if (ctx.ns != v3s(0.0f)) {
        V3 verts[3] = {tr.p0, tr.p1, tr.p2};
        V2 u = invert_spherical_triangle_sample(verts, ctx.p(), wi);
        V3 rp = ctx.p();
        V3 wi0 = normalize(tr.p0 - rp), wi1 = normalize(tr.p1 - rp), wi2 = normalize(tr.p2 - rp);
        Float w[4] = {max(0.01f, abs_dot(ctx.ns, wi1)), max(0.01f, abs_dot(ctx.ns, wi1)),
                      max(0.01f, abs_dot(ctx.ns, wi0)), max(0.01f, abs_dot(ctx.ns, wi2))};
        pdf *= bilinear_pdf(u, w);
    }
    return pdf;
}

SHM_HD Float sphere_area(const ShmSphere& s) { return s.phi_max * s.radius * (s.z_max - s.z_min); }  // sphere.rs:301-303
// sphere.rs:305-337
SHM_HD ShapeSample sphere_sample(const ShmSphere& s, V2 u) {
    V3 p_obj = v3s(0.0f) + sample_uniform_sphere(u) * s.radius;
    p_obj = p_obj * (s.radius / distance(p_obj, v3s(0.0f)));
    V3 p_obj_error = gamma(5) * abs3(p_obj);
    V3 n_obj = p_obj;
    Float normal_sign = s.reverse_orientation ? -1.0f : 1.0f;
    V3 n = normal_sign * normalize(xf_normal(s.object_from_render, n_obj));  // render_from_object.apply(Normal) uses m_inv
    ShapeSample ss;
    ss.pi = xf_point_i(s.render_from_object, p3i_from_value_and_error(p_obj, p_obj_error));
    ss.n = n;
    ss.pdf = 1.0f / sphere_area(s);
    return ss;
}
// sphere.rs:339-422
SHM_HD bool sphere_sample_with_context(const ShmSphere& s, const ShapeSampleContext& ctx, V2 u, ShapeSample& out) {
    V3 p_center = xf_point(s.render_from_object, v3s(0.0f));
    V3 p_origin = offset_ray_origin(ctx.pi, ctx.n, p_center - ctx.p());
    if (distance_squared(p_origin, p_center) <= sqr(s.radius)) {
        ShapeSample ss = sphere_sample(s, u);
        V3 wi = ss.pi.mid() - ctx.p();
        if (length_squared(wi) == 0.0f) return false;
        wi = normalize(wi);
        ss.pdf /= abs_dot(ss.n, -wi) / distance_squared(ctx.p(), ss.pi.mid());
        if (is_inf(ss.pdf)) return false;
        out = ss;
        return true;
    }
    Float sin_theta_max = s.radius / distance(ctx.p(), p_center);
    Float sin2_theta_max = sqr(sin_theta_max);
    Float cos_theta_max = safe_sqrt(1.0f - sin2_theta_max);
    Float one_minus_cos_theta_max = 1.0f - cos_theta_max;
    Float cos_theta_ = (cos_theta_max - 1.0f) * u.x + 1.0f;
    Float sin2_theta_ = 1.0f - sqr(cos_theta_);
    if (sin2_theta_max < 0.00068523f) {
        sin2_theta_ = sin2_theta_max * u.x;
        cos_theta_ = sqrt(1.0f - sin2_theta_);
        one_minus_cos_theta_max = sin2_theta_max / 2.0f;
    }
    Float cos_alpha = sin2_theta_ / sin_theta_max + cos_theta_ * safe_sqrt(1.0f - sin2_theta_ / sqr(sin_theta_max));
    Float sin_alpha = safe_sqrt(1.0f - sqr(cos_alpha));
    Float phi = u.y * 2.0f * PI_F;
    V3 w = spherical_direction(sin_alpha, cos_alpha, phi);
    Frame sampling_frame = frame_from_z(normalize(p_center - ctx.p()));
    Float normal_sign = s.reverse_orientation ? -1.0f : 1.0f;
    V3 n = normal_sign * sampling_frame.from_local(-w);
    V3 p = p_center + v3(n.x, n.y, n.z) * s.radius;
    V3 p_error = gamma(5) * abs3(p);
    out.pi = p3i_from_value_and_error(p, p_error);
    out.n = n;
    out.pdf = 1.0f / (2.0f * PI_F * one_minus_cos_theta_max);
    return true;
}
// sphere.rs:424-457 (quirk 1: 2.90 and the double division are the reference's)
SHM_HD Float sphere_pdf_with_context(const ShmSphere& s, const ShapeSampleContext& ctx, V3 wi, bool strict = false) {
    V3 p_center = xf_point(s.render_from_object, v3s(0.0f));
    V3 p_origin = offset_ray_origin(ctx.pi, ctx.n, p_center - ctx.p());
    if (distance_squared(p_origin, p_center) <= s.radius * s.radius) {
        V3 o = offset_ray_origin(ctx.pi, ctx.n, wi);
        QuadricIntersection qi;
        if (!sphere_basic_intersect(s, o, wi, infinity(), qi)) return 0.0f;
        SurfaceInteraction isect = sphere_interaction(s, qi, -wi);
        Float pdf = strict ? (1.0f / sphere_area(s)) / (abs_dot(isect.n, -wi) / distance_squared(ctx.p(), isect.p()))
                           : (1.0f / sphere_area(s)) / abs_dot(isect.n, -wi) / distance_squared(ctx.p(), isect.p());
        if (is_inf(pdf)) return 0.0f;
        return pdf;
    }
    Float sin2_theta_max = s.radius * s.radius / distance_squared(ctx.p(), p_center);
    Float cos_theta_max = safe_sqrt(1.0f - sin2_theta_max);
    Float one_minus_cos_theta_max = 1.0f - cos_theta_max;
    if (sin2_theta_max < 0.00068523f) one_minus_cos_theta_max = sin2_theta_max / 2.0f;
    return 1.0f / ((strict ? 2.0f : 2.90f) * PI_F * one_minus_cos_theta_max);
}

}  // namespace shm
