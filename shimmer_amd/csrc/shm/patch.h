// shimmer-hip — BilinearPatch, single-source for the HIP kernels and the CPU oracle.
// Restates (reference file:line under /root/reference/src):
//   shape/bilinear_patch.rs:38-75    BilinearPatch::new (area: exact for rectangles, 3x3 tessellation otherwise)
//   shape/bilinear_patch.rs:108-142  is_rectangle
//   shape/bilinear_patch.rs:144-236  intersect_blp (u from a quadratic, v and t from 3x3 determinants)
//   shape/bilinear_patch.rs:238-428  interaction_from_intersection (meshes without per-vertex uv / n: see below)
//   shape/bilinear_patch.rs:521-636  sample, pdf
//   shape/bilinear_patch.rs:638-783  sample_with_context, pdf_with_context
//   math.rs:377-410                  quadratic
//   square_matrix.rs:281-292         determinant of a 3x3 matrix
//   vecmath/mod.rs:118-140           spherical_quad_area
//   sampling.rs:501-579, 645-787     sample_spherical_rectangle, invert_spherical_rectangle_sample
//   transform.rs:227-253             Transform::rotate_from_to (shading frame of a patch with per-vertex normals)
//   vecmath/mod.rs:70-116            invert_bilinear
// Per-vertex `uv` (the (s,t) re-parameterisation, :258-318) and `n` (shading normals, :399-424; normal flipping in the
// samplers) are both carried. What is NOT: image-valued emission on a patch light (the `st` of a sample is therefore not
// produced: nothing on the path reads it).
// Reference behaviour kept as written: sample() interpolates pu0 / pu1 along different parameters (:549-553), pdf() uses
// uv[1] for both edge points (:627-628); both differ from PBRT-v4 and both are what the reference computes.
#pragma once
#include "shapes.h"

namespace shm {

struct PatchData {
    V3 p00, p10, p01, p11;
    bool flip;     // reverse_orientation ^ transform_swaps_handedness
    bool is_rect;  // is_rectangle(), evaluated once at scene creation with blp_is_rectangle
    Float area;    // BilinearPatch::new
    bool has_n, has_uv;
    V3 n00, n10, n01, n11;
    V2 uv00, uv10, uv01, uv11;
};
SHM_HD V2 lerp2(Float t, V2 a, V2 b) { return a * (1.0f - t) + b * t; }
// the interpolated per-vertex normal lerp(u, lerp(v, n00, n01), lerp(v, n10, n11)) of :407, :576, :714
SHM_HD V3 blp_interp_normal(const PatchData& pd, Float u, Float v);
struct BilinearIntersection {
    Float u, v, t;
};

SHM_HD V3 lerp3(Float t, V3 a, V3 b) { return a * (1.0f - t) + b * t; }  // math.rs:246-252 for vectors

// math.rs:377-410
SHM_HD bool quadratic(Float a, Float b, Float c, Float& t0, Float& t1) {
    if (a == 0.0f) {
        if (b == 0.0f) return false;
        t0 = t1 = -c / b;
        return true;
    }
    Float discrim = difference_of_products(b, b, 4.0f * a, c);
    if (discrim < 0.0f) return false;
    Float root_discrim = sqrt(discrim);
    Float q = -0.5f * (b + copysign(root_discrim, b));
    t0 = q / a;
    t1 = c / q;
    if (t0 > t1) { Float tmp = t0; t0 = t1; t1 = tmp; }
    return true;
}
// square_matrix.rs:281-292, rows r0, r1, r2
SHM_HD Float det3(V3 r0, V3 r1, V3 r2) {
    Float minor12 = difference_of_products(r1.y, r2.z, r1.z, r2.y);
    Float minor02 = difference_of_products(r1.x, r2.z, r1.z, r2.x);
    Float minor01 = difference_of_products(r1.x, r2.y, r1.y, r2.x);
    return fma(r0.z, minor01, difference_of_products(r0.x, minor12, r0.y, minor02));
}

// bilinear_patch.rs:108-142
SHM_HD bool blp_is_rectangle(V3 p00, V3 p10, V3 p01, V3 p11) {
    if (p00 == p01 || p01 == p11 || p11 == p10 || p10 == p00) return false;
    V3 n = normalize(cross(p10 - p00, p01 - p00));
    if (abs_dot(normalize(p11 - p00), n) > 1e-5f) return false;
    V3 p_center = (p00 + p01 + p10 + p11) * 0.25f;
    Float d2[4] = {length_squared(p00 - p_center), length_squared(p01 - p_center), length_squared(p10 - p_center),
                   length_squared(p11 - p_center)};
    for (int i = 1; i < 4; ++i)
        if (abs(d2[i] - d2[0]) / d2[0] > 1e-4f) return false;
    return true;
}
// bilinear_patch.rs:40-69
SHM_HD Float blp_area(V3 p00, V3 p10, V3 p01, V3 p11, bool is_rect) {
    if (is_rect) return distance(p00, p01) * distance(p00, p10);
    constexpr int NA = 3;
    V3 p[NA + 1][NA + 1];
    for (int i = 0; i <= NA; ++i) {
        Float u = (Float)i / (Float)NA;
        for (int j = 0; j <= NA; ++j) {
            Float v = (Float)j / (Float)NA;
            p[i][j] = lerp3(u, lerp3(v, p00, p01), lerp3(v, p10, p11));
        }
    }
    Float area = 0.0f;
    for (int i = 0; i < NA; ++i)
        for (int j = 0; j < NA; ++j) area += 0.5f * length(cross(p[i + 1][j + 1] - p[i][j], p[i + 1][j] - p[i][j + 1]));
    return area;
}

// bilinear_patch.rs:144-236
SHM_HD bool blp_intersect(V3 ro, V3 rd, Float t_max, V3 p00, V3 p10, V3 p01, V3 p11, BilinearIntersection& out) {
    Float a = dot(cross(p10 - p00, p01 - p11), rd);
    Float c = dot(cross(p00 - ro, rd), p01 - p00);
    Float b = dot(cross(p10 - ro, rd), p11 - p10) - (a + c);
    Float u1, u2;
    if (!quadratic(a, b, c, u1, u2)) return false;
    Float eps = gamma(10) * (max_component_value(abs3(ro)) + max_component_value(abs3(rd)) + max_component_value(abs3(p00)) +
                             max_component_value(abs3(p10)) + max_component_value(abs3(p01)) + max_component_value(abs3(p11)));
    Float t = t_max, u = 0.0f, v = 0.0f;
    if (0.0f <= u1 && u1 <= 1.0f) {
        V3 uo = lerp3(u1, p00, p10);
        V3 ud = lerp3(u1, p01, p11) - uo;
        V3 deltao = uo - ro;
        V3 perp = cross(rd, ud);
        Float p2 = length_squared(perp);
        Float v1 = det3(v3(deltao.x, rd.x, perp.x), v3(deltao.y, rd.y, perp.y), v3(deltao.z, rd.z, perp.z));
        Float t1 = det3(v3(deltao.x, ud.x, perp.x), v3(deltao.y, ud.y, perp.y), v3(deltao.z, ud.z, perp.z));
        if (t1 > p2 * eps && 0.0f <= v1 && v1 <= p2) {
            u = u1;
            v = v1 / p2;
            t = t1 / p2;
        }
    }
    if (0.0f <= u2 && u2 <= 1.0f && u2 != u1) {
        V3 uo = lerp3(u2, p00, p10);
        V3 ud = lerp3(u2, p01, p11) - uo;
        V3 deltao = uo - ro;
        V3 perp = cross(rd, ud);
        Float p2 = length_squared(perp);
        Float v2 = det3(v3(deltao.x, rd.x, perp.x), v3(deltao.y, rd.y, perp.y), v3(deltao.z, rd.z, perp.z));
        Float t2 = det3(v3(deltao.x, ud.x, perp.x), v3(deltao.y, ud.y, perp.y), v3(deltao.z, ud.z, perp.z));
        t2 /= p2;
        if (0.0f <= v2 && v2 <= p2 && t > t2 && t2 > eps) {
            t = t2;
            u = u2;
            v = v2 / p2;
        }
    }
    if (t >= t_max) return false;
    out.u = u; out.v = v; out.t = t;
    return true;
}

SHM_HD V3 blp_interp_normal(const PatchData& pd, Float u, Float v) { return lerp3(u, lerp3(v, pd.n00, pd.n01), lerp3(v, pd.n10, pd.n11)); }

// transform.rs:227-253, the 3x3 part (row-major); applied to vectors as Transform::apply does (transform.rs:420-431)
struct Rot3 { Float m[3][3]; };
SHM_HD Rot3 rotate_from_to(V3 from, V3 to) {
    V3 ref1 = (abs(from.x) < 0.72f && abs(to.x) < 0.72f) ? v3(1.0f, 0.0f, 0.0f)
              : ((abs(from.y) < 0.72f && abs(to.y) < 0.72f) ? v3(0.0f, 1.0f, 0.0f) : v3(0.0f, 0.0f, 1.0f));
    V3 u = ref1 - from, v = ref1 - to;
    const Float ua[3] = {u.x, u.y, u.z}, va[3] = {v.x, v.y, v.z};
    Rot3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            Float kronecker = (i == j) ? 1.0f : 0.0f;
            r.m[i][j] = kronecker - 2.0f / dot(u, u) * ua[i] * ua[j] - 2.0f / dot(v, v) * va[i] * va[j]
                        + 4.0f * dot(u, v) / (dot(u, u) * dot(v, v)) * va[i] * ua[j];
        }
    return r;
}
SHM_HD V3 rot3_apply(const Rot3& r, V3 a) {
    return v3(r.m[0][0] * a.x + r.m[0][1] * a.y + r.m[0][2] * a.z, r.m[1][0] * a.x + r.m[1][1] * a.y + r.m[1][2] * a.z,
              r.m[2][0] * a.x + r.m[2][1] * a.y + r.m[2][2] * a.z);
}

// bilinear_patch.rs:238-428
SHM_HD SurfaceInteraction blp_interaction(const PatchData& pd, Float u, Float v, V3 wo) {
    V3 p = lerp3(u, lerp3(v, pd.p00, pd.p01), lerp3(v, pd.p10, pd.p11));
    V3 dpdu = lerp3(v, pd.p10, pd.p11) - lerp3(v, pd.p00, pd.p01);
    V3 dpdv = lerp3(u, pd.p01, pd.p11) - lerp3(u, pd.p00, pd.p10);
    V2 st = v2(u, v);
    Float duds = 1.0f, dudt = 0.0f, dvds = 0.0f, dvdt = 1.0f;
    if (pd.has_uv) {
        st = lerp2(u, lerp2(v, pd.uv00, pd.uv01), lerp2(v, pd.uv10, pd.uv11));
        V2 dstdu = lerp2(v, pd.uv10, pd.uv11) - lerp2(v, pd.uv00, pd.uv01);
        V2 dstdv = lerp2(u, pd.uv01, pd.uv11) - lerp2(u, pd.uv00, pd.uv10);
        duds = (abs(dstdu.x) < 1e-8f) ? 0.0f : 1.0f / dstdu.x;
        dvds = (abs(dstdv.x) < 1e-8f) ? 0.0f : 1.0f / dstdv.x;
        dudt = (abs(dstdu.y) < 1e-8f) ? 0.0f : 1.0f / dstdu.y;
        dvdt = (abs(dstdv.y) < 1e-8f) ? 0.0f : 1.0f / dstdv.y;
        V3 dpds = dpdu * duds + dpdv * dvds;
        V3 dpdt = dpdu * dudt + dpdv * dvdt;
        if (cross(dpds, dpdt) != v3s(0.0f)) {
            if (dot(cross(dpdu, dpdv), cross(dpds, dpdt)) < 0.0f) dpdt = -dpdt;
            dpdu = dpds;
            dpdv = dpdt;
        }
    }
    V3 d2pduu = v3s(0.0f), d2pdvv = v3s(0.0f);
    V3 d2pduv = (pd.p00 - pd.p01) + (pd.p11 - pd.p10);
    Float e1 = dot(dpdu, dpdu), f1 = dot(dpdu, dpdv), g1 = dot(dpdv, dpdv);
    V3 n = normalize(cross(dpdu, dpdv));
    Float e2 = dot(n, d2pduu), f2 = dot(n, d2pduv), g2 = dot(n, d2pdvv);
    Float egf2 = difference_of_products(e1, g1, f1, f1);
    Float inv_egf2 = (egf2 != 0.0f) ? 1.0f / egf2 : 0.0f;
    V3 dndu = (f1 * f2 - e2 * g1) * inv_egf2 * dpdu + (e2 * f1 - f2 * e1) * inv_egf2 * dpdv;
    V3 dndv = (g2 * f1 - f2 * g1) * inv_egf2 * dpdu + (f2 * f1 - g2 * e1) * inv_egf2 * dpdv;
    V3 dnds = dndu * duds + dndv * dvds;
    V3 dndt = dndu * dudt + dndv * dvdt;
    dndu = dnds;
    dndv = dndt;
    V3 p_abs_sum = abs3(pd.p00) + abs3(pd.p01) + abs3(pd.p10) + abs3(pd.p11);
    V3 p_error = gamma(6) * p_abs_sum;
    SurfaceInteraction isect = surface_interaction_new(p3i_from_value_and_error(p, p_error), st, wo, dpdu, dpdv, dndu, dndv, pd.flip);
    if (pd.has_n) {  // :399-424
        V3 ns = blp_interp_normal(pd, u, v);
        if (length_squared(ns) > 0.0f) {
            ns = normalize(ns);
            V3 dndu_s = lerp3(v, pd.n10, pd.n11) - lerp3(v, pd.n00, pd.n01);
            V3 dndv_s = lerp3(u, pd.n01, pd.n11) - lerp3(u, pd.n00, pd.n10);
            V3 dnds_s = dndu_s * duds + dndv_s * dvds;
            V3 dndt_s = dndu_s * dudt + dndv_s * dvdt;
            Rot3 r = rotate_from_to(isect.n, ns);
            set_shading_geometry(isect, ns, rot3_apply(r, dpdu), rot3_apply(r, dpdv), dnds_s, dndt_s, true);
        }
    }
    return isect;
}

// vecmath/mod.rs:70-116
SHM_HD Float cross2d(V2 a, V2 b) { return difference_of_products(a.x, b.y, a.y, b.x); }
SHM_HD V2 invert_bilinear(V2 p, V2 v0, V2 v1, V2 v2_, V2 v3_) {
    V2 a = v0, b = v1, c = v3_, d = v2_;
    V2 e = b - a, f = d - a, g = (a - b) + (c - d), h = p - a;
    Float k2 = cross2d(g, f);
    Float k1 = cross2d(e, f) + cross2d(h, g);
    Float k0 = cross2d(h, e);
    if (abs(k2) < 0.001f) {
        if (abs(e.x * k1 - g.x * k0) < 1e-5f) return v2((h.y * k1 + f.y * k0) / (e.y * k1 - g.y * k0), -k0 / k1);
        return v2((h.x * k1 + f.x * k0) / (e.x * k1 - g.x * k0), -k0 / k1);
    }
    Float r0, r1;
    if (!quadratic(k2, k1, k0, r0, r1)) return v2(0.0f, 0.0f);
    Float u = (h.x - f.x * r0) / (e.x + g.x * r0);
    if (u < 0.0f || u > 1.0f || r0 < 0.0f || r0 > 1.0f) return v2((h.x - f.x * r1) / (e.x + g.x * r1), r1);
    return v2(u, r0);
}

// vecmath/mod.rs:118-140
SHM_HD Float spherical_quad_area(V3 a, V3 b, V3 c, V3 d) {
    V3 axb = cross(a, b), bxc = cross(b, c), cxd = cross(c, d), dxa = cross(d, a);
    if (length_squared(axb) == 0.0f || length_squared(bxc) == 0.0f || length_squared(cxd) == 0.0f || length_squared(dxa) == 0.0f)
        return 0.0f;
    axb = normalize(axb); bxc = normalize(bxc); cxd = normalize(cxd); dxa = normalize(dxa);
    Float alpha = angle_between(dxa, -axb);
    Float beta = angle_between(axb, -bxc);
    Float gamma_ = angle_between(bxc, -cxd);
    Float delta = angle_between(cxd, -dxa);
    return abs(alpha + beta + gamma_ + delta - 2.0f * PI_F);
}

// sampling.rs:501-579
SHM_HD V3 sample_spherical_rectangle(V3 p_ref, V3 s, V3 ex, V3 ey, V2 u, Float& pdf) {
    Float exl = length(ex), eyl = length(ey);
    Frame r;
    r.x = ex / exl;
    r.y = ey / eyl;
    r.z = cross(r.x, r.y);  // Frame::from_xy, frame.rs:19-22
    V3 d_local = r.to_local(s - p_ref);
    Float z0 = d_local.z;
    if (z0 > 0.0f) {
        r.z = -r.z;
        z0 *= -1.0f;
    }
    Float x0 = d_local.x, y0 = d_local.y;
    Float x1 = x0 + exl, y1 = y0 + eyl;
    V3 v00 = v3(x0, y0, z0), v01 = v3(x0, y1, z0), v10 = v3(x1, y0, z0), v11 = v3(x1, y1, z0);
    V3 n0 = normalize(cross(v00, v10)), n1 = normalize(cross(v10, v11)), n2 = normalize(cross(v11, v01)), n3 = normalize(cross(v01, v00));
    Float g0 = angle_between(-n0, n1), g1 = angle_between(-n1, n2), g2 = angle_between(-n2, n3), g3 = angle_between(-n3, n0);
    Float solid_angle = g0 + g1 + g2 + g3 - 2.0f * PI_F;
    if (solid_angle <= 0.0f) {
        pdf = 0.0f;
        return s + u.x * ex + u.y * ey;
    }
    pdf = max(0.0f, 1.0f / solid_angle);
    if (solid_angle < 1e-3f) return s + u.x * ex + u.y * ey;
    Float b0 = n0.z, b1 = n2.z;
    Float au = u.x * (g0 + g1 - 2.0f * PI_F) + (u.x - 1.0f) * (g2 + g3);
    Float fu = (cos(au) * b0 - b1) / sin(au);
    Float cu = copysign(1.0f / sqrt(sqr(fu) + sqr(b0)), fu);
    cu = clamp(cu, -(1.0f - 1.1920929e-7f), 1.0f - 1.1920929e-7f);  // Float::EPSILON
    Float xu = -(cu * z0) / safe_sqrt(1.0f - sqr(cu));
    xu = clamp(xu, x0, x1);
    Float dd = sqrt(sqr(xu) + sqr(z0));
    Float h0 = y0 / sqrt(sqr(dd) + sqr(y0));
    Float h1 = y1 / sqrt(sqr(dd) + sqr(y1));
    Float hv = h0 + u.y * (h1 - h0);
    Float hvsq = sqr(hv);
    Float yv = (hvsq < 1.0f - 1e-6f) ? (hv * dd) / sqrt(1.0f - hvsq) : y1;
    return p_ref + r.from_local(v3(xu, yv, z0));
}

// sampling.rs:645-787
SHM_HD V2 invert_spherical_rectangle_sample(V3 p_ref, V3 s, V3 ex, V3 ey, V3 p_rect) {
    Float exl = length(ex), eyl = length(ey);
    Frame r;
    r.x = ex / exl;
    r.y = ey / eyl;
    r.z = cross(r.x, r.y);
    V3 d_local = r.to_local(s - p_ref);
    Float z0 = d_local.z;
    if (z0 > 0.0f) {
        r.z = -r.z;
        z0 *= -1.0f;
    }
    Float z0sq = sqr(z0);
    Float x0 = d_local.x, y0 = d_local.y;
    Float x1 = x0 + exl, y1 = y0 + eyl;
    Float y0sq = sqr(y0), y1sq = sqr(y1);
    V3 v00 = v3(x0, y0, z0), v01 = v3(x0, y1, z0), v10 = v3(x1, y0, z0), v11 = v3(x1, y1, z0);
    V3 n0 = normalize(cross(v00, v10)), n1 = normalize(cross(v10, v11)), n2 = normalize(cross(v11, v01)), n3 = normalize(cross(v01, v00));
    Float g0 = angle_between(-n0, n1), g1 = angle_between(-n1, n2), g2 = angle_between(-n2, n3), g3 = angle_between(-n3, n0);
    Float b0 = n0.z, b1 = n2.z;
    Float b0sq = sqr(b0);
    Float solid_angle = g0 + g1 + g2 + g3 - 2.0f * PI_F;
    if (solid_angle < 1e-3f) {
        V3 pq = p_rect - s;
        return v2(dot(pq, ex) / length_squared(ex), dot(pq, ey) / length_squared(ey));
    }
    V3 vl = r.to_local(p_rect - p_ref);
    Float xu = vl.x, yv = vl.y;
    xu = clamp(xu, x0, x1);
    if (xu == 0.0f) xu = 1e-10f;
    Float invcusq = 1.0f + z0sq / sqr(xu);
    Float fusq = invcusq - b0sq;
    Float fu = copysign(sqrt(fusq), xu);
    Float sq = safe_sqrt(difference_of_products(b0, b0, b1, b1) + fusq);
    Float au = atan2(-(b1 * fu) - copysign(b0 * sq, fu * b0), b0 * b1 - sq * abs(fu));
    if (au > 0.0f) au -= 2.0f * PI_F;
    if (fu == 0.0f) au = PI_F;
    Float u0 = (au + g2 + g3) / solid_angle;
    Float ddsq = sqr(xu) + z0sq;
    Float dd = sqrt(ddsq);
    Float h0 = y0 / sqrt(ddsq + y0sq);
    Float h1 = y1 / sqrt(ddsq + y1sq);
    Float yvsq = sqr(yv);
    Float u1a = (difference_of_products(h0, h0, h0, h1) - abs(h0 - h1) * sqrt(yvsq * (ddsq + yvsq)) / (ddsq + yvsq)) / sqr(h0 - h1);
    Float u1b = (difference_of_products(h0, h0, h0, h1) + abs(h0 - h1) * sqrt(yvsq * (ddsq + yvsq)) / (ddsq + yvsq)) / sqr(h0 - h1);
    Float hva = lerp(u1a, h0, h1), hvb = lerp(u1b, h0, h1);
    Float yza = (hva * dd) / sqrt(1.0f - sqr(hva)), yzb = (hvb * dd) / sqrt(1.0f - sqr(hvb));
    if (abs(yza - yv) < abs(yzb - yv)) return v2(clamp(u0, 0.0f, 1.0f), u1a);
    return v2(clamp(u0, 0.0f, 1.0f), u1b);
}

// bilinear_patch.rs:521-600 (the sample's st is not produced: see the header). `strict` (ShmRenderParams::disable_reference_quirks, round 6): PBRT-v4's edge points
// lerp(v, p00, p01) and lerp(v, p10, p11) here and in blp_pdf — with the reference's (sic, below) the point is not uniformly distributed and the density is not the point's:
// a non-rectangular patch emitter renders at 40-50 % of its light (tests/test_quirks_switch.py, against an estimator that samples no lights)
SHM_HD bool blp_sample(const PatchData& pd, V2 u, ShapeSample& out, bool strict = false) {
    V2 uv;
    Float pdf;
    if (pd.is_rect) {
        uv = u;
        pdf = 1.0f;
    } else {
        Float w[4] = {length(cross(pd.p10 - pd.p00, pd.p01 - pd.p00)), length(cross(pd.p10 - pd.p00, pd.p11 - pd.p10)),
                      length(cross(pd.p01 - pd.p00, pd.p11 - pd.p01)), length(cross(pd.p11 - pd.p10, pd.p11 - pd.p01))};
        uv = sample_bilinear(u, w);
        pdf = bilinear_pdf(uv, w);
    }
    V3 pu0 = strict ? lerp3(uv.y, pd.p00, pd.p01) : lerp3(uv.x, pd.p00, pd.p10);  // (sic, :549)
    V3 pu1 = lerp3(uv.y, pd.p10, pd.p11);  // (sic, :550: right only beside PBRT-v4's pu0)
    V3 p = lerp3(uv.x, pu0, pu1);
    V3 dpdu = pu1 - pu0;
    V3 dpdv = lerp3(uv.x, pd.p01, pd.p11) - lerp3(uv.x, pd.p00, pd.p10);
    if (length_squared(dpdu) == 0.0f || length_squared(dpdv) == 0.0f) return false;
    V3 n = normalize(cross(dpdu, dpdv));
    if (pd.has_n) n = face_forward(n, blp_interp_normal(pd, uv.x, uv.y));  // :570-577
    else if (pd.flip) n = -n;
    V3 p_abs_sum = abs3(pd.p00) + abs3(pd.p01) + abs3(pd.p10) + abs3(pd.p11);
    V3 p_error = gamma(6) * p_abs_sum;
    out.pi = p3i_from_value_and_error(p, p_error);
    out.n = n;
    out.pdf = pdf / length(cross(dpdu, dpdv));
    return true;
}
// bilinear_patch.rs:602-636: `uv` is Interaction::uv, i.e. (s, t) when the mesh has a uv array (inverted first, :607-614)
SHM_HD Float blp_pdf(const PatchData& pd, V2 uv, bool strict = false) {
    if (pd.has_uv) uv = invert_bilinear(uv, pd.uv00, pd.uv10, pd.uv01, pd.uv11);
    Float pdf = 1.0f;
    if (!pd.is_rect) {
        Float w[4] = {length(cross(pd.p10 - pd.p00, pd.p01 - pd.p00)), length(cross(pd.p10 - pd.p00, pd.p11 - pd.p10)),
                      length(cross(pd.p01 - pd.p00, pd.p11 - pd.p01)), length(cross(pd.p11 - pd.p10, pd.p11 - pd.p01))};
        pdf = bilinear_pdf(uv, w);
    }
    V3 pu0 = strict ? lerp3(uv.y, pd.p00, pd.p01) : lerp3(uv.y, pd.p00, pd.p10);  // (sic, :627)
    V3 pu1 = lerp3(uv.y, pd.p10, pd.p11);
    V3 dpdu = pu1 - pu0;
    V3 dpdv = lerp3(uv.x, pd.p01, pd.p11) - lerp3(uv.x, pd.p00, pd.p10);
    return pdf / length(cross(dpdu, dpdv));
}

constexpr Float BLP_MIN_SPHERICAL_SAMPLE_AREA = 1e-4f;

// bilinear_patch.rs:638-737
SHM_HD bool blp_sample_with_context(const PatchData& pd, const ShapeSampleContext& ctx, V2 u, ShapeSample& out, bool strict = false) {
    V3 rp = ctx.p();
    V3 v00 = normalize(pd.p00 - rp), v10 = normalize(pd.p10 - rp), v01 = normalize(pd.p01 - rp), v11 = normalize(pd.p11 - rp);
    if (!pd.is_rect || spherical_quad_area(v00, v10, v11, v01) <= BLP_MIN_SPHERICAL_SAMPLE_AREA) {
        ShapeSample ss;
        if (!blp_sample(pd, u, ss, strict)) return false;  // (the reference unwraps: a degenerate patch would panic there)
        V3 wi = ss.pi.mid() - rp;
        if (length_squared(wi) == 0.0f) return false;
        wi = normalize(wi);
        ss.pdf /= abs_dot(ss.n, -wi) / length_squared(rp - ss.pi.mid());
        if (is_inf(ss.pdf)) return false;
        out = ss;
        return true;
    }
    Float pdf = 1.0f;
    if (ctx.ns != v3s(0.0f)) {
        Float w[4] = {max(0.01f, dot(v00, ctx.ns)), max(0.01f, dot(v10, ctx.ns)), max(0.01f, dot(v01, ctx.ns)), max(0.01f, dot(v11, ctx.ns))};
        u = sample_bilinear(u, w);
        pdf = bilinear_pdf(u, w);
    }
    V3 eu = pd.p10 - pd.p00, ev = pd.p01 - pd.p00;
    Float quad_pdf = 0.0f;
    V3 p = sample_spherical_rectangle(rp, pd.p00, eu, ev, u, quad_pdf);
    pdf *= quad_pdf;
    V2 uv = v2(dot(p - pd.p00, eu) / distance_squared(pd.p10, pd.p00), dot(p - pd.p00, ev) / distance_squared(pd.p01, pd.p00));  // :698-701
    V3 n = normalize(cross(eu, ev));
    if (pd.has_n) n = face_forward(n, blp_interp_normal(pd, uv.x, uv.y));  // :706-716
    else if (pd.flip) n = -n;
    out.pi = p3i_exact(p);
    out.n = n;
    out.pdf = pdf;
    return true;
}

// bilinear_patch.rs:739-783
SHM_HD Float blp_pdf_with_context(const PatchData& pd, const ShapeSampleContext& ctx, V3 wi, bool strict = false) {
    V3 o = offset_ray_origin(ctx.pi, ctx.n, wi);  // ctx.spawn_ray(wi)
    BilinearIntersection bi;
    if (!blp_intersect(o, wi, infinity(), pd.p00, pd.p10, pd.p01, pd.p11, bi)) return 0.0f;
    SurfaceInteraction isect = blp_interaction(pd, bi.u, bi.v, -wi);
    V3 rp = ctx.p();
    V3 v00 = normalize(pd.p00 - rp), v10 = normalize(pd.p10 - rp), v01 = normalize(pd.p01 - rp), v11 = normalize(pd.p11 - rp);
    if (!pd.is_rect || spherical_quad_area(v00, v10, v11, v01) <= BLP_MIN_SPHERICAL_SAMPLE_AREA) {
        Float pdf = blp_pdf(pd, isect.uv, strict) * distance_squared(rp, isect.p()) / abs_dot(isect.n, -wi);
        return is_inf(pdf) ? 0.0f : pdf;
    }
    Float pdf = 1.0f / spherical_quad_area(v00, v10, v11, v01);
    if (ctx.ns != v3s(0.0f)) {
        Float w[4] = {max(0.01f, dot(v00, ctx.ns)), max(0.01f, dot(v10, ctx.ns)), max(0.01f, dot(v01, ctx.ns)), max(0.01f, dot(v11, ctx.ns))};
        V2 u = invert_spherical_rectangle_sample(rp, pd.p00, pd.p10 - pd.p00, pd.p01 - pd.p00, isect.p());
        return bilinear_pdf(u, w) * pdf;
    }
    return pdf;
}

}  // namespace shm
