// pbrt_math.hpp — the f32 transform algebra of the reference's scene front end, for the C++ PBRT-v4 loader (host/pbrt_loader.cpp) and
// shm_look_at. Restates (paths relative to /root/reference/src):
//   transform.rs:84-107, 109-124, 225-252, 270-303   Transform::{translate, scale, rotate, look_at}
//   transform.rs:356-361                               Transform * Transform (m = a.m b.m, m_inv = b.m_inv a.m_inv)
//   transform.rs:753-786                               apply_{point,vector,normal}_helper
//   square_matrix.rs:254-266, 353-466                  4x4 product and inverse (Laplace expansion with shared 2x2 minors)
// The reference's inner_product is a compensated (TwoProd / TwoSum) sum; here the products are formed exactly in f64 and summed there:
// the f32 result is the correctly rounded sum up to double-rounding corner cases. Host-side only; nothing here runs on the device.
#pragma once
#include <cmath>
#include <cstring>

#include "../shm/vec.h"

namespace pbrt {

using shm::V3;

struct M4 {
    float m[4][4];
};
inline M4 m4_identity() {
    M4 r;
    memset(&r, 0, sizeof(r));
    for (int i = 0; i < 4; ++i) r.m[i][i] = 1.0f;
    return r;
}
inline float inner(const float* a, const float* b, int n) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += (double)a[i] * (double)b[i];
    return (float)s;
}
inline M4 m4_mul(const M4& a, const M4& b) {
    M4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            const float row[4] = {a.m[i][0], a.m[i][1], a.m[i][2], a.m[i][3]};
            const float col[4] = {b.m[0][j], b.m[1][j], b.m[2][j], b.m[3][j]};
            r.m[i][j] = inner(row, col, 4);
        }
    return r;
}
inline M4 m4_transpose(const M4& a) {
    M4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) r.m[i][j] = a.m[j][i];
    return r;
}
inline bool m4_inverse(const M4& a, M4& out) {  // square_matrix.rs:353-466
    auto dop = [](float p, float q, float r, float s) { return shm::difference_of_products(p, q, r, s); };
    const auto& m = a.m;
    const float s0 = dop(m[0][0], m[1][1], m[1][0], m[0][1]), s1 = dop(m[0][0], m[1][2], m[1][0], m[0][2]), s2 = dop(m[0][0], m[1][3], m[1][0], m[0][3]);
    const float s3 = dop(m[0][1], m[1][2], m[1][1], m[0][2]), s4 = dop(m[0][1], m[1][3], m[1][1], m[0][3]), s5 = dop(m[0][2], m[1][3], m[1][2], m[0][3]);
    const float c0 = dop(m[2][0], m[3][1], m[3][0], m[2][1]), c1 = dop(m[2][0], m[3][2], m[3][0], m[2][2]), c2 = dop(m[2][0], m[3][3], m[3][0], m[2][3]);
    const float c3 = dop(m[2][1], m[3][2], m[3][1], m[2][2]), c4 = dop(m[2][1], m[3][3], m[3][1], m[2][3]), c5 = dop(m[2][2], m[3][3], m[3][2], m[2][3]);
    const float da[6] = {s0, -s1, s2, s3, s5, -s4}, db[6] = {c5, c4, c3, c2, c0, c1};
    const float det = inner(da, db, 6);
    if (det == 0.0f) return false;
    const float s = 1.0f / det;
    auto ip3 = [](float a0, float a1, float a2, float b0, float b1, float b2) { const float x[3] = {a0, a1, a2}, y[3] = {b0, b1, b2}; return inner(x, y, 3); };
    out.m[0][0] = s * ip3(m[1][1], m[1][3], -m[1][2], c5, c3, c4);
    out.m[0][1] = s * ip3(-m[0][1], m[0][2], -m[0][3], c5, c4, c3);
    out.m[0][2] = s * ip3(m[3][1], m[3][3], -m[3][2], s5, s3, s4);
    out.m[0][3] = s * ip3(-m[2][1], m[2][2], -m[2][3], s5, s4, s3);
    out.m[1][0] = s * ip3(-m[1][0], m[1][2], -m[1][3], c5, c2, c1);
    out.m[1][1] = s * ip3(m[0][0], m[0][3], -m[0][2], c5, c1, c2);
    out.m[1][2] = s * ip3(-m[3][0], m[3][2], -m[3][3], s5, s2, s1);
    out.m[1][3] = s * ip3(m[2][0], m[2][3], -m[2][2], s5, s1, s2);
    out.m[2][0] = s * ip3(m[1][0], m[1][3], -m[1][1], c4, c0, c2);
    out.m[2][1] = s * ip3(-m[0][0], m[0][1], -m[0][3], c4, c2, c0);
    out.m[2][2] = s * ip3(m[3][0], m[3][3], -m[3][1], s4, s0, s2);
    out.m[2][3] = s * ip3(-m[2][0], m[2][1], -m[2][3], s4, s2, s0);
    out.m[3][0] = s * ip3(-m[1][0], m[1][1], -m[1][2], c3, c1, c0);
    out.m[3][1] = s * ip3(m[0][0], m[0][2], -m[0][1], c3, c0, c1);
    out.m[3][2] = s * ip3(-m[3][0], m[3][1], -m[3][2], s3, s1, s0);
    out.m[3][3] = s * ip3(m[2][0], m[2][2], -m[2][1], s3, s0, s1);
    return true;
}

struct Xf {  // Transform {m, m_inv}
    M4 m, inv;
};
inline Xf xf_identity() { return Xf{m4_identity(), m4_identity()}; }
inline Xf xf_mul(const Xf& a, const Xf& b) { return Xf{m4_mul(a.m, b.m), m4_mul(b.inv, a.inv)}; }
inline Xf xf_inverse(const Xf& a) { return Xf{a.inv, a.m}; }
inline Xf xf_translate(float x, float y, float z) {
    Xf r = xf_identity();
    r.m.m[0][3] = x; r.m.m[1][3] = y; r.m.m[2][3] = z;
    r.inv.m[0][3] = -x; r.inv.m[1][3] = -y; r.inv.m[2][3] = -z;
    return r;
}
inline Xf xf_scale(float x, float y, float z) {
    Xf r = xf_identity();
    r.m.m[0][0] = x; r.m.m[1][1] = y; r.m.m[2][2] = z;
    r.inv.m[0][0] = 1.0f / x; r.inv.m[1][1] = 1.0f / y; r.inv.m[2][2] = 1.0f / z;
    return r;
}
inline Xf xf_rotate(float theta_deg, V3 axis) {  // transform.rs:199-252 (sin / cos of the platform libm, as Rust's f32::sin / cos)
    const float theta = theta_deg * (3.14159265358979323846f / 180.0f);
    const float st = sinf(theta), ct = cosf(theta);
    const V3 a = shm::normalize(axis);
    Xf r = xf_identity();
    float(*m)[4] = r.m.m;
    m[0][0] = a.x * a.x + (1.0f - a.x * a.x) * ct;
    m[0][1] = a.x * a.y * (1.0f - ct) - a.z * st;
    m[0][2] = a.x * a.z * (1.0f - ct) + a.y * st;
    m[1][0] = a.x * a.y * (1.0f - ct) + a.z * st;
    m[1][1] = a.y * a.y + (1.0f - a.y * a.y) * ct;
    m[1][2] = a.y * a.z * (1.0f - ct) - a.x * st;
    m[2][0] = a.x * a.z * (1.0f - ct) - a.y * st;
    m[2][1] = a.y * a.z * (1.0f - ct) + a.x * st;
    m[2][2] = a.z * a.z + (1.0f - a.z * a.z) * ct;
    r.inv = m4_transpose(r.m);
    return r;
}
// transform.rs:270-303: world_from_camera from (pos, look_at, up) exactly as written there; returns false for a degenerate frame
inline bool look_at_world_from_camera(V3 pos, V3 look, V3 up, M4& wfc) {
    memset(&wfc, 0, sizeof(wfc));
    wfc.m[0][3] = pos.x; wfc.m[1][3] = pos.y; wfc.m[2][3] = pos.z; wfc.m[3][3] = 1.0f;
    const V3 d = look - pos;
    if (shm::length_squared(d) == 0.0f || shm::length_squared(up) == 0.0f) return false;
    const V3 dir = shm::normalize(d);
    const V3 c = shm::cross(shm::normalize(up), dir);
    if (shm::length_squared(c) == 0.0f) return false;
    const V3 right = shm::normalize(c);
    const V3 new_up = shm::cross(dir, right);
    wfc.m[0][0] = right.x; wfc.m[1][0] = right.y; wfc.m[2][0] = right.z;
    wfc.m[0][1] = new_up.x; wfc.m[1][1] = new_up.y; wfc.m[2][1] = new_up.z;
    wfc.m[0][2] = dir.x; wfc.m[1][2] = dir.y; wfc.m[2][2] = dir.z;
    return true;
}
inline bool xf_look_at(V3 pos, V3 look, V3 up, Xf& out) {
    M4 wfc;
    if (!look_at_world_from_camera(pos, look, up, wfc)) return false;
    out.inv = wfc;
    return m4_inverse(wfc, out.m);
}
inline V3 xf_point(const M4& m, V3 p) {  // transform.rs:753-767
    const float xp = m.m[0][0] * p.x + m.m[0][1] * p.y + m.m[0][2] * p.z + m.m[0][3];
    const float yp = m.m[1][0] * p.x + m.m[1][1] * p.y + m.m[1][2] * p.z + m.m[1][3];
    const float zp = m.m[2][0] * p.x + m.m[2][1] * p.y + m.m[2][2] * p.z + m.m[2][3];
    const float wp = m.m[3][0] * p.x + m.m[3][1] * p.y + m.m[3][2] * p.z + m.m[3][3];
    if (wp == 1.0f) return shm::v3(xp, yp, zp);
    return shm::v3(xp / wp, yp / wp, zp / wp);
}
inline V3 xf_vector(const M4& m, V3 v) {
    return shm::v3(m.m[0][0] * v.x + m.m[0][1] * v.y + m.m[0][2] * v.z, m.m[1][0] * v.x + m.m[1][1] * v.y + m.m[1][2] * v.z,
                   m.m[2][0] * v.x + m.m[2][1] * v.y + m.m[2][2] * v.z);
}
inline V3 xf_normal(const M4& m_inv, V3 n) {  // transposed inverse
    return shm::v3(m_inv.m[0][0] * n.x + m_inv.m[1][0] * n.y + m_inv.m[2][0] * n.z, m_inv.m[0][1] * n.x + m_inv.m[1][1] * n.y + m_inv.m[2][1] * n.z,
                   m_inv.m[0][2] * n.x + m_inv.m[1][2] * n.y + m_inv.m[2][2] * n.z);
}
inline bool swaps_handedness(const M4& m) {  // transform.rs:329-337 (3x3 determinant, square_matrix.rs:282-293)
    const float minor12 = shm::difference_of_products(m.m[1][1], m.m[2][2], m.m[1][2], m.m[2][1]);
    const float minor02 = shm::difference_of_products(m.m[1][0], m.m[2][2], m.m[1][2], m.m[2][0]);
    const float minor01 = shm::difference_of_products(m.m[1][0], m.m[2][1], m.m[1][1], m.m[2][0]);
    const float det = std::fma(m.m[0][2], minor01, shm::difference_of_products(m.m[0][0], minor12, m.m[0][1], minor02));
    return det < 0.0f;
}

}  // namespace pbrt
