// shm/vec.h — vector / point / normal / interval types of the Shimmer hot path.
//
// Restates (paths relative to /root/reference/src):
//   vecmath/tuple_fns.rs:40-78,162-213   cross (difference_of_products), dot3 (fma + sum_of_products),
//                                        angle_between, face_forward
//   vecmath/length_fns.rs:1-20           length_squared = x*x + y*y + z*z (no fma), length = sqrt
//   vecmath/normalize.rs:7-17            normalize = v / length (component-wise division)
//   vecmath/vector.rs:1034-1042,1144     coordinate_system (Duff et al.), gram_schmidt
//   vecmath/vector.rs:1195-1250          operator definitions (component-wise, v / s is a division)
//   vecmath/spherical.rs                 see sampling.h / scattering.h
//   interval.rs:19-420                   Interval (directed rounding via next_float_up/down)
//   vecmath/point.rs:1007-1026,1131-1190 Point3fi: from_value_and_error, error(), is_exact, +,-
//   frame.rs:1-60                        Frame::from_xz / from_z / to_local / from_local
//
// The reference distinguishes Vector3f / Point3f / Normal3f by type only; arithmetic is identical, so
// one V3 struct serves all three (the restating functions say which role an argument has).
#pragma once
#include "fp.h"

namespace shm {

struct V2 { Float x, y; };
SHM_HD V2 v2(Float x, Float y) { V2 r; r.x = x; r.y = y; return r; }

struct V3 {
    Float x, y, z;
    SHM_HD Float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
SHM_HD V3 v3(Float x, Float y, Float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
SHM_HD V3 v3s(Float s) { return v3(s, s, s); }
SHM_HD void v3_set(V3& v, int i, Float s) { if (i == 0) v.x = s; else if (i == 1) v.y = s; else v.z = s; }

SHM_HD V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
SHM_HD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
SHM_HD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
SHM_HD V3 operator*(V3 a, Float s) { return v3(a.x * s, a.y * s, a.z * s); }
SHM_HD V3 operator*(Float s, V3 a) { return v3(a.x * s, a.y * s, a.z * s); }
SHM_HD V3 operator/(V3 a, Float s) { return v3(a.x / s, a.y / s, a.z / s); }
SHM_HD V3 mul3(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
SHM_HD bool operator==(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
SHM_HD bool operator!=(V3 a, V3 b) { return !(a == b); }
SHM_HD V3 abs3(V3 a) { return v3(abs(a.x), abs(a.y), abs(a.z)); }

SHM_HD Float max_component_value(V3 a) { return max(a.x, max(a.y, a.z)); }
// Tuple3::max_component_index (vecmath/tuple.rs): x if x>y&&x>z? — restated from its definition below.
SHM_HD int max_component_index(V3 a) {
    // vecmath/tuple.rs max_component_index: if x > y { if x > z {0} else {2} } else { if y > z {1} else {2} }
    return (a.x > a.y) ? ((a.x > a.z) ? 0 : 2) : ((a.y > a.z) ? 1 : 2);
}
SHM_HD V3 permute(V3 a, int kx, int ky, int kz) { return v3(a[kx], a[ky], a[kz]); }

// tuple_fns.rs:40-52
SHM_HD V3 cross(V3 a, V3 b) {
    return v3(difference_of_products(a.y, b.z, a.z, b.y),
              difference_of_products(a.z, b.x, a.x, b.z),
              difference_of_products(a.x, b.y, a.y, b.x));
}
// tuple_fns.rs:68-78
SHM_HD Float dot(V3 v, V3 w) { return fma(v.x, w.x, sum_of_products(v.y, w.y, v.z, w.z)); }
SHM_HD Float abs_dot(V3 v, V3 w) { return abs(dot(v, w)); }
// length_fns.rs
SHM_HD Float length_squared(V3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; }
SHM_HD Float length(V3 v) { return sqrt(length_squared(v)); }
SHM_HD V3 normalize(V3 v) { return v / length(v); }
SHM_HD Float distance_squared(V3 a, V3 b) { return length_squared(a - b); }
SHM_HD Float distance(V3 a, V3 b) { return length(a - b); }

// tuple_fns.rs:162-183
SHM_HD Float angle_between(V3 v1, V3 v2) {
    if (dot(v1, v2) < 0.0f) return PI_F - 2.0f * safe_asin(length(v1 + v2) / 2.0f);
    return 2.0f * safe_asin(length(v2 - v1) / 2.0f);
}
// tuple_fns.rs:201-213
SHM_HD V3 face_forward(V3 a, V3 b) { return (dot(a, b) < 0.0f) ? -a : a; }
// vector.rs:1144-1146
SHM_HD V3 gram_schmidt(V3 v, V3 w) { return v - dot(v, w) * w; }
// vector.rs:1034-1042
SHM_HD void coordinate_system(V3 v, V3& v2, V3& v3o) {
    Float sign = copysign(1.0f, v.z);
    Float a = -1.0f / (sign + v.z);
    Float b = v.x * v.y * a;
    v2 = v3(1.0f + sign * sqr(v.x) * a, sign * b, -sign * v.x);
    v3o = v3(b, sign + sqr(v.y) * a, -v.y);
}
// vector.rs:1024-1032 / spherical.rs:8-14
SHM_HD V3 spherical_direction(Float sin_theta, Float cos_theta, Float phi) {
    return v3(clamp(sin_theta, -1.0f, 1.0f) * cos(phi), clamp(sin_theta, -1.0f, 1.0f) * sin(phi),
              clamp(cos_theta, -1.0f, 1.0f));
}

// V2 helpers (vector.rs Vector2f / point.rs Point2f)
SHM_HD V2 operator+(V2 a, V2 b) { return v2(a.x + b.x, a.y + b.y); }
SHM_HD V2 operator-(V2 a, V2 b) { return v2(a.x - b.x, a.y - b.y); }
SHM_HD V2 operator*(Float s, V2 a) { return v2(a.x * s, a.y * s); }
SHM_HD V2 operator*(V2 a, Float s) { return v2(a.x * s, a.y * s); }
SHM_HD Float length_squared(V2 v) { return v.x * v.x + v.y * v.y; }

// ---------------------------------------------------------------------------------------------
// Interval (interval.rs)
// ---------------------------------------------------------------------------------------------
struct Interval {
    Float low, high;
    SHM_HD Float midpoint() const { return (low + high) / 2.0f; }   // interval.rs:66
    SHM_HD Float width() const { return high - low; }               // interval.rs:78
    SHM_HD bool in_range(Float v) const { return v >= low && v <= high; }
};
SHM_HD Interval iv(Float v) { Interval r; r.low = v; r.high = v; return r; }                 // from_val
SHM_HD Interval iv2(Float lo, Float hi) { Interval r; r.low = lo; r.high = hi; return r; }  // raw
// interval.rs:26-33 Interval::new (sorts)
SHM_HD Interval iv_new(Float lo, Float hi) { return iv2(min(lo, hi), max(lo, hi)); }
// interval.rs:47-56
SHM_HD Interval iv_from_value_and_error(Float v, Float err) {
    if (err == 0.0f) return iv2(v, v);
    return iv2(sub_round_down(v, err), add_round_up(v, err));
}
SHM_HD Float min4(Float a, Float b, Float c, Float d) {
    // iter().fold(NAN, |a,b| a.min(b)) — NaN seed is ignored by f32::min
    return min(min(min(a, b), c), d);
}
SHM_HD Float max4(Float a, Float b, Float c, Float d) { return max(max(max(a, b), c), d); }
SHM_HD Interval operator-(Interval a) { return iv2(-a.high, -a.low); }
// interval.rs:345-347
SHM_HD Interval operator+(Interval a, Interval b) {
    return iv2(add_round_down(a.low, b.low), add_round_up(a.high, b.high));
}
// interval.rs:354-359 (note: low uses a.low - b.low, high uses a.high - b.high, as the reference writes it)
SHM_HD Interval operator-(Interval a, Interval b) {
    return iv2(sub_round_down(a.low, b.low), sub_round_up(a.high, b.high));
}
// interval.rs:366-384
SHM_HD Interval operator*(Interval a, Interval b) {
    Float lo = min4(mul_round_down(a.low, b.low), mul_round_down(a.high, b.low),
                    mul_round_down(a.low, b.high), mul_round_down(a.high, b.high));
    Float hi = max4(mul_round_up(a.low, b.low), mul_round_up(a.high, b.low),
                    mul_round_up(a.low, b.high), mul_round_up(a.high, b.high));
    return iv2(lo, hi);
}
// interval.rs:391-414
SHM_HD Interval operator/(Interval a, Interval b) {
    if (b.in_range(0.0f)) return iv2(-infinity(), infinity());
    Float lo = min4(div_round_down(a.low, b.low), div_round_down(a.high, b.low),
                    div_round_down(a.low, b.high), div_round_down(a.high, b.high));
    Float hi = max4(div_round_up(a.low, b.low), div_round_up(a.high, b.low),
                    div_round_up(a.low, b.high), div_round_up(a.high, b.high));
    return iv2(lo, hi);
}
// interval.rs:446-452 (Interval * Float has its own two-product form, then Interval::new sorts)
SHM_HD Interval operator*(Float f, Interval a) {
    if (f > 0.0f) return iv_new(mul_round_down(f, a.low), mul_round_up(f, a.high));
    return iv_new(mul_round_down(f, a.high), mul_round_up(f, a.low));
}
SHM_HD Interval operator*(Interval a, Float f) { return f * a; }
// interval.rs:426-443
SHM_HD Interval operator+(Interval a, Float f) { return a + iv(f); }
SHM_HD Interval operator-(Interval a, Float f) { return a - iv(f); }
SHM_HD Interval operator-(Float f, Interval a) { return iv(f) - a; }
// interval.rs:459-465
SHM_HD Interval operator/(Interval a, Float f) {
    if (f > 0.0f) return iv_new(div_round_down(a.low, f), div_round_up(a.high, f));
    return iv_new(div_round_down(a.high, f), div_round_up(a.low, f));
}
// interval.rs:84-103
SHM_HD Interval iv_sqr(Interval a) {
    Float alow = abs(a.low), ahigh = abs(a.high);
    if (alow > ahigh) { Float t = alow; alow = ahigh; ahigh = t; }
    if (a.in_range(0.0f)) return iv2(0.0f, mul_round_up(ahigh, ahigh));
    return iv2(mul_round_down(alow, alow), mul_round_up(ahigh, ahigh));
}
// interval.rs Sqrt for Interval: {sqrt_round_down(low), sqrt_round_up(high)}
SHM_HD Interval iv_sqrt(Interval a) { return iv2(sqrt_round_down(a.low), sqrt_round_up(a.high)); }
SHM_HD bool operator==(Interval a, Interval b) { return a.low == b.low && a.high == b.high; }

// Point3fi / Vector3fi (vecmath/point.rs:996-1190, vecmath/vector.rs:1310-1560)
struct P3i {
    Interval x, y, z;
    SHM_HD V3 mid() const { return v3(x.midpoint(), y.midpoint(), z.midpoint()); }  // From<Point3fi> for Point3f
    SHM_HD V3 error() const { return v3(x.width() / 2.0f, y.width() / 2.0f, z.width() / 2.0f); }  // point.rs:1015-1021
    SHM_HD bool is_exact() const { return x.width() == 0.0f && y.width() == 0.0f && z.width() == 0.0f; }
};
SHM_HD P3i p3i_exact(V3 p) { P3i r; r.x = iv(p.x); r.y = iv(p.y); r.z = iv(p.z); return r; }
SHM_HD P3i p3i_from_value_and_error(V3 p, V3 e) {  // point.rs:1007-1013
    P3i r;
    r.x = iv_from_value_and_error(p.x, e.x);
    r.y = iv_from_value_and_error(p.y, e.y);
    r.z = iv_from_value_and_error(p.z, e.z);
    return r;
}
SHM_HD P3i operator+(P3i a, P3i b) { P3i r; r.x = a.x + b.x; r.y = a.y + b.y; r.z = a.z + b.z; return r; }
SHM_HD P3i operator-(P3i a, P3i b) { P3i r; r.x = a.x - b.x; r.y = a.y - b.y; r.z = a.z - b.z; return r; }
SHM_HD P3i operator*(Interval s, P3i a) { P3i r; r.x = a.x * s; r.y = a.y * s; r.z = a.z * s; return r; }
SHM_HD P3i operator/(P3i a, Interval s) { P3i r; r.x = a.x / s; r.y = a.y / s; r.z = a.z / s; return r; }
// Vector3fi::length (vector.rs:1443-1451 -> length_fns.rs): x*x + y*y + z*z with the general Interval product
SHM_HD Interval length_squared(P3i v) { return v.x * v.x + v.y * v.y + v.z * v.z; }
SHM_HD Interval length(P3i v) { return iv_sqrt(length_squared(v)); }

// ---------------------------------------------------------------------------------------------
// Frame (frame.rs)
// ---------------------------------------------------------------------------------------------
struct Frame {
    V3 x, y, z;
    SHM_HD V3 to_local(V3 v) const { return v3(dot(v, x), dot(v, y), dot(v, z)); }         // frame.rs:41-43
    SHM_HD V3 from_local(V3 v) const { return v.x * x + v.y * y + v.z * z; }               // frame.rs:53-55
};
SHM_HD Frame frame_from_xz(V3 x, V3 z) { Frame f; f.x = x; f.y = cross(z, x); f.z = z; return f; }  // frame.rs:14-17
SHM_HD Frame frame_from_z(V3 z) { Frame f; f.z = z; coordinate_system(z, f.x, f.y); return f; }     // frame.rs:24-27

}  // namespace shm
