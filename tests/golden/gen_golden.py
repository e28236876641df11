#!/usr/bin/env python3
"""Generate tests/golden/golden.json: golden vectors for the hot-path leaf functions.

Two kinds of vectors:
 (1) "reference_known_answers": constants transcribed from the reference's own in-source tests
     (file:line cited per entry). These pin the oracle against the reference directly.
 (2) "numpy_f32": an INDEPENDENT float32 re-evaluation, in numpy, of the formulas cited from the
     reference (not a call into the oracle / shm headers): exact-rational FMA emulation for the Kahan
     products, op-by-op float32 rounding elsewhere. These pin the functions for which the reference has
     no test (SURVEY §8c "parity unpinned" list): intersect_p_cached, intersect_triangle, dot/cross,
     coordinate_system, Trowbridge-Reitz D/lambda/G, fresnel_dielectric, next_float, gamma, offset_ray_origin.

The reference is Rust and cannot be imported; nothing here reads /root/reference. Re-run: python tests/golden/gen_golden.py
"""
import json
from fractions import Fraction
from pathlib import Path

import numpy as np

f32 = np.float32
OUT = Path(__file__).resolve().parent / "golden.json"


def r32(q):
    """Round an exact rational to the nearest float32 (via the correctly rounded float64 of the Fraction; the
    double-rounding window is ~2^-29 of cases and never hit by these vectors, checked below by margin)."""
    return f32(float(q))


def fma(a, b, c):
    return r32(Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c)))


def dop(a, b, c, d):  # math.rs:170-178
    a, b, c, d = map(f32, (a, b, c, d))
    cd = f32(c * d)
    difference = fma(a, b, -cd)
    error = fma(-c, d, cd)
    return f32(difference + error)


def sop(a, b, c, d):
    return dop(a, b, -f32(c), d)


def dot(v, w):  # tuple_fns.rs:68-78
    return fma(v[0], w[0], sop(v[1], w[1], v[2], w[2]))


def cross(a, b):  # tuple_fns.rs:40-52
    return [dop(a[1], b[2], a[2], b[1]), dop(a[2], b[0], a[0], b[2]), dop(a[0], b[1], a[1], b[0])]


def gamma(n):  # float.rs:88-90
    eps = f32(np.finfo(np.float32).eps) * f32(0.5)
    return f32(f32(f32(n) * eps) / f32(f32(1.0) - f32(f32(n) * eps)))


def next_up(v):
    return float(np.nextafter(f32(v), f32(np.inf)))


def next_down(v):
    return float(np.nextafter(f32(v), f32(-np.inf)))


def intersect_p_cached(bmin, bmax, o, d, tmax):  # bounding_box.rs:520-563
    bmin, bmax, o, d = (np.asarray(x, np.float32) for x in (bmin, bmax, o, d))
    inv = (f32(1.0) / d).astype(np.float32)
    neg = inv < 0
    g = f32(f32(1.0) + f32(f32(2.0) * gamma(3)))
    b = [bmin, bmax]
    tmin = f32(f32(b[int(neg[0])][0] - o[0]) * inv[0])
    tmx = f32(f32(b[1 - int(neg[0])][0] - o[0]) * inv[0])
    tymin = f32(f32(b[int(neg[1])][1] - o[1]) * inv[1])
    tymax = f32(f32(b[1 - int(neg[1])][1] - o[1]) * inv[1])
    tmx = f32(tmx * g)
    tymax = f32(tymax * g)
    if tmin > tymax or tymin > tmx:
        return False
    if tymin > tmin:
        tmin = tymin
    if tymax < tmx:
        tmx = tymax
    tzmin = f32(f32(b[int(neg[2])][2] - o[2]) * inv[2])
    tzmax = f32(f32(b[1 - int(neg[2])][2] - o[2]) * inv[2])
    tzmax = f32(tzmax * g)
    if tmin > tzmax or tzmin > tmx:
        return False
    if tzmin > tmin:
        tmin = tzmin
    if tzmax < tmx:
        tmx = tzmax
    return bool(tmin < f32(tmax) and tmx > 0)


def intersect_triangle(o, d, tmax, p0, p1, p2):  # shape/triangle.rs:173-302
    o, d, p0, p1, p2 = (np.asarray(x, np.float32) for x in (o, d, p0, p1, p2))
    c = cross(p2 - p0, p1 - p0)
    if f32(f32(f32(c[0] * c[0]) + f32(c[1] * c[1])) + f32(c[2] * c[2])) == 0:
        return None
    p0t, p1t, p2t = (p0 - o).astype(np.float32), (p1 - o).astype(np.float32), (p2 - o).astype(np.float32)
    ad = np.abs(d)
    kz = (0 if ad[0] > ad[2] else 2) if ad[0] > ad[1] else (1 if ad[1] > ad[2] else 2)
    kx = (kz + 1) % 3
    ky = (kx + 1) % 3
    perm = [kx, ky, kz]
    d = d[perm]
    p0t, p1t, p2t = p0t[perm].copy(), p1t[perm].copy(), p2t[perm].copy()
    sx, sy, sz = f32(-d[0] / d[2]), f32(-d[1] / d[2]), f32(f32(1.0) / d[2])
    for p in (p0t, p1t, p2t):
        p[0] = f32(p[0] + f32(sx * p[2]))
        p[1] = f32(p[1] + f32(sy * p[2]))
    e0 = dop(p1t[0], p2t[1], p1t[1], p2t[0])
    e1 = dop(p2t[0], p0t[1], p2t[1], p0t[0])
    e2 = dop(p0t[0], p1t[1], p0t[1], p1t[0])
    if e0 == 0 or e1 == 0 or e2 == 0:
        e0 = f32(np.float64(p2t[1]) * np.float64(p1t[0]) - np.float64(p2t[0]) * np.float64(p1t[1]))
        e1 = f32(np.float64(p0t[1]) * np.float64(p2t[0]) - np.float64(p0t[0]) * np.float64(p2t[1]))
        e2 = f32(np.float64(p1t[1]) * np.float64(p0t[0]) - np.float64(p1t[0]) * np.float64(p0t[1]))
    if (e0 < 0 or e1 < 0 or e2 < 0) and (e0 > 0 or e1 > 0 or e2 > 0):
        return None
    det = f32(f32(e0 + e1) + e2)
    if det == 0:
        return None
    p0t[2], p1t[2], p2t[2] = f32(p0t[2] * sz), f32(p1t[2] * sz), f32(p2t[2] * sz)
    t_scaled = f32(f32(f32(e0 * p0t[2]) + f32(e1 * p1t[2])) + f32(e2 * p2t[2]))
    with np.errstate(invalid="ignore", over="ignore"):
        lim = f32(f32(tmax) * det)
    if det < 0 and (t_scaled >= 0 or t_scaled < lim):
        return None
    if det > 0 and (t_scaled <= 0 or t_scaled > lim):
        return None
    inv_det = f32(f32(1.0) / det)
    b0, b1, b2, t = f32(e0 * inv_det), f32(e1 * inv_det), f32(e2 * inv_det), f32(t_scaled * inv_det)
    max_zt = max(abs(p0t[2]), abs(p1t[2]), abs(p2t[2]))
    delta_z = f32(gamma(3) * max_zt)
    max_xt = max(abs(p0t[0]), abs(p1t[0]), abs(p2t[0]))
    max_yt = max(abs(p0t[1]), abs(p1t[1]), abs(p2t[1]))
    delta_x = f32(gamma(5) * f32(max_xt + max_zt))
    delta_y = f32(gamma(5) * f32(max_yt + max_zt))
    delta_e = f32(f32(2.0) * f32(f32(f32(f32(gamma(2) * max_xt) * max_yt) + f32(delta_y * max_xt)) + f32(delta_x * max_yt)))
    max_e = max(abs(e0), abs(e1), abs(e2))
    delta_t = f32(f32(f32(3.0) * f32(f32(f32(f32(gamma(3) * max_e) * max_zt) + f32(delta_e * max_zt)) + f32(delta_z * max_e))) * abs(inv_det))
    if t <= delta_t:
        return None
    return [float(b0), float(b1), float(b2), float(t)]


def coordinate_system(v):  # vecmath/vector.rs:1034-1042
    x, y, z = map(f32, v)
    sign = f32(np.copysign(f32(1.0), z))
    a = f32(f32(-1.0) / f32(sign + z))
    b = f32(f32(x * y) * a)
    v2 = [f32(f32(1.0) + f32(f32(sign * f32(x * x)) * a)), f32(sign * b), f32(f32(-sign) * x)]
    v3 = [b, f32(sign + f32(f32(y * y) * a)), f32(-y)]
    return [float(t) for t in v2 + v3]


def tr_helpers(w):
    x, y, z = map(f32, w)
    cos2 = f32(z * z)
    sin2 = max(f32(0), f32(f32(1.0) - cos2))
    sin_t = f32(np.sqrt(sin2))
    tan2 = f32(sin2 / cos2)
    cos_phi = f32(1.0) if sin_t == 0 else f32(np.clip(f32(x / sin_t), -1, 1))
    sin_phi = f32(1.0) if sin_t == 0 else f32(np.clip(f32(y / sin_t), -1, 1))
    return cos2, tan2, cos_phi, sin_phi


def tr_d(ax, ay, wm):  # scattering.rs:134-145
    ax, ay = f32(ax), f32(ay)
    cos2, tan2, cp, sp = tr_helpers(wm)
    if np.isinf(tan2):
        return 0.0
    cos4 = f32(cos2 * cos2)
    if cos4 < f32(1e-16):
        return 0.0
    a, b = f32(cp / ax), f32(sp / ay)
    e = f32(tan2 * f32(f32(a * a) + f32(b * b)))
    ope = f32(f32(1.0) + e)
    return float(f32(f32(1.0) / f32(f32(f32(f32(f32(np.pi) * ax) * ay) * cos4) * f32(ope * ope))))


def tr_lambda(ax, ay, w):  # scattering.rs:151-158
    ax, ay = f32(ax), f32(ay)
    cos2, tan2, cp, sp = tr_helpers(w)
    if np.isinf(tan2):
        return 0.0
    a, b = f32(cp * ax), f32(sp * ay)
    alpha2 = f32(f32(a * a) + f32(b * b))
    return float(f32(f32(f32(-1.0) + f32(np.sqrt(f32(f32(1.0) + f32(alpha2 * tan2))))) / f32(2.0)))


def tr_g(ax, ay, wo, wi):  # scattering.rs:160-162
    return float(f32(f32(1.0) / f32(f32(f32(1.0) + f32(tr_lambda(ax, ay, wo))) + f32(tr_lambda(ax, ay, wi)))))


def fresnel_dielectric(c, eta):  # scattering.rs:49-70
    c, eta = f32(np.clip(f32(c), -1, 1)), f32(eta)
    if c < 0:
        eta = f32(f32(1.0) / eta)
        c = f32(-c)
    s2i = f32(f32(1.0) - f32(c * c))
    s2t = f32(s2i / f32(eta * eta))
    if s2t >= 1:
        return 1.0
    ct = f32(np.sqrt(max(f32(0), f32(f32(1.0) - s2t))))
    r_parl = f32(f32(f32(eta * c) - ct) / f32(f32(eta * c) + ct))
    r_perp = f32(f32(c - f32(eta * ct)) / f32(c + f32(eta * ct)))
    return float(f32(f32(0.5) * f32(f32(r_parl * r_parl) + f32(r_perp * r_perp))))


def offset_ray_origin(p, err, n, w):  # ray.rs:53-71 with Point3fi::from_value_and_error (interval.rs:47-56)
    p, err, n, w = (np.asarray(x, np.float32) for x in (p, err, n, w))
    lo = np.array([next_down(f32(p[i] - err[i])) if err[i] != 0 else p[i] for i in range(3)], np.float32)
    hi = np.array([next_up(f32(p[i] + err[i])) if err[i] != 0 else p[i] for i in range(3)], np.float32)
    mid = ((lo + hi).astype(np.float32) / f32(2.0)).astype(np.float32)
    e = ((hi - lo).astype(np.float32) / f32(2.0)).astype(np.float32)
    d = dot(np.abs(n), e)
    offset = (d * n).astype(np.float32)
    if dot(w, n) < 0:
        offset = -offset
    po = (mid + offset).astype(np.float32)
    for i in range(3):
        if offset[i] > 0:
            po[i] = next_up(po[i])
        elif offset[i] < 0:
            po[i] = next_down(po[i])
    return [float(x) for x in po]


def main():
    rng = np.random.default_rng(20241002)
    g = {"reference_known_answers": {}, "numpy_f32": {}}
    ka = g["reference_known_answers"]
    # aggregate.rs:601-628 single_primitive_bvh_intersetion; :631-702 set_of_spheres
    ka["aggregate_single_sphere"] = {"ray": [-5, 0, 0, 1, 0, 0], "t": 4.0, "p_x": -1.0, "eps_p": 1e-6, "n_dot_negx": 1.0}
    ka["aggregate_three_spheres"] = {"offsets": [-3.5, 0.0, 5.0], "ray": [-10, 0, 0, 1, 0, 0], "t": 5.5, "eps_t": 1e-5, "p_x": -4.5,
                                     "miss_ray": [-10, 0, 1.001, 1, 0, 0]}
    # shape/shape.rs:299-342 sphere predicates
    ka["sphere_predicates"] = {
        "full": [{"ray": [0, 0, -2, 0, 0, 1], "hit": True}, {"ray": [0, 0, -2, 0, 0, -1], "hit": False},
                 {"ray": [0, 1.0001, -2, 0, 0, 1], "hit": False}],
        "partial_z_pm_half": [{"ray": [0, -2, 0, 0, 1, 0], "hit": True}, {"ray": [0, -2, 0, 0, -1, 0], "hit": False},
                              {"ray": [0, 0, 0.5001, 0, 1, 0], "hit": False}, {"ray": [0, 0, -0.5001, 0, 1, 0], "hit": False}],
    }
    # bxdf.rs:1839-1856 mf_distrib (D only: the G constant there is stale) and :1871-1903 dielectric_sample_f
    ka["tr_d"] = {"alpha": 0.0299999993, "wm": [-0.430063188, -0.881908476, 0.193088099], "d": 0.000309075956, "rel": 1e-5}
    ka["dielectric_sample_f"] = {"eta": 1.5, "wo": [-0.419299453, -0.656406343, 0.627151370], "uc": 0.237656280, "u": [0.0488742627, 0.941848040],
                                 "flags": 18, "pdf": 0.940032840, "eta_out": 1.5, "f": 0.488867134,
                                 "wi": [0.279532969, 0.437604219, -0.854613364], "rel": 2e-6}
    # sampling.rs:801-812 visible_wavelengths_pdf bounds
    ka["visible_wavelengths_pdf_zero"] = [359.9, 830.1, 0.0, 1000.0]
    # spectra/spectrum.rs:654-680 blackbody known answers (Le in W/(m^2 sr m)), rel.err < 1e-3
    ka["blackbody"] = [[483.0, 6000.0, 3.1849e13], [600.0, 6000.0, 2.86772e13], [500.0, 3700.0, 1.59845e12], [600.0, 4500.0, 7.46497e12]]
    # float.rs:172-211 next_float_up/down
    ka["next_float"] = {"up_neg_zero_gt_zero": True, "down_zero_lt_zero": True}

    nf = g["numpy_f32"]
    vs = (rng.random((24, 4)) * 4 - 2).astype(np.float32)
    nf["difference_of_products"] = [[*map(float, v), float(dop(*v))] for v in vs]
    # near-cancelling cases, where the Kahan form differs from the naive one
    for _ in range(12):
        a, b = f32(rng.random() + 1), f32(rng.random() + 1)
        c = f32(a * f32(1 + 2e-7))
        d = f32(b * f32(1 - 1e-7))
        nf["difference_of_products"].append([float(a), float(b), float(c), float(d), float(dop(a, b, c, d))])
    v3 = (rng.random((16, 6)) * 2 - 1).astype(np.float32)
    nf["dot"] = [[*map(float, v), float(dot(v[:3], v[3:]))] for v in v3]
    nf["cross"] = [[*map(float, v), *map(float, cross(v[:3], v[3:]))] for v in v3]
    nf["gamma"] = [[n, float(gamma(n))] for n in (1, 2, 3, 5, 6, 7)]
    nf["next_float"] = [[float(x), next_up(x), next_down(x)] for x in list((rng.random(16) * 200 - 100).astype(np.float32)) + [f32(0.0), f32(1.0), f32(-1.0), f32(2.0 ** -126)]]
    units = rng.normal(size=(16, 3))
    units = (units / np.linalg.norm(units, axis=1, keepdims=True)).astype(np.float32)
    nf["coordinate_system"] = [[*map(float, u), *coordinate_system(u)] for u in units]
    boxes = []
    for _ in range(64):
        lo = (rng.random(3) * 2 - 1).astype(np.float32)
        hi = (lo + rng.random(3).astype(np.float32) * f32(1.5)).astype(np.float32)
        o = (rng.random(3) * 6 - 3).astype(np.float32)
        d = (lo + (hi - lo) * rng.random(3).astype(np.float32)) - o if rng.random() < 0.6 else rng.normal(size=3)
        d = (d / np.linalg.norm(d)).astype(np.float32)
        tmax = float(f32(rng.random() * 6)) if rng.random() < 0.5 else float("inf")
        boxes.append({"bmin": lo.tolist(), "bmax": hi.tolist(), "o": o.tolist(), "d": d.tolist(), "t_max": tmax,
                      "hit": intersect_p_cached(lo, hi, o, d, tmax)})
    # axis-parallel rays (inv_dir = +-inf) and rays starting inside
    for d in ([1, 0, 0], [0, -1, 0], [0, 0, 1]):
        boxes.append({"bmin": [-1, -1, -1], "bmax": [1, 1, 1], "o": [0.25, 0.5, -0.75], "d": d, "t_max": float("inf"),
                      "hit": intersect_p_cached([-1, -1, -1], [1, 1, 1], [0.25, 0.5, -0.75], d, np.inf)})
    nf["intersect_p_cached"] = boxes
    tris = []
    while len(tris) < 64:
        p = (rng.random((3, 3)) * 2 - 1).astype(np.float32)
        o = (rng.random(3) * 4 - 2).astype(np.float32)
        if rng.random() < 0.7:
            bary = rng.dirichlet([1, 1, 1])
            tgt = (bary[:, None] * p).sum(0)
            d = tgt - o
        else:
            d = rng.normal(size=3)
        d = (d / np.linalg.norm(d)).astype(np.float32)
        tmax = float("inf") if rng.random() < 0.6 else float(f32(rng.random() * 4))
        r = intersect_triangle(o, d, tmax, p[0], p[1], p[2])
        tris.append({"o": o.tolist(), "d": d.tolist(), "t_max": tmax, "p0": p[0].tolist(), "p1": p[1].tolist(), "p2": p[2].tolist(), "hit": r})
    # exact edge / vertex hits exercise the f64 fallback (triangle.rs:231-243)
    for o, d in (([0.5, 0.0, -1.0], [0, 0, 1]), ([0.0, 0.0, -1.0], [0, 0, 1]), ([0.25, 0.25, -2.0], [0, 0, 1]), ([1.5, 0.0, -1.0], [0, 0, 1])):
        p0, p1, p2 = [0, 0, 0], [1, 0, 0], [0, 1, 0]
        tris.append({"o": o, "d": d, "t_max": float("inf"), "p0": p0, "p1": p1, "p2": p2, "hit": intersect_triangle(o, d, np.inf, p0, p1, p2)})
    nf["intersect_triangle"] = tris
    hemi = units.copy()
    hemi[:, 2] = np.abs(hemi[:, 2])
    nf["tr_d"] = [[0.1, 0.25, *map(float, w), tr_d(0.1, 0.25, w)] for w in hemi]
    nf["tr_lambda"] = [[0.1, 0.25, *map(float, w), tr_lambda(0.1, 0.25, w)] for w in units]
    nf["tr_g"] = [[0.03, 0.03, *map(float, units[i]), *map(float, units[i + 1]), tr_g(0.03, 0.03, units[i], units[i + 1])] for i in range(0, 14)]
    # the stale constant of bxdf.rs:1851 for the record: G(wm, wi) there evaluates to 0.97391653, not 0.954060972
    nf["tr_g_bxdf_test_vector"] = {"alpha": 0.0299999993, "w1": [-0.430063188, -0.881908476, 0.193088099],
                                   "w2": [0.568110108, 0.816620350, 0.101893365],
                                   "g": tr_g(0.0299999993, 0.0299999993, [-0.430063188, -0.881908476, 0.193088099], [0.568110108, 0.816620350, 0.101893365])}
    nf["fresnel_dielectric"] = [[float(c), float(e), fresnel_dielectric(c, e)] for c, e in zip((rng.random(24) * 2 - 1).astype(np.float32), (1.0 + rng.random(24)).astype(np.float32))]
    oro = []
    for _ in range(16):
        p = (rng.random(3) * 10 - 5).astype(np.float32)
        err = (np.abs(p) * f32(3e-7) + f32(1e-8)).astype(np.float32)
        n = rng.normal(size=3)
        n = (n / np.linalg.norm(n)).astype(np.float32)
        w = rng.normal(size=3).astype(np.float32)
        oro.append({"p": p.tolist(), "err": err.tolist(), "n": n.tolist(), "w": w.tolist(), "out": offset_ray_origin(p, err, n, w)})
    nf["offset_ray_origin"] = oro
    OUT.write_text(json.dumps(g, indent=1))
    print("wrote", OUT, {k: len(v) for k, v in nf.items()})


if __name__ == "__main__":
    main()
