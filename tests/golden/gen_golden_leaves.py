#!/usr/bin/env python3
"""Generate tests/golden/golden_leaves.json: fixtures for the shared leaf functions that round 1 left unpinned
(VERDICT r01 "what's weak" 1): rough Conductor / Dielectric f, pdf, sample_f, TrowbridgeReitz::sample_wm, Triangle::
interaction_from_intersection, Sphere::{sample,pdf}_with_context, DiffuseAreaLight::l / pdf_li, PixelSensor::to_sensor_rgb +
RgbFilm::add_sample, PerspectiveCamera::generate_ray_differential with a thin lens.

Everything here is an INDEPENDENT re-evaluation of the formulas cited from the reference (paths relative to
/root/reference/src), written in numpy from the Rust text — it does not call the oracle, the shm headers or the product:
  "f32":  op-by-op float32 with exact-rational FMA emulation (the helpers of gen_golden.py) for functions made of
          + - * / sqrt fma only; compared BITWISE by tests/test_leaf_golden.py;
  "f64":  float64 / complex128 evaluation of the same formulas for functions that go through sin / cos / atan2 / acos / complex
          sqrt (Rust std / num-complex there, shm::fp.h here: 1-2 ulp apart by construction); compared within the stated
          relative tolerance, on inputs kept away from branch boundaries.
The reference is Rust and cannot be imported; nothing here reads /root/reference. Re-run: python tests/golden/gen_golden_leaves.py
"""
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent))
from gen_golden import (cross, dop, dot, f32, fresnel_dielectric, gamma, next_down, next_up, coordinate_system, tr_d, tr_g,  # noqa: E402
                        tr_lambda)

OUT = Path(__file__).resolve().parent / "golden_leaves.json"
PI32 = f32(np.pi)


def v32(x):
    return np.asarray(x, np.float32)


def fl(a):
    return [float(x) for x in np.asarray(a).ravel()]


def length_squared(v):  # vecmath/length_fns.rs:6-13: x*x + y*y + z*z, left to right
    return f32(f32(f32(v[0] * v[0]) + f32(v[1] * v[1])) + f32(v[2] * v[2]))


def normalize(v):  # vecmath/normalize.rs:9-13: v / length, component-wise
    ln = f32(np.sqrt(length_squared(v)))
    return v32([f32(v[0] / ln), f32(v[1] / ln), f32(v[2] / ln)])


def face_forward(a, b):  # tuple_fns.rs:202-214
    return (-a).astype(np.float32) if dot(a, b) < 0 else a


# ---- rough dielectric f / pdf, float32-exact (bxdf.rs:532-583, 718-777) ----------------------------------------------------
def tr_pdf(ax, ay, w, wm):  # scattering.rs:164-170: g1(w) / abs_cos_theta(w) * d(wm) * abs_dot(w, wm)
    g1 = f32(f32(1.0) / f32(f32(1.0) + f32(tr_lambda(ax, ay, w))))
    return f32(f32(f32(g1 / f32(abs(w[2]))) * f32(tr_d(ax, ay, wm))) * f32(abs(dot(w, wm))))


def dielectric_generalized_half_vector(eta, wo, wi):
    cos_o, cos_i = f32(wo[2]), f32(wi[2])
    reflect = f32(cos_i * cos_o) > 0
    etap = f32(1.0) if reflect else (f32(eta) if cos_o > 0 else f32(f32(1.0) / f32(eta)))
    wm = v32([f32(f32(wi[k] * etap) + wo[k]) for k in range(3)])
    return cos_o, cos_i, reflect, etap, wm


def dielectric_f(eta, ax, ay, wo, wi):  # TransportMode::Radiance
    wo, wi = v32(wo), v32(wi)
    cos_o, cos_i, reflect, etap, wm = dielectric_generalized_half_vector(eta, wo, wi)
    if cos_i == 0 or cos_o == 0 or length_squared(wm) == 0:
        return 0.0
    wm = face_forward(normalize(wm), v32([0, 0, 1]))
    if f32(dot(wm, wi) * cos_i) < 0 or f32(dot(wm, wo) * cos_o) < 0:
        return 0.0
    fr = f32(fresnel_dielectric(dot(wo, wm), eta))
    d, g = f32(tr_d(ax, ay, wm)), f32(tr_g(ax, ay, wo, wi))
    if reflect:
        return float(f32(f32(f32(d * g) * fr) / f32(abs(f32(f32(f32(4.0) * cos_i) * cos_o)))))
    sm = f32(dot(wi, wm) + f32(dot(wo, wm) / etap))
    denom = f32(f32(f32(sm * sm) * cos_i) * cos_o)
    ft = f32(f32(f32(d * f32(f32(1.0) - fr)) * g) * f32(abs(f32(f32(dot(wi, wm) * dot(wo, wm)) / denom))))
    return float(f32(ft / f32(etap * etap)))


def dielectric_pdf(eta, ax, ay, wo, wi):  # sample_flags = ALL
    wo, wi = v32(wo), v32(wi)
    cos_o, cos_i, reflect, etap, wm = dielectric_generalized_half_vector(eta, wo, wi)
    if cos_i == 0 or cos_o == 0 or length_squared(wm) == 0:
        return 0.0
    wm = face_forward(normalize(wm), v32([0, 0, 1]))
    if f32(dot(wm, wi) * cos_i) < 0 or f32(dot(wm, wo) * cos_o) < 0:
        return 0.0
    r = f32(fresnel_dielectric(dot(wo, wm), eta))
    t = f32(f32(1.0) - r)
    if reflect:
        return float(f32(f32(f32(tr_pdf(ax, ay, wo, wm) / f32(f32(4.0) * f32(abs(dot(wo, wm))))) * r) / f32(r + t)))
    sm = f32(dot(wi, wm) + f32(dot(wo, wm) / etap))
    denom = f32(sm * sm)
    dwm_dwi = f32(f32(abs(dot(wi, wm))) / denom)
    return float(f32(f32(f32(tr_pdf(ax, ay, wo, wm) * dwm_dwi) * t) / f32(r + t)))


def conductor_pdf(ax, ay, wo, wi):  # bxdf.rs:425-444
    wo, wi = v32(wo), v32(wi)
    if not f32(wo[2] * wi[2]) > 0:
        return 0.0
    wm = (wo + wi).astype(np.float32)
    if length_squared(wm) == 0:
        return 0.0
    wm = face_forward(normalize(wm), v32([0, 0, 1]))
    return float(f32(tr_pdf(ax, ay, wo, wm) / f32(f32(4.0) * f32(abs(dot(wo, wm))))))


# ---- float64 evaluations (tolerance fixtures) --------------------------------------------------------------------------------
def d64(ax, ay, wm):
    c2 = wm[2] ** 2
    s2 = max(0.0, 1 - c2)
    if c2 == 0:
        return 0.0
    t2 = s2 / c2
    st = np.sqrt(s2)
    cp, sp = (1.0, 1.0) if st == 0 else (np.clip(wm[0] / st, -1, 1), np.clip(wm[1] / st, -1, 1))
    e = t2 * ((cp / ax) ** 2 + (sp / ay) ** 2)
    return 1.0 / (np.pi * ax * ay * c2 * c2 * (1 + e) ** 2)


def lam64(ax, ay, w):
    c2 = w[2] ** 2
    s2 = max(0.0, 1 - c2)
    if c2 == 0:
        return 0.0
    st = np.sqrt(s2)
    cp, sp = (1.0, 1.0) if st == 0 else (np.clip(w[0] / st, -1, 1), np.clip(w[1] / st, -1, 1))
    a2 = (cp * ax) ** 2 + (sp * ay) ** 2
    return (-1 + np.sqrt(1 + a2 * s2 / c2)) / 2


def g64(ax, ay, wo, wi):
    return 1 / (1 + lam64(ax, ay, wo) + lam64(ax, ay, wi))


def pdf64(ax, ay, w, wm):
    return 1 / (1 + lam64(ax, ay, w)) / abs(w[2]) * d64(ax, ay, wm) * abs(np.dot(w, wm))


def fresnel_complex64(c, eta, k):  # scattering.rs:77-90
    c = np.clip(c, 0, 1)
    e = complex(eta, k)
    s2t = (1 - c * c) / (e * e)
    ct = np.sqrt(1 - s2t + 0j)
    r_parl = (e * c - ct) / (e * c + ct)
    r_perp = (c - e * ct) / (c + e * ct)
    return (abs(r_parl) ** 2 + abs(r_perp) ** 2) / 2


def fresnel_dielectric64(c, eta):
    c = np.clip(c, -1, 1)
    if c < 0:
        eta, c = 1 / eta, -c
    s2t = (1 - c * c) / (eta * eta)
    if s2t >= 1:
        return 1.0
    ct = np.sqrt(max(0.0, 1 - s2t))
    return 0.5 * (((eta * c - ct) / (eta * c + ct)) ** 2 + ((c - eta * ct) / (c + eta * ct)) ** 2)


def conductor_f64(ax, ay, eta4, k4, wo, wi):  # bxdf.rs:348-375
    wo, wi = np.asarray(wo, float), np.asarray(wi, float)
    if wo[2] * wi[2] <= 0:
        return [0.0] * 4
    wm = wo + wi
    wm /= np.linalg.norm(wm)
    dg = d64(ax, ay, wm) * g64(ax, ay, wo, wi) / (4 * abs(wo[2]) * abs(wi[2]))
    return [dg * fresnel_complex64(abs(np.dot(wo, wm)), e, k) for e, k in zip(eta4, k4)]


def sample_wm64(ax, ay, w, u):  # scattering.rs:172-206
    w = np.asarray(w, float)
    wh = np.array([ax * w[0], ay * w[1], w[2]])
    wh /= np.linalg.norm(wh)
    if wh[2] < 0:
        wh = -wh
    if wh[2] < 0.99999:
        t1 = np.cross([0, 0, 1.0], wh)
        t1 /= np.linalg.norm(t1)
    else:
        t1 = np.array([1.0, 0, 0])
    t2 = np.cross(wh, t1)
    r, th = np.sqrt(u[0]), 2 * np.pi * u[1]  # sample_uniform_disk_polar, sampling.rs:341-345
    px, py = r * np.cos(th), r * np.sin(th)
    h = np.sqrt(1 - px * px)
    t = (1 + wh[2]) / 2
    py = (1 - t) * h + t * py  # lerp(t, h, p.y), math.rs:246-252
    pz = np.sqrt(max(0.0, 1 - px * px - py * py))
    nh = px * t1 + py * t2 + pz * wh
    out = np.array([ax * nh[0], ay * nh[1], max(1e-6, nh[2])])
    return out / np.linalg.norm(out)


def refract64(wi, n, eta):  # scattering.rs:21-45
    wi, n = np.asarray(wi, float), np.asarray(n, float)
    c = np.dot(n, wi)
    if c < 0:
        eta, c, n = 1 / eta, -c, -n
    s2i = max(0.0, 1 - c * c)
    s2t = s2i / (eta * eta)
    if s2t >= 1:
        return None
    ct = np.sqrt(1 - s2t)
    return -wi / eta + (c / eta - ct) * n, eta


def conductor_sample_f64(ax, ay, eta4, k4, wo, u):  # bxdf.rs:377-423 (rough branch)
    wo = np.asarray(wo, float)
    if wo[2] == 0:
        return None
    wm = sample_wm64(ax, ay, wo, u)
    wi = -wo + 2 * np.dot(wo, wm) * wm
    if wo[2] * wi[2] <= 0:
        return None
    pdf = pdf64(ax, ay, wo, wm) / (4 * abs(np.dot(wo, wm)))
    dg = d64(ax, ay, wm) * g64(ax, ay, wo, wi) / (4 * abs(wo[2]) * abs(wi[2]))
    f = [dg * fresnel_complex64(abs(np.dot(wo, wm)), e, k) for e, k in zip(eta4, k4)]
    return {"f": f, "wi": wi.tolist(), "pdf": pdf}


def dielectric_sample_f64(eta, ax, ay, wo, uc, u):  # bxdf.rs:649-712 (rough branch), Radiance mode, all lobes
    wo = np.asarray(wo, float)
    wm = sample_wm64(ax, ay, wo, u)
    r = fresnel_dielectric64(np.dot(wo, wm), eta)
    t = 1 - r
    margin = abs(uc - r / (r + t))
    if uc < r / (r + t):
        wi = -wo + 2 * np.dot(wo, wm) * wm
        if wo[2] * wi[2] <= 0:
            return None, margin
        pdf = pdf64(ax, ay, wo, wm) / (4 * abs(np.dot(wo, wm))) * r / (r + t)
        f = d64(ax, ay, wm) * g64(ax, ay, wo, wi) * r / (4 * wi[2] * wo[2])
        return {"f": f, "wi": wi.tolist(), "pdf": pdf, "reflect": True, "eta": 1.0}, margin
    rf = refract64(wo, wm, eta)
    if rf is None:
        return None, margin
    wi, etap = rf
    if wo[2] * wi[2] > 0 or wi[2] == 0:
        return None, margin
    denom = (np.dot(wi, wm) + np.dot(wo, wm) / etap) ** 2
    pdf = pdf64(ax, ay, wo, wm) * abs(np.dot(wi, wm)) / denom * t / (r + t)
    ft = t * d64(ax, ay, wm) * g64(ax, ay, wo, wi) * abs(np.dot(wi, wm) * np.dot(wo, wm) / (wi[2] * wo[2] * denom)) / (etap * etap)
    return {"f": ft, "wi": wi.tolist(), "pdf": pdf, "reflect": False, "eta": etap}, margin


# ---- Triangle::interaction_from_intersection, float32-exact (triangle.rs:305-504) -----------------------------------------
def dop_vec(a, b, c, d):  # math.rs:214-219 difference_of_products_float_vec: plain component-wise ops, no FMA
    a, c = f32(a), f32(c)
    cd = (c * d).astype(np.float32)
    difference = ((a * b).astype(np.float32) - cd).astype(np.float32)
    error = ((f32(-c) * d).astype(np.float32) + cd).astype(np.float32)
    return (difference + error).astype(np.float32)


def cs(v):  # Vector3f::coordinate_system -> (v2, v3)
    r = coordinate_system(v)
    return v32(r[:3]), v32(r[3:])


def cross32(a, b):
    return v32(cross(a, b))


def from_value_and_error(p, err):  # interval.rs:47-56 per component
    lo = v32([next_down(f32(p[i] - err[i])) if err[i] != 0 else p[i] for i in range(3)])
    hi = v32([next_up(f32(p[i] + err[i])) if err[i] != 0 else p[i] for i in range(3)])
    return lo, hi


def triangle_interaction(p, n, s, uv, flip, b):
    p0, p1, p2 = v32(p[0]), v32(p[1]), v32(p[2])
    b0, b1, b2 = map(f32, b)
    uvs = [v32([0, 0]), v32([1, 0]), v32([1, 1])] if uv is None else [v32(x) for x in uv]
    duv02, duv12 = (uvs[0] - uvs[2]).astype(np.float32), (uvs[1] - uvs[2]).astype(np.float32)
    dp02, dp12 = (p0 - p2).astype(np.float32), (p1 - p2).astype(np.float32)
    det = dop(duv02[0], duv12[1], duv02[1], duv12[0])
    degenerate = abs(det) < f32(1e-9)
    if not degenerate:
        inv_det = f32(f32(1.0) / det)
        dpdu = (dop_vec(duv12[1], dp02, duv02[1], dp12) * inv_det).astype(np.float32)
        dpdv = (dop_vec(duv02[0], dp12, duv12[0], dp02) * inv_det).astype(np.float32)
    else:
        dpdu = dpdv = v32([0, 0, 0])
    if degenerate or length_squared(cross32(dpdu, dpdv)) == 0:
        ng = cross32((p2 - p0).astype(np.float32), (p1 - p0).astype(np.float32))
        assert length_squared(ng) != 0  # (the f64 retry of :343-365 is not exercised by these vectors)
        dpdu, dpdv = cs(normalize(ng))
    p_hit = ((b0 * p0).astype(np.float32) + (b1 * p1).astype(np.float32)).astype(np.float32)
    p_hit = (p_hit + (b2 * p2).astype(np.float32)).astype(np.float32)
    uv_hit = ((b0 * uvs[0]).astype(np.float32) + (b1 * uvs[1]).astype(np.float32)).astype(np.float32)
    uv_hit = (uv_hit + (b2 * uvs[2]).astype(np.float32)).astype(np.float32)
    p_abs = (np.abs((b0 * p0).astype(np.float32)) + np.abs((b1 * p1).astype(np.float32))).astype(np.float32)
    p_abs = (p_abs + np.abs((b2 * p2).astype(np.float32))).astype(np.float32)
    p_err = (gamma(7) * p_abs).astype(np.float32)
    lo, hi = from_value_and_error(p_hit, p_err)
    ng = normalize(cross32(dp02, dp12))
    n_geo = (-ng).astype(np.float32) if flip else ng
    ns_out, ss_out, ts_out = n_geo, dpdu, dpdv
    dndu = dndv = v32([0, 0, 0])
    if n is not None or s is not None:
        if n is None:
            ns = n_geo
        else:
            nn = ((b0 * v32(n[0])).astype(np.float32) + (b1 * v32(n[1])).astype(np.float32)).astype(np.float32)
            nn = (nn + (b2 * v32(n[2])).astype(np.float32)).astype(np.float32)
            ns = normalize(nn) if length_squared(nn) > 0 else n_geo
        if s is None:
            ss = dpdu
        else:
            sv = ((b0 * v32(s[0])).astype(np.float32) + (b1 * v32(s[1])).astype(np.float32)).astype(np.float32)
            sv = (sv + (b2 * v32(s[2])).astype(np.float32)).astype(np.float32)
            ss = dpdu if length_squared(sv) == 0 else sv
        ts = cross32(ns, ss)
        if length_squared(ts) > 0:
            ss = cross32(ts, ns)
        else:
            ss, ts = cs(ns)
        if n is not None:
            if degenerate:
                dn = cross32((v32(n[2]) - v32(n[0])).astype(np.float32), (v32(n[1]) - v32(n[0])).astype(np.float32))
                if length_squared(dn) != 0:
                    dndu, dndv = cs(dn)
            else:
                inv_det = f32(f32(1.0) / det)
                dn1, dn2 = (v32(n[0]) - v32(n[2])).astype(np.float32), (v32(n[1]) - v32(n[2])).astype(np.float32)
                dndu = (dop_vec(duv12[1], dn1, duv02[1], dn2) * inv_det).astype(np.float32)
                dndv = (dop_vec(duv02[0], dn2, duv12[0], dn1) * inv_det).astype(np.float32)
        # set_shading_geometry(ns, ss, ts, dndu, dndv, true), interaction.rs:379-405
        ns_out = ns
        n_geo = face_forward(n_geo, ns_out)
        ss_out, ts_out = ss, ts
    return {"pi_low": fl(lo), "pi_high": fl(hi), "uv": fl(uv_hit), "n": fl(n_geo), "dpdu": fl(dpdu), "dpdv": fl(dpdv),
            "shading_n": fl(ns_out), "shading_dpdu": fl(ss_out), "shading_dpdv": fl(ts_out), "shading_dndu": fl(dndu), "shading_dndv": fl(dndv)}


# ---- Sphere::{sample,pdf}_with_context in float64 (sphere.rs:339-457); identity object transform + translation -------------
def sphere_sample64(center, radius, ctx_p, u):
    c, p = np.asarray(center, float), np.asarray(ctx_p, float)
    dist = np.linalg.norm(p - c)
    assert dist > radius * 1.05  # outside: the cone branch (the inside branch goes through Sphere::sample, pinned by the render tests)
    sin_max = radius / dist
    sin2_max = sin_max ** 2
    cos_max = np.sqrt(max(0.0, 1 - sin2_max))
    one_minus = 1 - cos_max
    cos_t = (cos_max - 1) * u[0] + 1
    sin2_t = 1 - cos_t ** 2
    if sin2_max < 0.00068523:
        sin2_t = sin2_max * u[0]
        cos_t = np.sqrt(1 - sin2_t)
        one_minus = sin2_max / 2
    cos_a = sin2_t / sin_max + cos_t * np.sqrt(max(0.0, 1 - sin2_t / sin_max ** 2))
    sin_a = np.sqrt(max(0.0, 1 - cos_a ** 2))
    phi = u[1] * 2 * np.pi
    w = np.array([sin_a * np.cos(phi), sin_a * np.sin(phi), cos_a])
    z = (c - p) / dist
    # Frame::from_z -> coordinate_system(z) (frame.rs:40-44, vector.rs:1034-1042)
    sign = np.copysign(1.0, z[2])
    a = -1 / (sign + z[2])
    b = z[0] * z[1] * a
    x = np.array([1 + sign * z[0] ** 2 * a, sign * b, -sign * z[0]])
    y = np.array([b, sign + z[1] ** 2 * a, -z[1]])
    n = -(w[0] * x + w[1] * y + w[2] * z)  # from_local(-w)
    return {"p": (c + n * radius).tolist(), "n": n.tolist(), "pdf": 1 / (2 * np.pi * one_minus)}


def sphere_pdf64(center, radius, ctx_p):
    c, p = np.asarray(center, float), np.asarray(ctx_p, float)
    sin2_max = radius ** 2 / np.dot(p - c, p - c)
    one_minus = 1 - np.sqrt(max(0.0, 1 - sin2_max))
    if sin2_max < 0.00068523:
        one_minus = sin2_max / 2
    return 1 / (2.90 * np.pi * one_minus)  # reference quirk 1 (sphere.rs:456): 2.90, where sample_with_context uses 2


# ---- PixelSensor::to_sensor_rgb + RgbFilm::add_sample, float32-exact (film.rs:548-574, 907-914) ----------------------------
def dense_lookup(table, lam, lambda_min=360):  # DenselySampledSpectrum::get, spectrum.rs:227-236: round-half-away index, 0 outside
    idx = int(np.floor(abs(float(lam)) + 0.5) * (1 if lam >= 0 else -1)) - lambda_min
    return f32(0.0) if idx < 0 or idx >= len(table) else f32(table[idx])


def sensor_rgb(r_bar, g_bar, b_bar, imaging_ratio, max_component, L, lam, pdf):
    l = [f32(f32(L[i]) / f32(pdf[i])) if f32(pdf[i]) != 0 else f32(0.0) for i in range(4)]
    rgb = []
    for tab in (r_bar, g_bar, b_bar):
        prod = [f32(dense_lookup(tab, lam[i]) * l[i]) for i in range(4)]
        total = f32(0.0)
        for v in prod:  # iter().sum::<f32>() starts from 0.0 and adds left to right
            total = f32(total + v)
        rgb.append(f32(f32(total / f32(4.0)) * f32(imaging_ratio)))
    m = max(rgb)
    if m > f32(max_component):
        rgb = [f32(f32(c * f32(max_component)) / m) for c in rgb]
    return rgb


# ---- PerspectiveCamera::generate_ray_differential in float64 (camera.rs:1003-1079) ---------------------------------------
def concentric64(u):  # sampling.rs:324-339
    ox, oy = 2 * u[0] - 1, 2 * u[1] - 1
    if ox == 0 and oy == 0:
        return np.zeros(2)
    if abs(ox) > abs(oy):
        r, th = ox, np.pi / 4 * (oy / ox)
    else:
        r, th = oy, np.pi / 2 - np.pi / 4 * (ox / oy)
    return r * np.array([np.cos(th), np.sin(th)])


def camera_ray64(cam, p_film, p_lens):
    """cam: dict with camera_from_raster, render_from_camera (4x4 float64 of the float32 matrices the ABI holds), dx_camera, dy_camera,
    lens_radius, focal_distance."""
    m = cam["camera_from_raster"]
    q = m @ np.array([p_film[0], p_film[1], 0.0, 1.0])
    p_cam = q[:3] / q[3]
    d = p_cam / np.linalg.norm(p_cam)
    o = np.zeros(3)
    pl = cam["lens_radius"] * concentric64(p_lens)
    if cam["lens_radius"] > 0:
        ft = cam["focal_distance"] / d[2]
        pf = o + d * ft
        o = np.array([pl[0], pl[1], 0.0])
        d = (pf - o) / np.linalg.norm(pf - o)
    aux = []
    for dc in (cam["dx_camera"], cam["dy_camera"]):
        dd = (p_cam + dc) / np.linalg.norm(p_cam + dc)
        if cam["lens_radius"] > 0:
            pf = cam["focal_distance"] / dd[2] * dd
            ao = np.array([pl[0], pl[1], 0.0])
            ad = (pf - ao) / np.linalg.norm(pf - ao)
        else:
            ao, ad = o.copy(), dd
        aux.append((ao, ad))
    r = cam["render_from_camera"]

    def pt(x):
        h = r @ np.array([*x, 1.0])
        return h[:3] / h[3]

    def vec(x):
        return r[:3, :3] @ x
    # (Transform::apply_ray nudges the origin by its float32 error bound, transform.rs:516-531: ~1e-7 relative, inside the tolerance)
    return {"o": pt(o).tolist(), "d": vec(d).tolist(), "rx_o": pt(aux[0][0]).tolist(), "rx_d": vec(aux[0][1]).tolist(),
            "ry_o": pt(aux[1][0]).tolist(), "ry_d": vec(aux[1][1]).tolist()}


def perspective_camera64(eye, look, up, fov_deg, res, lens_radius, focal_distance):
    """PerspectiveCamera::new in float64 (camera.rs:893-963, 594-642, 848-864; transform.rs look_at / perspective), CameraWorld
    render space: render_from_camera = rotation of world_from_camera only (camera.rs:507-523)."""
    eye, look, up = (np.asarray(x, float) for x in (eye, look, up))
    direction = (look - eye) / np.linalg.norm(look - eye)
    right = np.cross(up / np.linalg.norm(up), direction)
    right /= np.linalg.norm(right)
    new_up = np.cross(direction, right)
    world_from_camera = np.eye(4)
    world_from_camera[:3, 0], world_from_camera[:3, 1], world_from_camera[:3, 2], world_from_camera[:3, 3] = right, new_up, direction, eye
    render_from_camera = world_from_camera.copy()
    render_from_camera[:3, 3] = 0.0  # CameraWorld: render space is world space translated to the camera position
    n, f = 1e-2, 1000.0
    persp = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, f / (f - n), -f * n / (f - n)], [0, 0, 1, 0]], float)
    inv_tan = 1 / np.tan(np.radians(fov_deg) / 2)
    screen_from_camera = np.diag([inv_tan, inv_tan, 1, 1]) @ persp
    aspect = res[0] / res[1]
    sw = (-aspect, aspect, -1.0, 1.0) if aspect > 1 else (-1.0, 1.0, -1 / aspect, 1 / aspect)
    ndc_from_screen = np.diag([1 / (sw[1] - sw[0]), 1 / (sw[3] - sw[2]), 1, 1]) @ np.array(
        [[1, 0, 0, -sw[0]], [0, 1, 0, -sw[3]], [0, 0, 1, 0], [0, 0, 0, 1]], float)
    raster_from_ndc = np.diag([res[0], -res[1], 1, 1])
    raster_from_screen = raster_from_ndc @ ndc_from_screen
    camera_from_raster = np.linalg.inv(screen_from_camera) @ np.linalg.inv(raster_from_screen)

    def ap(m, x):
        h = m @ np.array([*x, 1.0])
        return h[:3] / h[3]
    dx = ap(camera_from_raster, [1, 0, 0]) - ap(camera_from_raster, [0, 0, 0])
    dy = ap(camera_from_raster, [0, 1, 0]) - ap(camera_from_raster, [0, 0, 0])
    return {"world_from_camera": world_from_camera, "camera_from_raster": camera_from_raster, "render_from_camera": render_from_camera,
            "dx_camera": dx, "dy_camera": dy, "lens_radius": lens_radius, "focal_distance": focal_distance}


def main():
    rng = np.random.default_rng(20261002)
    g = {"f32": {}, "f64": {}}

    def unit(z_sign=None):
        v = rng.normal(size=3)
        v /= np.linalg.norm(v)
        if z_sign is not None:
            v[2] = z_sign * abs(v[2])
        return v.astype(np.float32)

    # rough dielectric f / pdf: reflection and transmission, both sides of the interface
    cases = []
    for eta, ax, ay in ((1.5, 0.3, 0.3), (1.33, 0.1, 0.25), (2.4, 0.5, 0.05)):
        for so in (1, -1):
            for si in (1, -1):
                for _ in range(6):
                    wo, wi = unit(so), unit(si)
                    cases.append({"eta": eta, "ax": ax, "ay": ay, "wo": fl(wo), "wi": fl(wi), "f": dielectric_f(eta, ax, ay, wo, wi),
                                  "pdf": dielectric_pdf(eta, ax, ay, wo, wi)})
    # grazing / total-internal-reflection configurations (backfacing-microfacet rejection, sin2_theta_t >= 1 -> F = 1)
    for wo, wi in (([0.9, 0.0, -0.43589], [-0.95, 0.0, 0.3122499]), ([0.6, 0.0, 0.8], [0.6, 0.0, -0.8]), ([0.0, 0.0, 1.0], [0.0, 0.0, -1.0]),
                   ([0.99, 0.0, -0.1410674], [0.99, 0.0, -0.1410674])):
        wo, wi = v32(wo), v32(wi)
        cases.append({"eta": 1.5, "ax": 0.2, "ay": 0.2, "wo": fl(wo), "wi": fl(wi), "f": dielectric_f(1.5, 0.2, 0.2, wo, wi), "pdf": dielectric_pdf(1.5, 0.2, 0.2, wo, wi)})
    g["f32"]["dielectric_rough_f_pdf"] = cases
    assert sum(1 for c in cases if c["f"] > 0) > 30 and sum(1 for c in cases if c["f"] == 0) > 5
    g["f32"]["conductor_rough_pdf"] = [{"ax": ax, "ay": ay, "wo": fl(wo), "wi": fl(wi), "pdf": conductor_pdf(ax, ay, wo, wi)}
                                       for ax, ay in ((0.1, 0.1), (0.3, 0.05)) for so in (1, -1) for wo, wi in [(unit(so), unit(so)) for _ in range(8)] + [(unit(so), unit(-so))]]

    # conductor f (complex Fresnel), sample_wm, sample_f: float64
    au_eta, au_k = [0.143, 0.375, 1.442, 1.65], [3.98, 2.38, 1.60, 1.90]
    g["f64"]["rel_tol"] = 3e-5
    g["f64"]["conductor_rough_f"] = [{"ax": 0.15, "ay": 0.3, "eta": au_eta, "k": au_k, "wo": fl(wo), "wi": fl(wi),
                                      "f": conductor_f64(f32(0.15), f32(0.3), au_eta, au_k, wo, wi)} for so in (1, -1) for wo, wi in [(unit(so), unit(so)) for _ in range(8)]]
    wm_cases = []
    for ax, ay in ((0.2, 0.2), (0.05, 0.4), (0.7, 0.7)):
        for so in (1, -1):
            for _ in range(5):
                w, u = unit(so), rng.random(2).astype(np.float32)
                wm_cases.append({"ax": ax, "ay": ay, "w": fl(w), "u": fl(u), "wm": sample_wm64(float(f32(ax)), float(f32(ay)), w, u).tolist()})
    g["f64"]["tr_sample_wm"] = wm_cases
    sf = []
    for so in (1, -1):
        for _ in range(8):
            wo, u = unit(so), rng.random(2).astype(np.float32)
            r = conductor_sample_f64(float(f32(0.15)), float(f32(0.3)), au_eta, au_k, wo, u)
            sf.append({"ax": 0.15, "ay": 0.3, "eta": au_eta, "k": au_k, "wo": fl(wo), "u": fl(u), "sample": r})
    g["f64"]["conductor_rough_sample_f"] = sf
    sd = []
    while len(sd) < 32:
        so = 1 if len(sd) % 2 == 0 else -1
        wo, u, uc = unit(so), rng.random(2).astype(np.float32), float(f32(rng.random()))
        r, margin = dielectric_sample_f64(1.5, float(f32(0.25)), float(f32(0.1)), wo, uc, u)
        if margin < 1e-3:
            continue  # too close to the reflect / transmit decision for a cross-precision comparison
        sd.append({"eta": 1.5, "ax": 0.25, "ay": 0.1, "wo": fl(wo), "uc": uc, "u": fl(u), "sample": r})
    g["f64"]["dielectric_rough_sample_f"] = sd
    assert sum(1 for c in sd if c["sample"] and c["sample"]["reflect"]) > 3 and sum(1 for c in sd if c["sample"] and not c["sample"]["reflect"]) > 8

    # triangle interaction: bare mesh, uv only, n only, n + s + uv, degenerate uv with normals, flipped orientation
    tri = []
    for k in range(18):
        p = (rng.random((3, 3)) * 4 - 2).astype(np.float32)
        bb = rng.dirichlet([1, 1, 1]).astype(np.float32)
        mode = k % 6
        uv = None if mode in (0, 2) else (rng.random((3, 2)).astype(np.float32))
        if mode == 4:
            uv = np.tile(rng.random((1, 2)).astype(np.float32), (3, 1))  # degenerate parameterisation
        nrm = None
        if mode in (2, 3, 4, 5):
            ng = np.cross(p[0] - p[2], p[1] - p[2])
            ng /= np.linalg.norm(ng)
            nrm = np.stack([(ng + 0.3 * rng.normal(size=3)) for _ in range(3)])
            nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
        s = (rng.normal(size=(3, 3)).astype(np.float32)) if mode in (3, 5) else None
        flip = mode == 5 or k == 0
        out = triangle_interaction(p, nrm, s, uv, flip, bb)
        tri.append({"p": fl(p), "n": None if nrm is None else fl(nrm), "s": None if s is None else fl(s), "uv": None if uv is None else fl(uv),
                    "flip": bool(flip), "b": fl(bb), "out": out})
    g["f32"]["triangle_interaction"] = tri

    # sphere light sampling (outside: cone sampling; one far-away small-angle case)
    sph = []
    for k in range(12):
        c = (rng.random(3) * 4 - 2).astype(np.float32)
        radius = float(f32(0.25 + rng.random()))
        far = 80.0 if k >= 10 else 1.0
        p = (c + (unit() * f32(radius * (1.3 + 3 * rng.random()) * far))).astype(np.float32)
        u = rng.random(2).astype(np.float32)
        sph.append({"center": fl(c), "radius": radius, "ctx_p": fl(p), "u": fl(u), "sample": sphere_sample64(c, radius, p, u), "pdf_with_context": sphere_pdf64(c, radius, p)})
    g["f64"]["sphere_sample_with_context"] = sph

    # area light l(): one- and two-sided, emission table = a ramp; exact
    table = (np.linspace(0.5, 2.0, 471)).astype(np.float32)
    al = []
    for two_sided in (0, 1):
        for _ in range(6):
            n, w = unit(), unit()
            lam = (360 + rng.random(4) * 470).astype(np.float32)
            scale = f32(3.25)
            lit = two_sided or dot(n, w) >= 0
            al.append({"two_sided": two_sided, "scale": float(scale), "n": fl(n), "w": fl(w), "lambda": fl(lam),
                       "l": [float(f32(scale * dense_lookup(table, x))) if lit else 0.0 for x in lam]})
    g["f32"]["area_light_l"] = {"table": fl(table), "cases": al}

    # sensor + film
    xs = np.arange(360, 831)
    r_bar = np.exp(-0.5 * ((xs - 600) / 40.0) ** 2).astype(np.float32)
    g_bar = np.exp(-0.5 * ((xs - 550) / 45.0) ** 2).astype(np.float32)
    b_bar = np.exp(-0.5 * ((xs - 450) / 30.0) ** 2).astype(np.float32)
    film = {"r_bar": fl(r_bar), "g_bar": fl(g_bar), "b_bar": fl(b_bar), "imaging_ratio": float(f32(0.0123)), "max_component_value": 0.75, "samples": []}
    px = np.zeros(4, np.float64)
    for k in range(24):
        lam = (355 + rng.random(4) * 480).astype(np.float32)  # some outside 360..830 -> sensor 0
        pdf = (0.001 + rng.random(4) * 0.004).astype(np.float32)
        if k % 5 == 0:
            pdf[1:] = 0  # terminate_secondary: safe_div -> 0
        L = (rng.random(4) * (40 if k % 7 == 0 else 2)).astype(np.float32)  # the bright ones hit the clamp
        weight = f32(1.0)
        rgb = sensor_rgb(r_bar, g_bar, b_bar, film["imaging_ratio"], film["max_component_value"], L, lam, pdf)
        for c in range(3):
            px[c] += float(f32(weight * rgb[c]))
        px[3] += float(weight)
        film["samples"].append({"L": fl(L), "lambda": fl(lam), "pdf": fl(pdf), "rgb": fl(rgb), "pixel_after": px.tolist()})
    g["f32"]["film_add_sample"] = film

    # camera: pinhole and thin lens, square and wide frames
    cams = []
    for res, lens, focal in (((64, 64), 0.0, 1e6), ((96, 54), 0.05, 3.5), ((48, 80), 0.2, 2.0)):
        cam = perspective_camera64([1.0, 2.0, -4.0], [0.2, 0.1, 0.3], [0.0, 1.0, 0.0], 37.0, res, lens, focal)
        samples = []
        for _ in range(8):
            pf = (rng.random(2) * np.array(res)).astype(np.float32)
            plens = rng.random(2).astype(np.float32)
            samples.append({"p_film": fl(pf), "p_lens": fl(plens), "ray": camera_ray64(cam, pf, plens)})
        cams.append({"world_from_camera": cam["world_from_camera"].ravel().tolist(), "fov": 37.0, "res": list(res), "lens_radius": lens, "focal_distance": focal,
                     "samples": samples})
    g["f64"]["perspective_camera"] = cams

    OUT.write_text(json.dumps(g, indent=1))
    print("wrote", OUT, {k: {n: (len(v) if hasattr(v, "__len__") else v) for n, v in d.items()} for k, d in g.items()})


if __name__ == "__main__":
    main()
