#!/usr/bin/env python3
"""An INDEPENDENT restatement of LayeredBxDF (CoatedDiffuse / CoatedConductor) in Python, written from the Rust text of the reference —
/root/reference/src/bxdf.rs:883-1620 (`tr`, `f`, `sample_f`, `pdf`, `flags`), the three interface BxDFs it walks between (DiffuseBxDF
bxdf.rs:185-266, ConductorBxDF :328-458, DielectricBxDF :517-791, with TransportMode and BxDFReflTransFlags), scattering.rs:12-260 (reflect,
refract, Fresnel, TrowbridgeReitzDistribution, Henyey-Greenstein), media.rs:8-40, sampling.rs:187-194, 310-345, 789-792, frame.rs:24-53,
vecmath/vector.rs:1034-1042, vecmath/spherical.rs — NOT from shimmer_amd/csrc/shm/bxdf.h, which it exists to check (the
largest leaf of the path was the one without an independent evaluation).

The reference seeds the three walks from OS entropy (bxdf.rs:1014, 1292, 1426: "TODO Use a seed"); this repository defines the stream instead:
PCG32 with sequence / seed hashed from the call's arguments. That DEFINITION — the hash, the stream, `(u32 >> 8) * 2^-24` clamped below one — is restated
here from DESIGN.md / shm/sampling.h as integer arithmetic (there is no reference text for it); everything the random numbers then drive follows the Rust.

Arithmetic: float64 throughout, fed with the float32 inputs and the float32 random numbers the product draws; literals that are not exactly representable
(1e-3, 1e-4, 0.99999, 0.25, 0.9, pi ...) are rounded to float32 first, as rustc does. The comparison (tests/test_layered_golden.py) is within 2e-5 relative
(+ 1e-7 absolute): the walks go through exp, sin, cos and sqrt, whose float32 results differ from float64's in the last place, and a vector whose control flow
would hinge on such a last place (a Russian-roulette draw within 1e-6 of its threshold, ...) is not emitted — the generator checks every decision's margin.

    python tests/golden/gen_golden_layered.py        # writes tests/golden/golden_layered.json (inputs + expected outputs only)
"""
import json
import math
import struct
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent


def F(x):
    """A Rust f32 literal."""
    return float(np.float32(x))


PI = F(math.pi)
INV_PI = F(1.0 / math.pi)
INV_4PI = F(1.0 / (4.0 * math.pi))
PI_OVER_2 = F(math.pi / 2.0)
PI_OVER_4 = F(math.pi / 4.0)
ONE_MINUS_EPSILON = float(np.nextafter(np.float32(1.0), np.float32(0.0)))
REFLECTION, TRANSMISSION, DIFFUSE, GLOSSY, SPECULAR = 1, 2, 4, 8, 16  # BxDFFLags, bxdf.rs:21-45
RT_REFLECTION, RT_TRANSMISSION, RT_ALL = 1, 2, 3                      # BxDFReflTransFlags
RADIANCE, IMPORTANCE = 0, 1                                            # TransportMode
M64 = (1 << 64) - 1


class Margin:
    """Records how close any data-dependent decision came to flipping; a vector with a tiny margin is dropped (module docstring)."""
    worst = math.inf

    @classmethod
    def reset(cls):
        cls.worst = math.inf

    @classmethod
    def lt(cls, a, b):
        if math.isfinite(a) and math.isfinite(b):
            cls.worst = min(cls.worst, abs(a - b) / max(1e-30, abs(a), abs(b)))
        return a < b


lt = Margin.lt


# ---------------------------------------------------------------- vectors (vecmath/*.rs)
def vec(x, y, z):
    return np.array([x, y, z], np.float64)


def dot(a, b):
    return float(a[0] * b[0] + a[1] * b[1] + a[2] * b[2])


def length_squared(v):
    return dot(v, v)


def normalize(v):
    return v / math.sqrt(length_squared(v))


def cross(a, b):
    return vec(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def same_hemisphere(w, wp):  # spherical.rs:88-90
    return w[2] * wp[2] > 0.0


def abs_cos_theta(w):
    return abs(float(w[2]))


def cos2_theta(w):
    return float(w[2] * w[2])


def sin2_theta(w):
    return max(0.0, 1.0 - cos2_theta(w))


def sin_theta(w):
    return math.sqrt(sin2_theta(w))


def tan2_theta(w):
    c = cos2_theta(w)
    return sin2_theta(w) / c if c != 0.0 else math.inf


def cos_phi(w):  # spherical.rs:57-64
    s = sin_theta(w)
    return 1.0 if s == 0.0 else min(1.0, max(-1.0, float(w[0]) / s))


def sin_phi(w):  # spherical.rs:66-73 (returns 1, not 0, at the pole: as written)
    s = sin_theta(w)
    return 1.0 if s == 0.0 else min(1.0, max(-1.0, float(w[1]) / s))


def sqr(x):
    return x * x


def safe_sqrt(x):
    return math.sqrt(max(0.0, x))


def lerp(t, a, b):  # math.rs:246-252
    return a * (1.0 - t) + b * t


def coordinate_system(v):  # vector.rs:1034-1042
    sign = math.copysign(1.0, v[2])
    a = -1.0 / (sign + v[2])
    b = v[0] * v[1] * a
    return vec(1.0 + sign * sqr(v[0]) * a, sign * b, -sign * v[0]), vec(b, sign + sqr(v[1]) * a, -v[1])


def spherical_direction(sin_t, cos_t, phi):  # spherical.rs:8-14
    s = min(1.0, max(-1.0, sin_t))
    return vec(s * math.cos(phi), s * math.sin(phi), min(1.0, max(-1.0, cos_t)))


# ---------------------------------------------------------------- scattering.rs
def reflect(wo, n):  # :12-14
    return -wo + 2.0 * dot(wo, n) * n


def refract(wi, n, eta):  # :21-45 -> (wt, etap) or None
    cos_i = dot(n, wi)
    if cos_i < 0.0:
        eta = 1.0 / eta
        cos_i = -cos_i
        n = -n
    sin2_i = max(0.0, 1.0 - sqr(cos_i))
    sin2_t = sin2_i / sqr(eta)
    if not lt(sin2_t, 1.0):
        return None
    cos_t = math.sqrt(1.0 - sin2_t)
    return -wi / eta + (cos_i / eta - cos_t) * n, eta


def fresnel_dielectric(cos_i, eta):  # :51-73
    cos_i = min(1.0, max(-1.0, cos_i))
    if cos_i < 0.0:
        eta = 1.0 / eta
        cos_i = -cos_i
    sin2_i = 1.0 - cos_i * cos_i
    sin2_t = sin2_i / (eta * eta)
    if not lt(sin2_t, 1.0):
        return 1.0
    cos_t = safe_sqrt(1.0 - sin2_t)
    r_parl = (eta * cos_i - cos_t) / (eta * cos_i + cos_t)
    r_perp = (cos_i - eta * cos_t) / (cos_i + eta * cos_t)
    return 0.5 * (r_parl * r_parl + r_perp * r_perp)


def fresnel_complex_spectral(cos_i, eta4, k4):  # :80-107
    cos_i = min(1.0, max(0.0, cos_i))
    out = np.zeros(4)
    for i in range(4):
        eta = complex(eta4[i], k4[i])
        sin2_i = 1.0 - sqr(cos_i)
        sin2_t = sin2_i / (eta * eta)
        cos_t = np.sqrt(1.0 - sin2_t + 0j)
        r_parl = (eta * cos_i - cos_t) / (eta * cos_i + cos_t)
        r_perp = (cos_i - eta * cos_t) / (cos_i + eta * cos_t)
        out[i] = (abs(r_parl) ** 2 + abs(r_perp) ** 2) / 2.0
    return out


class TrowbridgeReitz:  # :110-225
    def __init__(self, ax, ay):
        self.ax, self.ay = ax, ay
        if not self.effectively_smooth():
            self.ax, self.ay = max(ax, F(1e-4)), max(ay, F(1e-4))

    def effectively_smooth(self):
        return self.ax < F(1e-3) and self.ay < F(1e-3)

    def d(self, wm):
        t2 = tan2_theta(wm)
        if math.isinf(t2):
            return 0.0
        c4 = sqr(cos2_theta(wm))
        if c4 < F(1e-16):
            return 0.0
        e = t2 * (sqr(cos_phi(wm) / self.ax) + sqr(sin_phi(wm) / self.ay))
        return 1.0 / (PI * self.ax * self.ay * c4 * sqr(1.0 + e))

    def lam(self, w):
        t2 = tan2_theta(w)
        if math.isinf(t2):
            return 0.0
        a2 = sqr(cos_phi(w) * self.ax) + sqr(sin_phi(w) * self.ay)
        return (-1.0 + math.sqrt(1.0 + a2 * t2)) / 2.0

    def g1(self, w):
        return 1.0 / (1.0 + self.lam(w))

    def g(self, wo, wi):
        return 1.0 / (1.0 + self.lam(wo) + self.lam(wi))

    def pdf(self, w, wm):  # d_w
        return self.g1(w) / abs_cos_theta(w) * self.d(wm) * abs(dot(w, wm))

    def sample_wm(self, w, u):
        wh = normalize(vec(self.ax * w[0], self.ay * w[1], w[2]))
        if wh[2] < 0.0:
            wh = -wh
        t1 = normalize(cross(vec(0.0, 0.0, 1.0), wh)) if lt(wh[2], F(0.99999)) else vec(1.0, 0.0, 0.0)
        t2 = cross(wh, t1)
        r, theta = math.sqrt(u[0]), 2.0 * PI * u[1]  # sample_uniform_disk_polar, sampling.rs:341-345
        px, py = r * math.cos(theta), r * math.sin(theta)
        h = math.sqrt(1.0 - sqr(px))
        py = lerp((1.0 + wh[2]) / 2.0, h, py)
        pz = math.sqrt(max(0.0, 1.0 - (px * px + py * py)))
        nh = px * t1 + py * t2 + pz * wh
        return normalize(vec(self.ax * nh[0], self.ay * nh[1], max(F(1e-6), nh[2])))


def henyey_greenstein(cos_t, g):  # :231-236
    g = min(F(0.99), max(F(-0.99), g))
    denom = 1.0 + sqr(g) + 2.0 * g * cos_t
    return INV_4PI * (1.0 - sqr(g)) / (denom * safe_sqrt(denom))


def sample_henyey_greenstein(wo, g, u):  # :239-260 -> (pdf, wi)
    g = min(F(0.99), max(F(-0.99), g))
    if abs(g) < F(1e-3):
        cos_t = 1.0 - 2.0 * u[0]
    else:
        cos_t = -1.0 / (2.0 * g) * (1.0 + sqr(g) - sqr((1.0 - sqr(g)) / (1.0 + g - 2.0 * g * u[0])))
    sin_t = safe_sqrt(1.0 - sqr(cos_t))
    phi = 2.0 * PI * u[1]
    x, y = coordinate_system(wo)  # Frame::from_z, frame.rs:24-27
    l = spherical_direction(sin_t, cos_t, phi)
    wi = l[0] * x + l[1] * y + l[2] * wo  # from_local_v, frame.rs:51-53
    return henyey_greenstein(cos_t, g), wi


class HGPhase:  # media.rs:8-40: p == pdf, sample_p returns the pdf as p
    def __init__(self, g):
        self.g = g

    def p(self, wo, wi):
        return henyey_greenstein(dot(wo, wi), self.g)

    pdf = p

    def sample_p(self, wo, u):
        pdf, wi = sample_henyey_greenstein(wo, self.g, u)
        return {"p": pdf, "wi": wi, "pdf": pdf}


# ---------------------------------------------------------------- sampling.rs
def power_heuristic(nf, f_pdf, ng, g_pdf):  # :187-194 (f32 overflow of f*f cannot arise for the vectors emitted: asserted)
    f, g = nf * f_pdf, ng * g_pdf
    assert sqr(f) < 3e38
    return (f * f) / (f * f + g * g)


def sample_uniform_disk_concentric(u):  # :324-339
    ox, oy = 2.0 * u[0] - 1.0, 2.0 * u[1] - 1.0
    if ox == 0.0 and oy == 0.0:
        return 0.0, 0.0
    if abs(ox) > abs(oy):
        r, theta = ox, PI_OVER_4 * (oy / ox)
    else:
        r, theta = oy, PI_OVER_2 - PI_OVER_4 * (ox / oy)
    return r * math.cos(theta), r * math.sin(theta)


def sample_cosine_hemisphere(u):  # :310-318
    dx, dy = sample_uniform_disk_concentric(u)
    return vec(dx, dy, safe_sqrt(1.0 - sqr(dx) - sqr(dy)))


def sample_exponential(x, a):  # :789-792 — the DENSITY at x, as written
    return a * math.exp(-a * x)


# ---------------------------------------------------------------- the interface BxDFs
def sample(f, wi, pdf, flags, eta=1.0):
    return {"f": np.array(f, np.float64) * np.ones(4), "wi": wi, "pdf": pdf, "flags": flags, "eta": eta, "pdf_is_proportional": False}


class Diffuse:  # bxdf.rs:185-266
    def __init__(self, r4):
        self.r = np.array(r4, np.float64)

    def f(self, wo, wi, mode):
        return self.r * INV_PI if same_hemisphere(wo, wi) else np.zeros(4)

    def sample_f(self, wo, uc, u, mode, flags):
        if flags & RT_REFLECTION == 0:
            return None
        wi = sample_cosine_hemisphere(u)
        if wo[2] < 0.0:
            wi[2] *= -1.0
        return sample(self.r * INV_PI, wi, abs_cos_theta(wi) * INV_PI, DIFFUSE | REFLECTION)

    def pdf(self, wo, wi, mode, flags):
        return 0.0 if (flags & RT_REFLECTION == 0 or not same_hemisphere(wo, wi)) else abs_cos_theta(wi) * INV_PI

    def flags(self):
        return 0 if not self.r.any() else DIFFUSE | REFLECTION


class Conductor:  # bxdf.rs:328-458
    def __init__(self, mf, eta4, k4):
        self.mf, self.eta, self.k = mf, np.array(eta4, np.float64), np.array(k4, np.float64)

    def f(self, wo, wi, mode):
        if not same_hemisphere(wo, wi) or self.mf.effectively_smooth():
            return np.zeros(4)
        co, ci = abs_cos_theta(wo), abs_cos_theta(wi)
        if ci == 0.0 or co == 0.0:
            return np.zeros(4)
        wm = wi + wo
        if length_squared(wm) == 0.0:
            return np.zeros(4)
        wm = normalize(wm)
        fr = fresnel_complex_spectral(abs(dot(wo, wm)), self.eta, self.k)
        return self.mf.d(wm) * fr * self.mf.g(wo, wi) / (4.0 * co * ci)

    def sample_f(self, wo, uc, u, mode, flags):
        if flags & RT_REFLECTION == 0:
            return None
        if self.mf.effectively_smooth():
            wi = vec(-wo[0], -wo[1], wo[2])
            return sample(fresnel_complex_spectral(abs_cos_theta(wi), self.eta, self.k) / abs_cos_theta(wi), wi, 1.0, SPECULAR | REFLECTION)
        if wo[2] == 0.0:
            return None
        wm = self.mf.sample_wm(wo, u)
        wi = reflect(wo, wm)
        if not same_hemisphere(wo, wi):
            return None
        pdf = self.mf.pdf(wo, wm) / (4.0 * abs(dot(wo, wm)))
        co, ci = abs_cos_theta(wo), abs_cos_theta(wi)
        if ci == 0.0 or co == 0.0:
            return None
        fr = fresnel_complex_spectral(abs(dot(wo, wm)), self.eta, self.k)
        return sample(self.mf.d(wm) * fr * self.mf.g(wo, wi) / (4.0 * co * ci), wi, pdf, GLOSSY | REFLECTION)

    def pdf(self, wo, wi, mode, flags):
        if flags & RT_REFLECTION == 0 or not same_hemisphere(wo, wi) or self.mf.effectively_smooth():
            return 0.0
        wm = wo + wi
        if length_squared(wm) == 0.0:
            return 0.0
        wm = normalize(wm)
        if wm[2] < 0.0:  # face_forward_n(Normal3f::Z)
            wm = -wm
        return self.mf.pdf(wo, wm) / (4.0 * abs(dot(wo, wm)))

    def flags(self):
        return (SPECULAR if self.mf.effectively_smooth() else GLOSSY) | REFLECTION


class Dielectric:  # bxdf.rs:517-791
    def __init__(self, eta, mf):
        self.eta, self.mf = eta, mf

    def _half(self, wo, wi):
        co, ci = float(wo[2]), float(wi[2])
        refl = ci * co > 0.0
        etap = 1.0 if refl else (self.eta if co > 0.0 else 1.0 / self.eta)
        wm = wi * etap + wo
        if ci == 0.0 or co == 0.0 or length_squared(wm) == 0.0:
            return None
        wm = normalize(wm)
        if wm[2] < 0.0:
            wm = -wm
        if dot(wm, wi) * ci < 0.0 or dot(wm, wo) * co < 0.0:  # backfacing microfacets
            return None
        return co, ci, refl, etap, wm

    def f(self, wo, wi, mode):
        if self.eta == 1.0 or self.mf.effectively_smooth():
            return np.zeros(4)
        h = self._half(wo, wi)
        if h is None:
            return np.zeros(4)
        co, ci, refl, etap, wm = h
        fr = fresnel_dielectric(dot(wo, wm), self.eta)
        if refl:
            return np.ones(4) * (self.mf.d(wm) * self.mf.g(wo, wi) * fr / abs(4.0 * ci * co))
        denom = sqr(dot(wi, wm) + dot(wo, wm) / etap) * ci * co
        ft = self.mf.d(wm) * (1.0 - fr) * self.mf.g(wo, wi) * abs(dot(wi, wm) * dot(wo, wm) / denom)
        if mode == RADIANCE:
            ft /= sqr(etap)
        return np.ones(4) * ft

    def sample_f(self, wo, uc, u, mode, flags):
        smooth = self.eta == 1.0 or self.mf.effectively_smooth()
        wm = vec(0.0, 0.0, 1.0) if smooth else self.mf.sample_wm(wo, u)
        r = fresnel_dielectric(float(wo[2]) if smooth else dot(wo, wm), self.eta)
        t = 1.0 - r
        pr, pt = r, t
        if flags & RT_REFLECTION == 0:
            pr = 0.0
        if flags & RT_TRANSMISSION == 0:
            pt = 0.0
        if pr == 0.0 and pt == 0.0:
            return None
        if lt(uc, pr / (pr + pt)):
            if smooth:
                wi = vec(-wo[0], -wo[1], wo[2])
                return sample(r / abs_cos_theta(wi), wi, pr / (pr + pt), SPECULAR | REFLECTION)
            wi = reflect(wo, wm)
            if not same_hemisphere(wo, wi):
                return None
            pdf = self.mf.pdf(wo, wm) / (4.0 * abs(dot(wo, wm))) * pr / (pr + pt)
            return sample(self.mf.d(wm) * self.mf.g(wo, wi) * r / (4.0 * float(wi[2]) * float(wo[2])), wi, pdf, GLOSSY | REFLECTION)
        rf = refract(wo, wm, self.eta)
        if rf is None:
            return None
        wi, etap = rf
        if smooth:
            ft = t / abs_cos_theta(wi)
            if mode == RADIANCE:
                ft /= sqr(etap)
            return sample(ft, wi, pt / (pr + pt), SPECULAR | TRANSMISSION, etap)
        if same_hemisphere(wo, wi) or wi[2] == 0.0:
            return None
        denom = sqr(dot(wi, wm) + dot(wo, wm) / etap)
        pdf = self.mf.pdf(wo, wm) * (abs(dot(wi, wm)) / denom) * pt / (pr + pt)
        ft = t * self.mf.d(wm) * self.mf.g(wo, wi) * abs(dot(wi, wm) * dot(wo, wm) / (float(wi[2]) * float(wo[2]) * denom))
        if mode == RADIANCE:
            ft /= sqr(etap)
        return sample(ft, wi, pdf, GLOSSY | TRANSMISSION, etap)

    def pdf(self, wo, wi, mode, flags):
        if self.eta == 1.0 or self.mf.effectively_smooth():
            return 0.0
        h = self._half(wo, wi)
        if h is None:
            return 0.0
        co, ci, refl, etap, wm = h
        r = fresnel_dielectric(dot(wo, wm), self.eta)
        pr, pt = r, 1.0 - r
        if flags & RT_REFLECTION == 0:
            pr = 0.0
        if flags & RT_TRANSMISSION == 0:
            pt = 0.0
        if pr == 0.0 and pt == 0.0:
            return 0.0
        if refl:
            return self.mf.pdf(wo, wm) / (4.0 * abs(dot(wo, wm))) * pr / (pr + pt)
        denom = sqr(dot(wi, wm) + dot(wo, wm) / etap)
        return self.mf.pdf(wo, wm) * (abs(dot(wi, wm)) / denom) * pt / (pr + pt)

    def flags(self):
        return (TRANSMISSION if self.eta == 1.0 else REFLECTION | TRANSMISSION) | (SPECULAR if self.mf.effectively_smooth() else GLOSSY)


# ---------------------------------------------------------------- this repository's defined random stream (no reference text exists)
def f32_bits(x):
    return struct.unpack("<I", struct.pack("<f", float(np.float32(x))))[0]


def mix_bits(v):
    v ^= v >> 31
    v = (v * 0x7FB5D329728EA185) & M64
    v ^= v >> 27
    v = (v * 0x81DADEF4BC2DD44D) & M64
    v ^= v >> 33
    return v


def hash_f32(h, x):
    return mix_bits(h ^ ((f32_bits(x) + 0x9E3779B97F4A7C15) & M64))


def hash_v3(h, v):
    return hash_f32(hash_f32(hash_f32(h, v[0]), v[1]), v[2])


class Pcg32:
    MULT = 0x5851F42D4C957F2D

    def __init__(self, sequence, seed):
        self.state, self.inc = 0, ((sequence << 1) | 1) & M64
        self.next_u32()
        self.state = (self.state + seed) & M64
        self.next_u32()

    def next_u32(self):
        old = self.state
        self.state = (old * self.MULT + self.inc) & M64
        x = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((x >> rot) | (x << ((-rot) & 31))) & 0xFFFFFFFF

    def r(self):  # the closure `r` of bxdf.rs:1015-1020 over rand's f32 mapping: (u32 >> 8) * 2^-24, min(v, next_float_down(1.0))
        return min((self.next_u32() >> 8) * 2.0 ** -24, ONE_MINUS_EPSILON)

    def r2(self):
        a = self.r()
        return (a, self.r())


# ---------------------------------------------------------------- LayeredBxDF<Top, Bottom, TWO_SIDED = true>, bxdf.rs:883-1620
def unusable(s):
    return s is None or not s["f"].any() or s["pdf"] == 0.0 or s["wi"][2] == 0.0


class Layered:
    def __init__(self, top, bottom, thickness, albedo4, g, max_depth, n_samples):
        self.top, self.bottom, self.thickness, self.albedo, self.g = top, bottom, thickness, np.array(albedo4, np.float64), g
        self.max_depth, self.n_samples = max_depth, n_samples

    def tr(self, dz, w):  # :925-933 (`Float::MIN` is the most negative float: the early-out never fires)
        return math.exp(-abs(dz / w[2]))

    def flags(self):  # :1577-1603
        tf, bf = self.top.flags(), self.bottom.flags()
        fl = REFLECTION
        if tf & SPECULAR:
            fl |= SPECULAR
        if (tf & DIFFUSE) or (bf & DIFFUSE) or self.albedo.any():
            fl |= DIFFUSE
        elif (tf & GLOSSY) or (bf & GLOSSY):
            fl |= GLOSSY
        if (tf & TRANSMISSION) and (bf & TRANSMISSION):
            fl |= TRANSMISSION
        return fl

    def f(self, wo, wi, mode):  # :941-1218
        f = np.zeros(4)
        if wo[2] < 0.0:  # TWO_SIDED
            wo, wi = -wo, -wi
        entered_top = True
        enter_i = self.top
        exit_is_bottom = same_hemisphere(wo, wi) ^ entered_top
        exit_i, non_exit_i = (self.bottom, self.top) if exit_is_bottom else (self.top, self.bottom)
        exit_z = 0.0 if exit_is_bottom else self.thickness
        if same_hemisphere(wo, wi):
            f = enter_i.f(wo, wi, mode) * float(self.n_samples)
        rng = Pcg32(hash_v3(0x5EED0001, wi), hash_v3(0, wo))
        wis_mode = IMPORTANCE if mode == RADIANCE else RADIANCE
        for _s in range(self.n_samples):
            uc = rng.r()
            wos = enter_i.sample_f(wo, uc, rng.r2(), mode, RT_TRANSMISSION)
            if unusable(wos):
                continue
            uc = rng.r()
            wis = exit_i.sample_f(wi, uc, rng.r2(), wis_mode, RT_TRANSMISSION)
            if unusable(wis):
                continue
            beta = wos["f"] * abs_cos_theta(wos["wi"]) / wos["pdf"]
            z = self.thickness if entered_top else 0.0
            w = wos["wi"]
            phase = HGPhase(self.g)
            for depth in range(self.max_depth):
                if depth > 3 and lt(beta.max(), F(0.25)):
                    q = max(0.0, 1.0 - beta.max())
                    if lt(rng.r(), q):
                        break
                    beta = beta / (1.0 - q)
                if not self.albedo.any():
                    z = 0.0 if z == self.thickness else self.thickness
                    beta = beta * self.tr(self.thickness, w)
                else:
                    dz = sample_exponential(rng.r(), 1.0 / abs(float(w[2])))
                    zp = z + dz if w[2] > 0.0 else z - dz
                    if z == zp:
                        continue
                    if lt(0.0, zp) and lt(zp, self.thickness):
                        wt = 1.0
                        if not (exit_i.flags() & SPECULAR):
                            wt = power_heuristic(1, wis["pdf"], 1, phase.pdf(-w, -wis["wi"]))
                        f = f + beta * self.albedo * phase.p(-w, -wis["wi"]) * wt * self.tr(zp - exit_z, wis["wi"]) * wis["f"] / wis["pdf"]
                        ps = phase.sample_p(-w, rng.r2())
                        if ps["pdf"] == 0.0 or ps["wi"][2] == 0.0:
                            continue
                        beta = beta * (self.albedo * ps["p"] / ps["pdf"])
                        w = ps["wi"]
                        z = zp
                        if ((z < exit_z and w[2] > 0.0) or (z > exit_z and w[2] < 0.0)) and not (exit_i.flags() & SPECULAR):
                            f_exit = exit_i.f(-w, wi, mode)
                            if f_exit.any():
                                exit_pdf = exit_i.pdf(-w, wi, mode, RT_TRANSMISSION)
                                f = f + beta * self.tr(zp - exit_z, ps["wi"]) * f_exit * power_heuristic(1, ps["pdf"], 1, exit_pdf)
                        continue
                    z = min(self.thickness, max(0.0, zp))
                if z == exit_z:
                    uc = rng.r()
                    bs = exit_i.sample_f(-w, uc, rng.r2(), mode, RT_REFLECTION)
                    if unusable(bs):
                        break
                    beta = beta * (bs["f"] * abs_cos_theta(bs["wi"]) / bs["pdf"])
                    w = bs["wi"]
                else:
                    if not (non_exit_i.flags() & SPECULAR):
                        wt = 1.0
                        if not (exit_i.flags() & SPECULAR):
                            wt = power_heuristic(1, wis["pdf"], 1, non_exit_i.pdf(-w, -wis["wi"], mode, RT_ALL))
                        f = f + beta * non_exit_i.f(-w, -wis["wi"], mode) * abs_cos_theta(wis["wi"]) * wt * self.tr(self.thickness, wis["wi"]) * wis["f"] / wis["pdf"]
                    uc = rng.r()
                    u = rng.r2()
                    bs = non_exit_i.sample_f(-w, uc, u, mode, RT_REFLECTION)
                    if unusable(bs):
                        break
                    beta = beta * (bs["f"] * abs_cos_theta(bs["wi"]) / bs["pdf"])
                    w = bs["wi"]
                    if not (exit_i.flags() & SPECULAR):
                        f_exit = exit_i.f(-w, wi, mode)
                        if f_exit.any():
                            wt = 1.0
                            if not (non_exit_i.flags() & SPECULAR):
                                wt = power_heuristic(1, bs["pdf"], 1, exit_i.pdf(-w, wi, mode, RT_TRANSMISSION))
                            f = f + beta * self.tr(self.thickness, bs["wi"]) * f_exit * wt
        return f / float(self.n_samples)

    def sample_f(self, wo, uc, u, mode):  # :1220-1404 (sample_flags = ALL)
        flip_wi = False
        if wo[2] < 0.0:
            wo, flip_wi = -wo, True
        entered_top = True
        bs = self.top.sample_f(wo, uc, u, mode, RT_ALL)
        if unusable(bs):
            return None
        if bs["flags"] & REFLECTION:
            if flip_wi:
                bs["wi"] = -bs["wi"]
            bs["pdf_is_proportional"] = True
            return bs
        w = bs["wi"]
        specular_path = bool(bs["flags"] & SPECULAR)
        rng = Pcg32(hash_f32(hash_f32(hash_f32(0x5EED0002, uc), u[0]), u[1]), hash_v3(0, wo))
        f = bs["f"] * abs_cos_theta(bs["wi"])
        pdf = bs["pdf"]
        z = self.thickness if entered_top else 0.0
        phase = HGPhase(self.g)
        for depth in range(self.max_depth):
            rr_beta = f.max() / pdf
            if depth > 3 and lt(rr_beta, F(0.25)):
                q = max(0.0, 1.0 - rr_beta)
                if lt(rng.r(), q):
                    return None
                pdf *= 1.0 - q
            if w[2] == 0.0:
                return None
            if self.albedo.any():
                dz = sample_exponential(rng.r(), 1.0 / abs_cos_theta(w))
                zp = z + dz if w[2] > 0.0 else z - dz
                if zp == z:
                    return None
                if lt(0.0, zp) and lt(zp, self.thickness):
                    ps = phase.sample_p(-w, rng.r2())
                    if ps["pdf"] == 0.0 or ps["wi"][2] == 0.0:
                        return None
                    f = f * (self.albedo * ps["p"])
                    pdf *= ps["pdf"]
                    specular_path = False
                    w = ps["wi"]
                    z = zp
                    continue
                z = min(self.thickness, max(0.0, zp))
            else:
                z = 0.0 if z == self.thickness else self.thickness
                f = f * self.tr(self.thickness, w)
            iface = self.bottom if z == 0.0 else self.top
            uc2 = rng.r()
            u2 = rng.r2()
            bs = iface.sample_f(-w, uc2, u2, mode, RT_ALL)
            if unusable(bs):
                return None
            f = f * bs["f"]
            pdf *= bs["pdf"]
            specular_path = specular_path and bool(bs["flags"] & SPECULAR)
            w = bs["wi"]
            if bs["flags"] & TRANSMISSION:
                flags = (REFLECTION if same_hemisphere(wo, w) else TRANSMISSION) | (SPECULAR if specular_path else GLOSSY)
                if flip_wi:
                    w = -w
                return {"f": f, "wi": w, "pdf": pdf, "flags": flags, "eta": 1.0, "pdf_is_proportional": True}
            f = f * abs_cos_theta(bs["wi"])
        return None

    def pdf(self, wo, wi, mode):  # :1406-1575 (sample_flags = ALL), reference behaviour: `rs` is used without testing its pdf
        if wo[2] < 0.0:
            wo, wi = -wo, -wi
        rng = Pcg32(hash_v3(0x5EED0003, wo), hash_v3(0, wi))
        entered_top = True
        pdf_sum = 0.0
        if same_hemisphere(wo, wi):
            pdf_sum += float(self.n_samples) * self.top.pdf(wo, wi, mode, RT_REFLECTION)
        wis_mode = IMPORTANCE if mode == RADIANCE else RADIANCE
        for _s in range(self.n_samples):
            if same_hemisphere(wo, wi):
                r_i, t_i = self.bottom, self.top
                wos = t_i.sample_f(wo, rng.r(), rng.r2(), mode, RT_TRANSMISSION)
                wis = t_i.sample_f(wi, rng.r(), rng.r2(), wis_mode, RT_TRANSMISSION)
                if wos is not None and wis is not None and wos["f"].any() and wos["pdf"] > 0.0 and wis["f"].any() and wis["pdf"] > 0.0:
                    if not (t_i.flags() & (DIFFUSE | GLOSSY)):
                        pdf_sum += r_i.pdf(-wos["wi"], -wis["wi"], mode, RT_ALL)
                    else:
                        rs = r_i.sample_f(-wos["wi"], rng.r(), rng.r2(), mode, RT_ALL)
                        if rs is not None:
                            if not (r_i.flags() & (DIFFUSE | GLOSSY)):
                                pdf_sum += t_i.pdf(-rs["wi"], wi, mode, RT_ALL)
                            else:
                                r_pdf = r_i.pdf(-wos["wi"], -wis["wi"], mode, RT_ALL)
                                pdf_sum += power_heuristic(1, wis["pdf"], 1, r_pdf) * r_pdf
                                t_pdf = t_i.pdf(-rs["wi"], wi, mode, RT_ALL)
                                pdf_sum += power_heuristic(1, rs["pdf"], 1, t_pdf) * t_pdf
            else:
                to_i, ti_i = self.top, self.bottom
                uc = rng.r()
                u = rng.r2()
                wos = to_i.sample_f(wo, uc, u, mode, RT_ALL)
                if unusable(wos) or (wos["flags"] & REFLECTION):
                    continue
                uc = rng.r()
                u = rng.r2()
                wis = ti_i.sample_f(wi, uc, u, wis_mode, RT_ALL)
                if unusable(wis) or (wis["flags"] & REFLECTION):
                    continue
                if to_i.flags() & SPECULAR:
                    pdf_sum += ti_i.pdf(-wos["wi"], wi, mode, RT_ALL)
                elif ti_i.flags() & SPECULAR:
                    pdf_sum += to_i.pdf(wo, -wis["wi"], mode, RT_ALL)
                else:
                    pdf_sum += (to_i.pdf(wo, -wis["wi"], mode, RT_ALL) + ti_i.pdf(-wos["wi"], wi, mode, RT_ALL)) / 2.0
        return lerp(F(0.9), 1.0 / (4.0 * PI), pdf_sum / float(self.n_samples))


# ---------------------------------------------------------------- vectors
COATED_DIFFUSE, COATED_CONDUCTOR = 4, 5  # SHM_MATERIAL_* (include/shimmer_hip.h); the oracle's test entry takes (kind, 19 floats, 2 ints)


def make(kind, p, max_depth, n_samples):
    """The oracle entry's parameter block -> the reference's objects (material.rs:917-963, 1189-1256 build them from a material; here the
    parameters are given directly): r[4] | k[4] | albedo[4] | eta | ax ay | ax2 ay2 | thickness | g."""
    p = [float(np.float32(x)) for x in p]
    top = Dielectric(p[12], TrowbridgeReitz(p[13], p[14]))
    bottom = Diffuse(p[0:4]) if kind == COATED_DIFFUSE else Conductor(TrowbridgeReitz(p[15], p[16]), p[0:4], p[4:8])
    return Layered(top, bottom, p[17], p[8:12], p[18], max_depth, n_samples)


def unit(theta, phi):
    return np.array([math.sin(theta) * math.cos(phi), math.sin(theta) * math.sin(phi), math.cos(theta)], np.float32)


CONFIGS = [  # name, kind, r, k, albedo, eta, (ax, ay), (ax2, ay2), thickness, g, max_depth, n_samples
    ("cd_smooth", COATED_DIFFUSE, [0.5] * 4, [0] * 4, [0] * 4, 1.5, (0.0, 0.0), (0.0, 0.0), 0.01, 0.0, 10, 1),
    ("cd_rough", COATED_DIFFUSE, [0.7, 0.5, 0.3, 0.6], [0] * 4, [0] * 4, 1.5, (0.3, 0.3), (0.0, 0.0), 0.01, 0.0, 10, 2),
    ("cd_rough_aniso_medium", COATED_DIFFUSE, [0.7, 0.5, 0.3, 0.6], [0] * 4, [0.6, 0.7, 0.8, 0.5], 1.33, (0.2, 0.4), (0.0, 0.0), 0.05, 0.3, 10, 2),
    ("cd_smooth_medium_back", COATED_DIFFUSE, [0.4] * 4, [0] * 4, [0.8] * 4, 1.5, (0.0, 0.0), (0.0, 0.0), 0.1, -0.4, 12, 3),
    ("cc_smooth_rough_metal", COATED_CONDUCTOR, [0.2, 0.4, 1.1, 0.6], [3.9, 2.4, 2.2, 3.0], [0] * 4, 1.5, (0.0, 0.0), (0.25, 0.25), 0.01, 0.0, 10, 2),
    ("cc_rough_rough", COATED_CONDUCTOR, [0.2, 0.4, 1.1, 0.6], [3.9, 2.4, 2.2, 3.0], [0] * 4, 1.5, (0.15, 0.15), (0.3, 0.2), 0.02, 0.0, 10, 2),
    ("cc_rough_smooth_metal_medium", COATED_CONDUCTOR, [0.2, 0.4, 1.1, 0.6], [3.9, 2.4, 2.2, 3.0], [0.5] * 4, 1.5, (0.2, 0.2), (0.0, 0.0), 0.05, 0.2, 10, 2),
    ("cc_smooth_smooth", COATED_CONDUCTOR, [0.2, 0.4, 1.1, 0.6], [3.9, 2.4, 2.2, 3.0], [0] * 4, 1.5, (0.0, 0.0), (0.0, 0.0), 0.01, 0.0, 10, 1),
]
MIN_MARGIN = 1e-4  # a decision closer than this (relative) to its threshold could go the other way in float32: such a vector is not emitted


def main():
    rng = np.random.default_rng(20260403)
    out = {"generator": "tests/golden/gen_golden_layered.py", "tolerance": {"rel": 2e-5, "abs": 1e-7}, "f_pdf": [], "sample_f": []}
    dropped = 0
    for name, kind, r, k, albedo, eta, a1, a2, thick, g, max_depth, n_samples in CONFIGS:
        p = [*r, *k, *albedo, eta, a1[0], a1[1], a2[0], a2[1], thick, g]
        lay = make(kind, p, max_depth, n_samples)
        n_fp = n_sf = 0
        while n_fp < 10 or n_sf < 6:
            # both hemispheres for wo and wi (TWO_SIDED), grazing directions included
            up = rng.random() < 0.5
            wo = unit(rng.uniform(0.05, 1.5) if up else rng.uniform(1.65, 3.1), rng.uniform(0, 2 * math.pi))
            same = rng.random() < 0.75  # (f is zero by construction across the shading plane: three vectors in four are on the side where the walks run)
            wi = unit(rng.uniform(0.05, 1.5) if up == same else rng.uniform(1.65, 3.1), rng.uniform(0, 2 * math.pi))
            wo64, wi64 = wo.astype(np.float64), wi.astype(np.float64)
            if n_fp < 10:
                Margin.reset()
                f = lay.f(wo64, wi64, RADIANCE)
                m_f = Margin.worst
                Margin.reset()
                pdf = lay.pdf(wo64, wi64, RADIANCE)
                if min(m_f, Margin.worst) < MIN_MARGIN:
                    dropped += 1
                else:
                    out["f_pdf"].append({"config": name, "kind": kind, "p": [float(np.float32(x)) for x in p], "max_depth": max_depth, "n_samples": n_samples,
                                         "wo": [float(x) for x in wo], "wi": [float(x) for x in wi], "f": [float(x) for x in f], "pdf": float(pdf), "flags": lay.flags()})
                    n_fp += 1
            if n_sf < 6:
                uc, u = np.float32(rng.random()), rng.random(2).astype(np.float32)
                Margin.reset()
                s = lay.sample_f(wo64, float(uc), (float(u[0]), float(u[1])), RADIANCE)
                if Margin.worst < MIN_MARGIN:
                    dropped += 1
                    continue
                rec = {"config": name, "kind": kind, "p": [float(np.float32(x)) for x in p], "max_depth": max_depth, "n_samples": n_samples,
                       "wo": [float(x) for x in wo], "uc": float(uc), "u": [float(x) for x in u], "sample": None}
                if s is not None:
                    rec["sample"] = {"f": [float(x) for x in s["f"]], "wi": [float(x) for x in s["wi"]], "pdf": float(s["pdf"]), "flags": int(s["flags"]),
                                     "pdf_is_proportional": bool(s["pdf_is_proportional"])}
                out["sample_f"].append(rec)
                n_sf += 1
    out["dropped_for_margin"] = dropped
    (HERE / "golden_layered.json").write_text(json.dumps(out, indent=0))
    n_none = sum(1 for r in out["sample_f"] if r["sample"] is None)
    n_walk = sum(1 for r in out["sample_f"] if r["sample"] is not None and r["sample"]["flags"] & GLOSSY and not (r["sample"]["flags"] & 0))
    print(f"{len(out['f_pdf'])} f / pdf vectors, {len(out['sample_f'])} sample_f vectors ({n_none} None), {dropped} dropped for a decision margin below {MIN_MARGIN}")


if __name__ == "__main__":
    main()
