"""Physical properties of the single-layer BxDFs (bxdf.rs:184-267 Diffuse, 328-458 Conductor, 518-795 Dielectric) through the oracle's leaf entry points
— checks that depend on NO reading of the Rust text (tests/test_leaf_golden.py pins the same leaves bit for bit against the builder's numpy reading of it):

  * the density sample_f reports is the density pdf() evaluates at the direction it returns, and so is f;
  * pdf() integrates to at most 1 over the sphere (what is missing is the mass of rejected microfacet samples), and close to it;
  * directions drawn by sample_f are distributed as pdf() says (counts in solid-angle bins against the quadrature of pdf over the bins; draws that return None count
    as draws: the reference's pdf is the density per DRAW);
  * reflection is reciprocal, f(wo, wi) = f(wi, wo);
  * a conductor reflects at most what arrives (albedo <= 1), a diffuse surface exactly its reflectance, and the estimator f cos / pdf averages to the quadrature's albedo.

The same leaves run on the device (tests/test_gpu_leaf_replay.py) and in every film compared with the oracle."""
import ctypes as C
import math

import numpy as np
import pytest

import oracle_py
from shimmer_amd import abi

F = C.c_float
KIND_DIFFUSE, KIND_CONDUCTOR, KIND_DIELECTRIC = abi.SHM_MATERIAL_DIFFUSE, abi.SHM_MATERIAL_CONDUCTOR, abi.SHM_MATERIAL_DIELECTRIC
GOLD_ETA, GOLD_K = [0.2, 0.4, 1.4, 1.6], [3.9, 2.4, 1.6, 1.9]
CASES = {
    "diffuse": dict(kind=KIND_DIFFUSE, r=[0.1, 0.4, 0.7, 1.0], k=[0] * 4, eta=1.0, ax=0.0, ay=0.0),
    "conductor_rough": dict(kind=KIND_CONDUCTOR, r=GOLD_ETA, k=GOLD_K, eta=1.0, ax=0.3, ay=0.3),
    "conductor_anisotropic": dict(kind=KIND_CONDUCTOR, r=GOLD_ETA, k=GOLD_K, eta=1.0, ax=0.15, ay=0.5),
    "dielectric_rough": dict(kind=KIND_DIELECTRIC, r=[0] * 4, k=[0] * 4, eta=1.5, ax=0.3, ay=0.3),
    "dielectric_rough_from_inside": dict(kind=KIND_DIELECTRIC, r=[0] * 4, k=[0] * 4, eta=1.5, ax=0.25, ay=0.4, inside=True),
}


@pytest.fixture(scope="module")
def olib():
    return oracle_py.load()


def fa(v):
    v = np.asarray(v, np.float32).ravel()
    return (F * len(v))(*[float(x) for x in v])


class Bx:
    def __init__(self, olib, kind, r, k, eta, ax, ay, inside=False):
        self.olib, self.kind, self.r, self.k, self.eta, self.ax, self.ay = olib, kind, fa(r), fa(k), eta, ax, ay
        self.wo = np.array([0.48, -0.31, -0.82 if inside else 0.82], np.float32)
        self.wo /= np.float32(np.linalg.norm(self.wo))

    def f_pdf(self, wo, wi):
        out = (F * 5)()
        self.olib.orc_fn_bxdf_f_pdf(self.kind, self.r, self.k, self.eta, self.ax, self.ay, fa(wo), fa(wi), out)
        return np.array(out[:4], np.float64), float(out[4])

    def sample_f(self, wo, uc, u):
        out = (F * 10)()
        if not self.olib.orc_fn_bxdf_sample_f(self.kind, self.r, self.k, self.eta, self.ax, self.ay, fa(wo), uc, fa(u), out):
            return None
        return np.array(out[:4], np.float64), np.array(out[4:7], np.float32), float(out[7]), int(out[8])


# a midpoint grid over the sphere in (cos theta, phi): cells of equal solid angle 4 pi / (NZ * NPHI)
NZ, NPHI = 160, 320
BZ, BPHI = 8, 8  # the bins of the distribution test: blocks of 20 x 40 cells


def sphere_grid():
    z = -1.0 + (np.arange(NZ) + 0.5) * (2.0 / NZ)
    phi = (np.arange(NPHI) + 0.5) * (2.0 * math.pi / NPHI)
    s = np.sqrt(1.0 - z * z)
    return z, phi, s


def quadrature(bx):
    """pdf and f |cos| at every cell centre: (pdf[NZ, NPHI], fcos[NZ, NPHI, 4])."""
    z, phi, s = sphere_grid()
    pdf = np.zeros((NZ, NPHI))
    fcos = np.zeros((NZ, NPHI, 4))
    for i in range(NZ):
        for j in range(NPHI):
            wi = (s[i] * math.cos(phi[j]), s[i] * math.sin(phi[j]), z[i])
            f, p = bx.f_pdf(bx.wo, wi)
            pdf[i, j] = p
            fcos[i, j] = f * abs(z[i])
    return pdf, fcos


@pytest.fixture(scope="module")
def quad(olib):
    cache = {}

    def get(name):
        if name not in cache:
            kw = dict(CASES[name])
            bx = Bx(olib, **kw)
            cache[name] = (bx,) + quadrature(bx)
        return cache[name]
    return get


@pytest.mark.parametrize("name", list(CASES))
def test_pdf_integrates_to_at_most_one(quad, name):
    bx, pdf, _ = quad(name)
    cell = 4.0 * math.pi / (NZ * NPHI)
    total = pdf.sum() * cell
    assert np.all(pdf >= 0) and np.all(np.isfinite(pdf))
    assert total <= 1.0 + 5e-3, total
    # (what is missing: visible-normal samples whose reflection / refraction lands on the wrong side are rejected, bxdf.rs:396-399, 667-669, 689-694)
    assert total >= (0.999 if name == "diffuse" else 0.85), total


@pytest.mark.parametrize("name", list(CASES))
def test_sample_f_reports_the_density_and_value_of_its_own_direction(olib, name):
    bx = Bx(olib, **CASES[name])
    rng = np.random.default_rng(5)
    n_some = 0
    for _ in range(400):
        s = bx.sample_f(bx.wo, float(rng.random()), rng.random(2))
        if s is None:
            continue
        n_some += 1
        f_s, wi, pdf_s, flags = s
        assert abs(float(np.linalg.norm(wi.astype(np.float64))) - 1.0) < 1e-5
        f_e, pdf_e = bx.f_pdf(bx.wo, wi)
        assert pdf_s > 0 and abs(pdf_s - pdf_e) <= 2e-3 * max(pdf_s, pdf_e), (wi, pdf_s, pdf_e)  # (wi is rounded to f32 in between: the half vector is recomputed from it)
        assert np.all(np.abs(f_s - f_e) <= 2e-3 * np.maximum(np.abs(f_s), np.abs(f_e)) + 1e-7), (wi, f_s, f_e)
        reflected = (wi[2] > 0) == (bx.wo[2] > 0)
        assert bool(flags & 1) == reflected and bool(flags & 2) == (not reflected)  # BxDFFlags::REFLECTION = 1, TRANSMISSION = 2 (bxdf.rs:100-133)
    assert n_some > 300


@pytest.mark.parametrize("name", list(CASES))
def test_sampled_directions_follow_the_pdf(quad, name):
    bx, pdf, _ = quad(name)
    cell = 4.0 * math.pi / (NZ * NPHI)
    expect = pdf.reshape(BZ, NZ // BZ, BPHI, NPHI // BPHI).sum(axis=(1, 3)) * cell  # probability per draw of each bin
    rng = np.random.default_rng(9)
    n = 24000
    counts = np.zeros((BZ, BPHI))
    for _ in range(n):
        s = bx.sample_f(bx.wo, float(rng.random()), rng.random(2))
        if s is None:
            continue
        wi = s[1].astype(np.float64)
        iz = min(BZ - 1, int((wi[2] + 1.0) * 0.5 * BZ))
        ip = min(BPHI - 1, int((math.atan2(wi[1], wi[0]) % (2.0 * math.pi)) / (2.0 * math.pi) * BPHI))
        counts[iz, ip] += 1
    got = counts / n
    sigma = np.sqrt(np.maximum(expect, 1e-6) / n)
    # 5 sigma of the counting noise + the grid's own error on a peaked density (2 % of the bin) + a floor
    tol = 5.0 * sigma + 0.02 * expect + 5e-4
    worst = np.max(np.abs(got - expect) / tol)
    assert worst < 1.0, (worst, got.round(4).tolist(), expect.round(4).tolist())
    assert abs(got.sum() - expect.sum()) < 0.02


@pytest.mark.parametrize("name", ["diffuse", "conductor_rough", "conductor_anisotropic", "dielectric_rough"])
def test_reflection_is_reciprocal(olib, name):
    bx = Bx(olib, **CASES[name])
    rng = np.random.default_rng(3)
    n_pos = 0
    for _ in range(300):
        def up():
            z, phi = 0.05 + 0.95 * rng.random(), 2.0 * math.pi * rng.random()
            return np.array([math.sqrt(1 - z * z) * math.cos(phi), math.sqrt(1 - z * z) * math.sin(phi), z], np.float32)
        a, b = up(), up()
        fab, _ = bx.f_pdf(a, b)
        fba, _ = bx.f_pdf(b, a)
        assert np.all(np.abs(fab - fba) <= 2e-5 * np.maximum(fab, fba) + 1e-9), (a, b, fab, fba)
        n_pos += bool(np.any(fab > 0))
    assert n_pos > 250


@pytest.mark.parametrize("name", ["diffuse", "conductor_rough", "conductor_anisotropic"])
def test_albedo_is_bounded_and_the_estimator_agrees(quad, name):
    bx, _, fcos = quad(name)
    cell = 4.0 * math.pi / (NZ * NPHI)
    albedo = fcos.sum(axis=(0, 1)) * cell
    assert np.all(albedo <= 1.0 + 5e-3) and np.all(albedo > 0.02), albedo
    if name == "diffuse":
        assert np.allclose(albedo, [0.1, 0.4, 0.7, 1.0], rtol=2e-3)  # f = R / pi (bxdf.rs:196-203): the cosine-weighted integral is R
    rng = np.random.default_rng(17)
    n, acc = 8000, np.zeros(4)
    for _ in range(n):
        s = bx.sample_f(bx.wo, float(rng.random()), rng.random(2))
        if s is not None:
            acc += s[0] * abs(float(s[1][2])) / s[2]
    assert np.allclose(acc / n, albedo, rtol=0.05, atol=5e-3), (acc / n, albedo)


def test_fresnel_terms_against_their_closed_forms(olib):
    """scattering.rs:32-104 at the points where optics gives the answer without the formula: normal incidence R = ((eta - 1) / (eta + 1))^2 from either side and
    ((n - 1)^2 + k^2) / ((n + 1)^2 + k^2) for a conductor, grazing incidence R = 1, total internal reflection beyond the critical angle, Brewster's angle (the p-polarised
    term vanishes: R = r_s^2 / 2), and R + T = 1 is what DielectricBxDF::sample_f's choice between the two lobes assumes (pr + pt, bxdf.rs:594-600)."""
    olib.orc_fn_fresnel_dielectric.restype, olib.orc_fn_fresnel_dielectric.argtypes = F, [F, F]
    olib.orc_fn_fresnel_complex.restype, olib.orc_fn_fresnel_complex.argtypes = F, [F, F, F]
    for eta in (1.33, 1.5, 2.4):
        r0 = ((eta - 1.0) / (eta + 1.0)) ** 2
        assert olib.orc_fn_fresnel_dielectric(1.0, eta) == pytest.approx(r0, rel=1e-5)
        assert olib.orc_fn_fresnel_dielectric(-1.0, eta) == pytest.approx(r0, rel=1e-5)  # from inside (cos < 0 flips the interface: scattering.rs:36-41)
        assert olib.orc_fn_fresnel_dielectric(1e-6, eta) == pytest.approx(1.0, abs=1e-4)
        cos_crit = math.sqrt(1.0 - 1.0 / (eta * eta))
        assert olib.orc_fn_fresnel_dielectric(-(cos_crit - 1e-3), eta) == 1.0            # beyond the critical angle, from inside
        assert olib.orc_fn_fresnel_dielectric(-(cos_crit + 1e-2), eta) < 1.0
        cb = math.cos(math.atan(eta))                                                     # Brewster
        ct = math.sqrt(1.0 - (1.0 - cb * cb) / (eta * eta))
        rs = (cb - eta * ct) / (cb + eta * ct)
        assert olib.orc_fn_fresnel_dielectric(cb, eta) == pytest.approx(0.5 * rs * rs, rel=1e-4)
    for n, k in ((0.2, 3.9), (1.4, 1.6), (2.0, 0.0)):
        r0 = ((n - 1.0) ** 2 + k * k) / ((n + 1.0) ** 2 + k * k)
        assert olib.orc_fn_fresnel_complex(1.0, n, k) == pytest.approx(r0, rel=1e-5)
        assert olib.orc_fn_fresnel_complex(1e-6, n, k) == pytest.approx(1.0, abs=1e-4)
    # a conductor without absorption is a dielectric
    for c in (1.0, 0.7, 0.2):
        assert olib.orc_fn_fresnel_complex(c, 1.5, 0.0) == pytest.approx(olib.orc_fn_fresnel_dielectric(c, 1.5), rel=1e-5)
