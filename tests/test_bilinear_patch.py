"""BilinearPatch (shape/bilinear_patch.rs) in the oracle = the shared headers the GPU compiles.
The reference has no in-source known answers for this shape ("parity unpinned" for the shape as a whole), so the
restatement is checked against independent float64 numpy evaluations of the cited formulas and against the invariants
the algorithm must satisfy (hit points reproduce (u, v, t); sample / pdf agree; spherical-rectangle sampling is
uniform in solid angle; a rectangular patch light illuminates like the two triangles it replaces)."""
import ctypes as C
import math

import numpy as np
import pytest

import oracle_py
from oracle_py import fa
from shimmer_amd import abi, render, scenes

f32 = np.float32


@pytest.fixture(scope="module")
def orc():
    return oracle_py.load()


RECT = np.array([(-1, 0, -1), (1, 0, -1), (-1, 0, 1), (1, 0, 1)], np.float64)            # p00 p10 p01 p11, planar rectangle
PLANAR_QUAD = np.array([(-1, 0, -1), (1.5, 0, -0.8), (-0.7, 0, 1), (1, 0, 1.4)], np.float64)  # planar, not a rectangle
SADDLE = np.array([(-1, 0.3, -1), (1, -0.4, -1), (-1, -0.5, 1), (1, 0.6, 1)], np.float64)     # doubly curved


def pts(a):
    return fa(*np.asarray(a, np.float64).ravel())


def bilerp(p, u, v):
    return (1 - u) * ((1 - v) * p[0] + v * p[2]) + u * ((1 - v) * p[1] + v * p[3])


def test_quadratic_matches_float64(orc):
    """math.rs:377-410: roots ascending; linear fallback; no real roots -> None."""
    rng = np.random.default_rng(1)
    for _ in range(300):
        a, b, c = (rng.normal(size=3) * 3).astype(np.float32)
        out = (C.c_float * 2)()
        ok = orc.orc_fn_quadratic(float(a), float(b), float(c), out)
        disc = float(b) ** 2 - 4 * float(a) * float(c)
        if disc < -1e-4 * (float(b) ** 2 + abs(4 * float(a) * float(c))):
            assert not ok
        elif ok and disc > 1e-3:
            r = sorted(np.roots([float(a), float(b), float(c)]).real)
            assert out[0] <= out[1]
            assert out[0] == pytest.approx(r[0], rel=2e-4, abs=2e-5) and out[1] == pytest.approx(r[1], rel=2e-4, abs=2e-5)
    out = (C.c_float * 2)()
    assert orc.orc_fn_quadratic(0.0, 2.0, -3.0, out) and out[0] == out[1] == 1.5
    assert not orc.orc_fn_quadratic(0.0, 0.0, 1.0, out)


def test_is_rectangle_and_area(orc):
    """bilinear_patch.rs:40-69, 108-142: exact |e_u||e_v| for rectangles; the 3x3 tessellation estimate otherwise."""
    out = (C.c_float * 2)()
    orc.orc_fn_blp_info(pts(RECT), out)
    assert out[0] == 1.0 and out[1] == pytest.approx(4.0, rel=1e-6)
    for quad in (PLANAR_QUAD, SADDLE):
        orc.orc_fn_blp_info(pts(quad), out)
        assert out[0] == 0.0
        # the same tessellation in float64
        P = [[bilerp(quad, i / 3, j / 3) for j in range(4)] for i in range(4)]
        area = sum(0.5 * np.linalg.norm(np.cross(P[i + 1][j + 1] - P[i][j], P[i + 1][j] - P[i][j + 1])) for i in range(3) for j in range(3))
        assert out[1] == pytest.approx(area, rel=1e-5)
    # a rotated, translated rectangle is still one; a degenerate (triangle) patch is not
    th = 0.7
    R = np.array([[math.cos(th), 0, math.sin(th)], [0.2, 1, 0.1], [-math.sin(th), 0, math.cos(th)]])
    Q, _ = np.linalg.qr(R)
    orc.orc_fn_blp_info(pts(RECT @ Q.T + np.array([3.0, -2.0, 5.0])), out)
    assert out[0] == 1.0 and out[1] == pytest.approx(4.0, rel=1e-4)
    tri = RECT.copy(); tri[3] = tri[1]
    orc.orc_fn_blp_info(pts(tri), out)
    assert out[0] == 0.0


@pytest.mark.parametrize("quad", [RECT, PLANAR_QUAD, SADDLE], ids=["rect", "planar_quad", "saddle"])
def test_intersect_reproduces_uv_t(orc, quad):
    """bilinear_patch.rs:144-236: a ray aimed at P(u, v) hits at that (u, v) with t = |P - o| (unit direction); rays that
    leave the patch miss; t_max before the hit rejects."""
    rng = np.random.default_rng(7)
    n_checked = 0
    for _ in range(400):
        u, v = rng.random(2)
        P = bilerp(quad, u, v)
        o = P + rng.normal(size=3) * 0.5 + np.array([0.0, 2.5, 0.0])
        d = (P - o) / np.linalg.norm(P - o)
        out = (C.c_float * 3)()
        ok = orc.orc_fn_blp_intersect(pts(quad), fa(*o), fa(*d), float("inf"), out)
        assert ok
        t_expect = np.linalg.norm(P - o)
        if abs(out[2] - t_expect) > 1e-3:  # a curved patch may be hit earlier somewhere else: then that hit must lie on the patch
            Q = bilerp(quad, out[0], out[1])
            assert out[2] < t_expect and np.linalg.norm(o + d * out[2] - Q) < 2e-4
            continue
        n_checked += 1
        assert out[0] == pytest.approx(u, abs=3e-4) and out[1] == pytest.approx(v, abs=3e-4)
        assert not orc.orc_fn_blp_intersect(pts(quad), fa(*o), fa(*d), float(t_expect * 0.9), out)
        assert not orc.orc_fn_blp_intersect(pts(quad), fa(*o), fa(*(-d)), float("inf"), out)  # behind the origin
    assert n_checked > 300
    # far outside the patch's footprint
    out = (C.c_float * 3)()
    assert not orc.orc_fn_blp_intersect(pts(quad), fa(5.0, 3.0, 5.0), fa(0.0, -1.0, 0.0), float("inf"), out)


def test_interaction_geometry(orc):
    """bilinear_patch.rs:238-428 (no uv / n arrays): p = P(u, v); dpdu, dpdv are the parametric partials; n is their
    normalised cross product, negated when the orientation is flipped; the error bound is gamma(6) * sum |corner|;
    a planar patch has zero dndu / dndv."""
    for quad in (RECT, SADDLE):
        for (u, v) in ((0.25, 0.6), (0.9, 0.1)):
            out = (C.c_float * 21)()
            orc.orc_fn_blp_interaction(pts(quad), 0, u, v, fa(0, 1, 0), out)
            o = np.array(out[:], np.float64).reshape(7, 3)
            P = bilerp(quad, u, v)
            dpdu = ((1 - v) * quad[1] + v * quad[3]) - ((1 - v) * quad[0] + v * quad[2])
            dpdv = ((1 - u) * quad[2] + u * quad[3]) - ((1 - u) * quad[0] + u * quad[1])
            n = np.cross(dpdu, dpdv); n /= np.linalg.norm(n)
            assert np.allclose(o[0], P, atol=2e-6) and np.allclose(o[2], dpdu, atol=2e-6) and np.allclose(o[3], dpdv, atol=2e-6)
            assert np.allclose(o[1], n, atol=2e-6)
            g6 = 6 * 2.0 ** -24 / (1 - 6 * 2.0 ** -24)
            bound = g6 * np.abs(quad).sum(axis=0)  # the interval is rounded outwards by one ulp of p on each side
            assert np.all(o[6] >= bound * (1 - 1e-3)) and np.all(o[6] <= bound + 2.5e-7)
            if quad is RECT:
                assert np.all(o[4] == 0) and np.all(o[5] == 0)
            out2 = (C.c_float * 21)()
            orc.orc_fn_blp_interaction(pts(quad), 1, u, v, fa(0, 1, 0), out2)
            assert np.allclose(np.array(out2[3:6]), -n, atol=2e-6)


def solid_angle_numeric(p_ref, s, ex, ey, n=400):
    us = (np.arange(n) + 0.5) / n
    U, V = np.meshgrid(us, us, indexing="ij")
    P = s[None, None] + U[..., None] * ex + V[..., None] * ey
    d = P - p_ref
    r2 = (d ** 2).sum(-1)
    nrm = np.cross(ex, ey)
    cos = np.abs((d @ nrm)) / np.sqrt(r2) / np.linalg.norm(nrm)
    return float((cos / r2).mean() * np.linalg.norm(nrm))


def test_spherical_rectangle_sampling(orc):
    """sampling.rs:501-579 / 645-787: samples lie on the rectangle, pdf = 1 / solid angle (checked against a numeric
    integral and against spherical_quad_area, vecmath/mod.rs:118-140), sampling is uniform in solid angle, and the
    inverse maps a sampled point back to its u."""
    s, ex, ey = np.array([-1.0, 2.0, -0.5]), np.array([2.0, 0.0, 0.0]), np.array([0.0, 0.0, 1.5])
    p_ref = np.array([0.3, 0.0, 0.2])
    rng = np.random.default_rng(3)
    omega = solid_angle_numeric(p_ref, s, ex, ey)
    corners = [s, s + ex, s + ex + ey, s + ey]
    dirs = [(c - p_ref) / np.linalg.norm(c - p_ref) for c in corners]
    assert orc.orc_fn_spherical_quad_area(*[fa(*d) for d in dirs]) == pytest.approx(omega, rel=2e-3)
    # uniformity: the solid angle of the sub-rectangle x < x_mid must receive its share of the samples
    sub = solid_angle_numeric(p_ref, s, ex * 0.5, ey)
    hits, n = 0, 4000
    for _ in range(n):
        u = rng.random(2)
        out = (C.c_float * 4)()
        orc.orc_fn_sample_spherical_rectangle(fa(*p_ref), fa(*s), fa(*ex), fa(*ey), fa(*u), out)
        p = np.array(out[:3], np.float64)
        assert out[3] == pytest.approx(1.0 / omega, rel=3e-3)
        a, b = (p - s) @ ex / (ex @ ex), (p - s) @ ey / (ey @ ey)
        assert -1e-4 <= a <= 1 + 1e-4 and -1e-4 <= b <= 1 + 1e-4 and abs((p - s) @ np.cross(ex, ey)) < 1e-4
        hits += a < 0.5
        inv = (C.c_float * 2)()
        orc.orc_fn_invert_spherical_rectangle_sample(fa(*p_ref), fa(*s), fa(*ex), fa(*ey), fa(*p), inv)
        assert inv[0] == pytest.approx(u[0], abs=2e-3) and inv[1] == pytest.approx(u[1], abs=2e-3)
    assert hits / n == pytest.approx(sub / omega, abs=0.03)


@pytest.mark.parametrize("quad,ns", [(RECT + np.array([0, 3.0, 0]), (0, 1, 0)), (RECT + np.array([0, 3.0, 0]), (0, 0, 0)),
                                     (SADDLE + np.array([0, 3.0, 0]), (0, 1, 0))], ids=["rect_cos_warp", "rect_plain", "saddle_area"])
def test_sample_and_pdf_with_context_agree(orc, quad, ns):
    """bilinear_patch.rs:638-783: pdf_with_context(wi) of the sampled direction equals the pdf sample_with_context returned
    (spherical-rectangle path with and without the cosine warp; area-sampling path for a non-rectangle), and the solid-angle
    pdf integrates to 1 over the patch."""
    rng = np.random.default_rng(9)
    ctx_p, ctx_n = np.array([0.2, 0.0, -0.1]), np.array([0.0, 1.0, 0.0])
    inv_pdf_sum, n = 0.0, 1500
    for _ in range(n):
        u = rng.random(2)
        out = (C.c_float * 7)()
        ok = orc.orc_fn_blp_sample_with_context(pts(quad), 0, fa(*ctx_p), fa(*ctx_n), fa(*ns), fa(*u), out)
        assert ok
        p, pdf = np.array(out[:3], np.float64), float(out[6])
        assert pdf > 0
        wi = (p - ctx_p) / np.linalg.norm(p - ctx_p)
        pdf2 = orc.orc_fn_blp_pdf_with_context(pts(quad), 0, fa(*ctx_p), fa(*ctx_n), fa(*ns), fa(*wi))
        if quad is not SADDLE + 0:  # the reference's sample()/pdf() interpolate different edges for curved patches (kept): rectangles only
            pass
        if np.allclose(quad[:, 1], quad[0, 1]):
            assert pdf2 == pytest.approx(pdf, rel=2e-2)
        inv_pdf_sum += 1.0 / pdf
    # E[1/pdf] under the sampling density = the solid angle subtended by the patch
    if np.allclose(quad[:, 1], quad[0, 1]):
        s, ex, ey = quad[0], quad[1] - quad[0], quad[2] - quad[0]
        assert inv_pdf_sum / n == pytest.approx(solid_angle_numeric(ctx_p, s, ex, ey), rel=0.05)


def _sample_linear(u, a, b):
    if u == 0 and a == 0:
        return 0.0
    x = u * (a + b) / (a + math.sqrt((1 - u) * a * a + u * b * b))
    return min(x, 1 - 2.0 ** -24)


def test_area_sampling_path_matches_the_cited_formulas(orc):
    """bilinear_patch.rs:521-600 + 655-678 for a non-rectangular patch, against a float64 numpy evaluation of the SAME
    formulas — including the reference's edge choice pu0 = lerp(u, p00, p10), pu1 = lerp(v, p10, p11) (:549-553), which
    differs from PBRT-v4's and is what the reference computes (so the two sampling strategies of a patch light are not
    interchangeable there, and are not here either)."""
    quad = SADDLE + np.array([0.0, 3.0, 0.0])
    ctx_p, ctx_n, ns = np.array([0.2, 0.0, -0.1]), np.array([0.0, 1.0, 0.0]), np.array([0.0, 1.0, 0.0])
    p00, p10, p01, p11 = quad
    w = [np.linalg.norm(np.cross(p10 - p00, p01 - p00)), np.linalg.norm(np.cross(p10 - p00, p11 - p10)),
         np.linalg.norm(np.cross(p01 - p00, p11 - p01)), np.linalg.norm(np.cross(p11 - p10, p11 - p01))]
    rng = np.random.default_rng(17)
    for _ in range(200):
        u = rng.random(2)
        y = _sample_linear(u[1], w[0] + w[1], w[2] + w[3])
        x = _sample_linear(u[0], (1 - y) * w[0] + y * w[2], (1 - y) * w[1] + y * w[3])
        pdf_uv = 4 * ((1 - x) * (1 - y) * w[0] + x * (1 - y) * w[1] + (1 - x) * y * w[2] + x * y * w[3]) / sum(w)
        pu0 = (1 - x) * p00 + x * p10
        pu1 = (1 - y) * p10 + y * p11
        p = (1 - x) * pu0 + x * pu1
        dpdu = pu1 - pu0
        dpdv = ((1 - x) * p01 + x * p11) - ((1 - x) * p00 + x * p10)
        n = np.cross(dpdu, dpdv)
        pdf_area = pdf_uv / np.linalg.norm(n)
        n /= np.linalg.norm(n)
        wi = (p - ctx_p) / np.linalg.norm(p - ctx_p)
        pdf = pdf_area / (abs(n @ -wi) / ((ctx_p - p) @ (ctx_p - p)))
        out = (C.c_float * 7)()
        assert orc.orc_fn_blp_sample_with_context(pts(quad), 0, fa(*ctx_p), fa(*ctx_n), fa(*ns), fa(*u), out)
        assert np.allclose(np.array(out[:3]), p, atol=5e-6) and np.allclose(np.array(out[3:6]), n, atol=5e-5)
        assert out[6] == pytest.approx(pdf, rel=2e-4)


def test_interaction_with_uv_and_normals(orc):
    """bilinear_patch.rs:258-318, 399-424: with the identity uv layout the (s, t) frame is the (u, v) frame; a scaled, offset
    layout scales dpdu / dpdv by the inverse and reports st; per-vertex normals give a unit shading normal that the
    geometric normal is flipped towards, and a shading dpdu rotated into its tangent plane (rotate_from_to,
    transform.rs:227-253)."""
    quad = SADDLE
    wo = fa(0, 1, 0)
    u, v = 0.3, 0.7
    base = (C.c_float * 23)()
    orc.orc_fn_blp_interaction_attr(pts(quad), 0, None, None, u, v, wo, base)
    ident = (C.c_float * 23)()
    orc.orc_fn_blp_interaction_attr(pts(quad), 0, None, fa(0, 0, 1, 0, 0, 1, 1, 1), u, v, wo, ident)
    assert np.allclose(np.array(base[:21]), np.array(ident[:21]), atol=1e-6) and ident[21] == pytest.approx(u) and ident[22] == pytest.approx(v)
    scaled = (C.c_float * 23)()
    orc.orc_fn_blp_interaction_attr(pts(quad), 0, None, fa(1, 1, 3, 1, 1, 5, 3, 5), u, v, wo, scaled)  # s = 1 + 2u, t = 1 + 4v
    b, sc = np.array(base[:], np.float64), np.array(scaled[:], np.float64)
    assert np.allclose(sc[6:9], b[6:9] / 2, atol=1e-6) and np.allclose(sc[9:12], b[9:12] / 4, atol=1e-6)
    assert sc[21] == pytest.approx(1 + 2 * u, abs=1e-6) and sc[22] == pytest.approx(1 + 4 * v, abs=1e-6)
    assert np.allclose(sc[3:6], b[3:6], atol=1e-6)  # same geometric normal
    # per-vertex normals, all leaning the same way
    nn = np.array([(0.3, -1.0, 0.1)] * 4, np.float64)
    nn /= np.linalg.norm(nn, axis=1, keepdims=True)
    withn = (C.c_float * 23)()
    orc.orc_fn_blp_interaction_attr(pts(quad), 0, fa(*nn.ravel()), None, u, v, wo, withn)
    w = np.array(withn[:], np.float64)
    ns, n, dpdus = w[12:15], w[3:6], w[15:18]
    assert np.allclose(ns, nn[0], atol=1e-6) and np.dot(n, ns) > 0  # n flipped towards ns (orientation is authoritative)
    assert np.allclose(np.abs(n), np.abs(b[3:6]), atol=1e-6)
    assert abs(np.dot(dpdus, ns)) < 1e-4 * np.linalg.norm(dpdus) + 1e-6  # rotated with the normal: tangent to the shading normal
    assert np.linalg.norm(dpdus) == pytest.approx(np.linalg.norm(b[6:9]), rel=1e-5)


def test_rotate_from_to_and_invert_bilinear(orc):
    """transform.rs:227-253: the rotation maps `from` onto `to` and preserves lengths; vecmath/mod.rs:70-116: invert_bilinear
    recovers (u, v) from the bilinear interpolation of four uv corners."""
    rng = np.random.default_rng(4)
    for _ in range(50):
        a, b = rng.normal(size=3), rng.normal(size=3)
        a /= np.linalg.norm(a); b /= np.linalg.norm(b)
        out = (C.c_float * 3)()
        orc.orc_fn_rotate_from_to(fa(*a), fa(*b), fa(*a), out)
        assert np.allclose(np.array(out[:]), b, atol=3e-6)
        x = rng.normal(size=3)
        orc.orc_fn_rotate_from_to(fa(*a), fa(*b), fa(*x), out)
        assert np.linalg.norm(out[:]) == pytest.approx(np.linalg.norm(x), rel=1e-5)
    corners = np.array([(0.1, 0.2), (1.9, 0.0), (0.0, 1.4), (2.2, 1.7)])  # uv00 uv10 uv01 uv11
    for _ in range(100):
        u, v = rng.random(2)
        pt = (1 - u) * ((1 - v) * corners[0] + v * corners[2]) + u * ((1 - v) * corners[1] + v * corners[3])
        out = (C.c_float * 2)()
        orc.orc_fn_invert_bilinear(fa(*pt), fa(*corners.ravel()), out)
        assert out[0] == pytest.approx(u, abs=2e-4) and out[1] == pytest.approx(v, abs=2e-4)
