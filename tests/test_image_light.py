"""ImageInfinitelight (light.rs:805-981; SURVEY §8f row 2, second half) in the oracle = the shared headers: the equal-area
octahedral mapping, PiecewiseConstant2D sampling (plain and compensated), the radiance lookup, and whole renders in which three
differently-sampled estimators must agree. No reference known answers exist for any of it; `poly_array` (fast_polynomial) and
rgb2spec's `fetch` are un-vendored crates (parity unpinned there, texture.h)."""
import ctypes as C
import math

import numpy as np
import pytest

import oracle_py
from oracle_py import fa
from shimmer_amd import abi, render, scenes

f32 = np.float32
LAMS = (452.0, 533.0, 601.5, 688.25)


@pytest.fixture(scope="module")
def env(lib):
    sc = scenes.three_spheres(lib, 48, 32, camera=(0.75, 0.5, 9.0), environment=scenes.environment_image(16))
    o = oracle_py.Oracle(sc.desc)
    li = [i for i, l in enumerate(sc.builder.lights) if l.kind == abi.SHM_LIGHT_IMAGE_INFINITE][0]
    yield sc, o, li
    o.close()


def square_to_sphere_as_written(p):
    """math.rs:456-485 in float64, including `vp - up / r + 1.0` (PBRT-v4: (vp - up) / r + 1)."""
    u, v = 2 * p[0] - 1, 2 * p[1] - 1
    up, vp = abs(u), abs(v)
    sd = 1 - (up + vp)
    r = 1 - abs(sd)
    phi = (1.0 if r == 0 else vp - up / r + 1.0) * math.pi / 4
    z = math.copysign(1 - r * r, sd)
    return np.array([math.copysign(math.cos(phi), u) * r * math.sqrt(max(0, 2 - r * r)), math.copysign(math.sin(phi), v) * r * math.sqrt(max(0, 2 - r * r)), z])


def sphere_to_square_exact(d):
    """The exact equal-area octahedral map (Clarberg), float64 — what math.rs:488-540 approximates with a polynomial atan."""
    x, y, z = abs(d[0]), abs(d[1]), abs(d[2])
    r = math.sqrt(max(0, 1 - z))
    phi = math.atan2(min(x, y), max(x, y)) * 2 / math.pi if max(x, y) > 0 else 0.0
    if x < y:
        phi = 1 - phi
    v = phi * r
    u = r - v
    if d[2] < 0:
        u, v = 1 - v, 1 - u
    return np.array([0.5 * (math.copysign(u, d[0]) + 1), 0.5 * (math.copysign(v, d[1]) + 1)])


def test_equal_area_mappings(orc):
    rng = np.random.default_rng(1)
    out3, out2 = (C.c_float * 3)(), (C.c_float * 2)()
    for _ in range(300):
        p = rng.uniform(0, 1, 2).astype(np.float32)
        orc.orc_fn_equal_area_square_to_sphere(fa(*p), out3)
        assert np.allclose(out3[:], square_to_sphere_as_written(p.astype(np.float64)), atol=3e-6)
        d = rng.normal(size=3)
        d = (d / np.linalg.norm(d)).astype(np.float32)
        orc.orc_fn_equal_area_sphere_to_square(fa(*d), out2)
        assert np.allclose(out2[:], sphere_to_square_exact(d.astype(np.float64)), atol=2e-5)  # the 6th-degree minimax atan
        assert 0.0 <= out2[0] <= 1.0 and 0.0 <= out2[1] <= 1.0
    for p, want in (((0.5, 0.5), (0, 0, 1)), ((0.0, 0.0), (0, 0, -1))):  # the pole and a corner (the opposite pole)
        orc.orc_fn_equal_area_square_to_sphere(fa(*p), out3)
        assert np.allclose(out3[:], want, atol=1e-6)


def distribution(o, li, which):
    n = o.lib.orc_fn_image_light_distribution(o.handle, li, which, None, None, None)
    func, mcdf, integral = (C.c_float * (n * n))(), (C.c_float * (n + 1))(), C.c_float()
    o.lib.orc_fn_image_light_distribution(o.handle, li, which, func, mcdf, C.byref(integral))
    return n, np.array(func[:], np.float64).reshape(n, n), np.array(mcdf[:], np.float64), integral.value


def test_distributions_follow_the_image(env):
    """light.rs:939-955: d = channel average per pixel; compensated = max(d - mean(d), 0); PiecewiseConstant2D over [0,1]^2."""
    sc, o, li = env
    img = scenes.environment_image(16).astype(np.float64)
    d = img.sum(axis=2) / 3
    n, func, mcdf, integral = distribution(o, li, 0)
    assert n == 16 and np.allclose(func, d, rtol=1e-6)
    assert integral == pytest.approx(d.mean(), rel=1e-5)
    assert mcdf[0] == 0 and mcdf[-1] == pytest.approx(1.0, abs=1e-6) and np.all(np.diff(mcdf) >= 0)
    assert np.allclose(np.diff(mcdf), d.mean(axis=1) / d.mean() / n, atol=1e-6)
    _, cfunc, _, cint = distribution(o, li, 1)
    assert np.allclose(cfunc, np.maximum(d - d.mean(), 0), rtol=1e-5, atol=1e-6)
    assert (cfunc == 0).sum() > 0.5 * n * n  # the sun dominates the mean: most of the sky is left to BSDF sampling


def pc2d_sample64(func, u):
    """PiecewiseConstant2D::sample (sampling.rs:160-168) over [0,1]^2 in float64."""
    n = func.shape[0]

    def sample1d(f, x):
        cdf = np.concatenate([[0.0], np.cumsum(f / n)])
        integral = cdf[-1]
        cdf = cdf / integral if integral > 0 else np.arange(n + 1) / n
        off = min(max(int(np.searchsorted(cdf, x, side="right")) - 1, 0), n - 1)
        du = x - cdf[off]
        if cdf[off + 1] - cdf[off] > 0:
            du /= cdf[off + 1] - cdf[off]
        return (off + du) / n, (f[off] / integral if integral > 0 else 0.0), off

    d1, pdf1, iv = sample1d(func.mean(axis=1), u[1])
    d0, pdf0, iu = sample1d(func[iv], u[0])
    return np.array([d0, d1]), pdf0 * pdf1, (iu, iv)


@pytest.mark.parametrize("allow_incomplete_pdf", [0, 1])
def test_sample_li_against_float64(env, allow_incomplete_pdf):
    """sample_li (light.rs:848-880) re-evaluated in float64: the PiecewiseConstant2D sample, the direction through
    equal_area_square_to_sphere AS WRITTEN and render_from_light, pdf = map_pdf / 4pi, radiance = image_le at the sampled uv.
    (The reference's square_to_sphere is not the inverse of its sphere_to_square — `vp - up / r + 1` — so pdf_li(wi) and le(wi) of
    a sampled direction are NOT the sample's pdf and radiance away from the equator; the restatement keeps that.)"""
    sc, o, li = env
    n, func, _, integral = distribution(o, li, allow_incomplete_pdf)
    img = scenes.environment_image(16)
    cs, light = sc.builder.color_space, sc.builder.lights[li]
    rfl = np.array(sc.builder.image_lights[0].render_from_light[:], np.float64).reshape(4, 4)[:3, :3]
    from test_textures import fetch64
    rng = np.random.default_rng(5 + allow_incomplete_pdf)
    out8 = (C.c_float * 8)()
    cells = np.zeros((n, n))
    total = 1500
    for k in range(total):
        u = rng.uniform(0, 1, 2).astype(np.float32)
        assert o.lib.orc_fn_light_sample_li(o.handle, li, fa(*u), allow_incomplete_pdf, fa(*LAMS), out8) == 1
        wi, pdf, L = np.array(out8[0:3], np.float64), out8[3], np.array(out8[4:8])
        uv, map_pdf, (iu, iv) = pc2d_sample64(func, u.astype(np.float64))
        cells[iv, iu] += 1
        frac = np.abs((uv * n) % 1 - 0.5)
        if frac.max() > 0.49:
            continue  # a sample on a texel border may round into the neighbour in float32
        assert pdf == pytest.approx(map_pdf / (4 * math.pi), rel=1e-4)
        assert np.allclose(wi, rfl @ square_to_sphere_as_written(uv), atol=2e-4)
        rgb = img[iv, iu].astype(np.float64)
        sc2 = 2 * rgb.max()
        c = fetch64(cs, rgb / sc2)
        want = [light.scale * sc2 * (0.5 + 0.5 * x / math.sqrt(1 + x * x)) * cs["illuminant"][int(math.floor(l + 0.5)) - 360]
                for l in LAMS for x in [(c[0] * l + c[1]) * l + c[2]]]
        assert np.allclose(L, want, rtol=5e-4)
    # texels are drawn in proportion to the distribution (coarse 4x4 blocks)
    want = func.reshape(4, 4, 4, 4).sum(axis=(1, 3))
    got = cells.reshape(4, 4, 4, 4).sum(axis=(1, 3))
    mask = want / want.sum() > 0.03
    assert np.allclose(got[mask] / total, want[mask] / want.sum(), rtol=0.3, atol=0.02)
    # on the equator (r = 1) the quirk vanishes: there pdf_li and le of the sampled direction are the sample's own
    out3, out4 = (C.c_float * 3)(), (C.c_float * 4)()
    for uv in ((0.3, 0.2), (0.85, 0.35), (0.4, 0.9)):
        o.lib.orc_fn_equal_area_square_to_sphere(fa(*uv), out3)
        assert abs(out3[2]) < 1e-6
        back = (C.c_float * 2)()
        o.lib.orc_fn_equal_area_sphere_to_square(out3, back)
        assert np.allclose(back[:], uv, atol=1e-5)


def test_le_is_the_nearest_texel_as_an_illuminant_spectrum(env):
    """image_le (light.rs:968-977): nearest texel with the octahedral wrap, clamp_zero, RgbIlluminantSpectrum, times scale."""
    sc, o, li = env
    img = scenes.environment_image(16)
    cs = sc.builder.color_space
    light = sc.builder.lights[li]
    rot = np.array(sc.builder.image_lights[0].light_from_render[:], np.float64).reshape(4, 4)[:3, :3]
    from test_textures import fetch64
    rng = np.random.default_rng(2)
    out4 = (C.c_float * 4)()
    checked = 0
    for _ in range(200):
        d = rng.normal(size=3)
        d = (d / np.linalg.norm(d)).astype(np.float32)
        uv = sphere_to_square_exact(rot @ d.astype(np.float64))
        if np.abs((uv * 16) % 1 - 0.5).max() > 0.45:
            continue
        rgb = img[min(int(uv[1] * 16), 15), min(int(uv[0] * 16), 15)].astype(np.float64)
        sc2 = 2 * rgb.max()
        c = fetch64(cs, rgb / sc2)
        want = []
        for l in LAMS:
            x = (c[0] * l + c[1]) * l + c[2]
            want.append(light.scale * sc2 * (0.5 + 0.5 * x / math.sqrt(1 + x * x)) * cs["illuminant"][int(math.floor(l + 0.5)) - 360])
        o.lib.orc_fn_infinite_light_le(o.handle, li, fa(*d), fa(*LAMS), out4)
        assert np.allclose(out4[:], want, rtol=5e-4)
        checked += 1
    assert checked > 100


def test_three_estimators_agree_under_the_environment_map(lib):
    """PathIntegrator samples the COMPENSATED distribution and MIS-combines it with BSDF sampling; SimplePathIntegrator samples the
    plain distribution without MIS; RandomWalkIntegrator never samples the light. All three estimate the same image."""
    sc = scenes.three_spheres(lib, 36, 24, camera=(0.75, 0.5, 9.0), environment=scenes.environment_image(16))
    o = oracle_py.Oracle(sc.desc)
    means = {}
    try:
        for integ, spp in (("path", 64), ("simplepath", 64), ("randomwalk", 1024)):
            film, _ = o.render(render.make_params(spp=spp, max_depth=4, seed=11, integrator=integ), n_threads=8)
            rgb = film["rgb_sum"] / film["weight_sum"][..., None]
            assert np.isfinite(rgb).all()
            means[integ] = rgb[12:, :, :].mean()  # the lower half: spheres and ground, lit only through the estimators
    finally:
        o.close()
    assert means["simplepath"] == pytest.approx(means["path"], rel=0.03)
    assert means["randomwalk"] == pytest.approx(means["path"], rel=0.06)


def test_scene_creation_errors(lib):
    b = scenes.three_spheres(lib, 16, 16, camera=(0.75, 0.5, 9.0), environment=scenes.environment_image(8)).builder
    b.tex_levels[-1] = (8, 4, b.tex_levels[-1][2])  # not square (light.rs:934-937 panics)
    desc, _ = b.build(lib)
    with pytest.raises(RuntimeError, match="square"):
        oracle_py.Oracle(desc)
