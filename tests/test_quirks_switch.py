"""SHM_REFERENCE_QUIRKS (ShmRenderParams::disable_reference_quirks, ABI v8). The default (0) reproduces every deviation of
the reference from PBRT-v4 and is what every parity test renders with. Switched off, the sites where the reference produces invalid or
biased values follow PBRT-v4 instead. These tests pin the switch on the CPU oracle (the shared arithmetic); the GPU half is in
tests/test_gpu_parity.py (GPU == oracle, bit for bit, in both settings)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_py
from shimmer_amd import abi, render, scene as scn, scenes


def test_params_default_is_reference_exact():
    p = render.make_params()
    assert p.disable_reference_quirks == 0 and C.sizeof(abi.ShmRenderParams) == 32
    assert render.make_params(reference_quirks=False).disable_reference_quirks == 1


@pytest.fixture(scope="module")
def coated_s3(lib):
    return scenes.ganesha_proxy(lib, 1024, 1024, coated=True)


def test_layered_pdf_nan_pixel_disappears_with_the_quirks_off(lib, coated_s3):
    """bxdf.rs:1491-1506: LayeredBxDF::pdf uses the reflected sample `rs` without PBRT-v4's `rs.f != 0 && rs.pdf > 0`; pixel (714, 268),
    sample 83 of the coated S3 frame then gets power_heuristic(1, 0, 1, 0) = 0 / 0 (tests/test_gpu_parity.py pins that NaN). With the
    quirks off the same sample is finite."""
    x, y, sample = 714, 268, 83
    x0, y0 = x & ~7, y & ~7
    tiles, n = scn.tiles_for(lib, (x0, y0, x0 + 8, y0 + 8))
    orc = oracle_py.Oracle(coated_s3.desc)
    on, _ = orc.render(render.make_params(seed=0, spp=128, max_depth=5), n_threads=1, tiles=tiles, n_tiles=n, waves=[(sample, sample + 1)])
    off, _ = orc.render(render.make_params(seed=0, spp=128, max_depth=5, reference_quirks=False), n_threads=1, tiles=tiles, n_tiles=n, waves=[(sample, sample + 1)])
    orc.close()
    a, b = on["rgb_sum"][y0:y0 + 8, x0:x0 + 8], off["rgb_sum"][y0:y0 + 8, x0:x0 + 8]
    assert np.isnan(a[y - y0, x - x0]).all() and np.isfinite(a).sum() == 3 * 63
    assert np.isfinite(b).all()
    # (until round 6 the other 63 pixels kept their bits — the guard only fires on such samples; the switch now also covers the triangle emitters' sampling, which every
    #  lit pixel goes through: test_layered_pdf_guard_only_changes_degenerate_samples below holds the guard's own footprint at function level)


def test_layered_pdf_guard_only_changes_degenerate_samples(orc):
    """Function level: LayeredBxDF::pdf with and without the guard over random directions (test hook: a negative n_samples asks for it).
    Finite values of the unguarded version are reproduced bit for bit unless a reflected sample was degenerate; the guarded one is always finite."""
    rng = np.random.default_rng(5)
    p = np.zeros(19, np.float32)
    p[0:4], p[8:12], p[12], p[13], p[14], p[17], p[18] = 0.6, 0.0, 1.5, 0.0, 0.0, 0.01, 0.0   # CoatedDiffuse, smooth interface
    out_a, out_b = np.zeros(6, np.float32), np.zeros(6, np.float32)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    n_equal = n_total = 0
    for _ in range(4000):
        wo, wi = rng.normal(size=3).astype(np.float32), rng.normal(size=3).astype(np.float32)
        wo, wi = wo / np.linalg.norm(wo), wi / np.linalg.norm(wi)
        if rng.random() < 0.2:
            wi[2] = np.float32(1e-9)  # grazing: where horizon samples happen
        orc.orc_fn_layered_f_pdf(4, fp(p), (C.c_int * 2)(10, 1), fp(wo), fp(wi), fp(out_a))
        orc.orc_fn_layered_f_pdf(4, fp(p), (C.c_int * 2)(10, -1), fp(wo), fp(wi), fp(out_b))
        assert np.isfinite(out_b[4]) and out_b[4] >= 0
        assert np.array_equal(out_a[:4], out_b[:4])          # f is not touched by the switch
        n_total += 1
        n_equal += int(out_a[4].view(np.uint32) == out_b[4].view(np.uint32))
    assert n_equal >= n_total - 40  # (a handful of horizon samples at most)


def test_sphere_light_scene_with_the_quirks_off(lib):
    """S1 (sphere + quad light) has no sphere EMITTER, so Sphere::pdf_with_context is not reached; its two triangle lights are sampled as PBRT-v4 samples them since
    round 6; three spheres
    under a uniform sky lit through SimplePath's uniform sampling change with uniform_hemisphere_pdf (1/4pi -> 1/2pi halves those terms)."""
    s1 = scenes.sphere_light(lib, 32, 32)
    o = oracle_py.Oracle(s1.desc)
    a, _ = o.render(render.make_params(seed=1, spp=4), n_threads=4)
    b, _ = o.render(render.make_params(seed=1, spp=4, reference_quirks=False), n_threads=4)
    o.close()
    assert np.isfinite(b["rgb_sum"]).all() and not np.array_equal(a, b)
    assert b["rgb_sum"].mean() > a["rgb_sum"].mean()  # (by how much, and which of the two is the physical one: test_s1_against_an_estimator_that_never_samples_lights)
    ts = scenes.three_spheres(lib, 32, 32, camera=(0.0, 0.0, 12.0))
    o = oracle_py.Oracle(ts.desc)
    kw = dict(seed=2, spp=16, max_depth=3, integrator="simplepath", sample_lights=False, sample_bsdf=False)
    a, _ = o.render(render.make_params(**kw), n_threads=4)
    b, _ = o.render(render.make_params(reference_quirks=False, **kw), n_threads=4)
    o.close()
    ma, mb = a["rgb_sum"].mean(), b["rgb_sum"].mean()
    assert np.isfinite(b["rgb_sum"]).all() and ma > 0 and mb > 0 and not np.array_equal(a, b)
    assert mb < ma  # a larger pdf in the denominator: the uniformly sampled terms shrink


def test_sphere_emitter_pdf_constant(lib):
    """Sphere::pdf_with_context outside the sphere: 1 / (2.90 pi (1 - cos theta_max)) in the reference (sphere.rs:456), 2 pi with the quirks off."""
    import inspect
    src = inspect.getsource(scenes.random_scene)
    assert "add_sphere" in src  # the fuzz scenes (tests/test_fuzz_scenes.py, GPU parity) contain sphere emitters: the GPU test renders them both ways
    sc = scenes.random_scene(lib, 3)
    o = oracle_py.Oracle(sc.desc)
    a, _ = o.render(render.make_params(seed=3, spp=4), n_threads=4)
    b, _ = o.render(render.make_params(seed=3, spp=4, reference_quirks=False), n_threads=4)
    o.close()
    assert np.isfinite(b["rgb_sum"]).all()


def _tri_sample(orc, tri, ctx_p, ctx_n, ctx_ns, u, strict):
    fa = lambda v: (C.c_float * len(v))(*[float(x) for x in v])
    out = (C.c_float * 7)()
    orc.orc_fn_triangle_sample_with_context_strict.restype = C.c_int
    ok = orc.orc_fn_triangle_sample_with_context_strict(fa(tri[0]), fa(tri[1]), fa(tri[2]), fa(ctx_p), fa(ctx_n), fa(ctx_ns), fa(u), int(strict), out)
    return (np.array(out[0:3], np.float64), np.array(out[3:6], np.float64), float(out[6])) if ok else None


def test_triangle_emitters_are_sampled_as_pbrt_v4_does_with_the_quirks_off(orc):
    """Two reference behaviours of Triangle sampling that bias an image, behind the switch since round 6 (shm/shapes.h):
    (a) triangle.rs:639-641 computes the cosine-warped sample in a block whose `u` shadows the outer one: the direction is drawn from the unwarped u, the density carries the
        warp's factor. Quirks off: the warped u is the sample's — and only then does pdf_with_context return the density of the sampled direction;
    (c) sampling.rs:477 (above all: it is what keeps the reference's point off the sampled direction even without a warp);
    (b) triangle.rs:558-560 negates the sampled normal of a mesh without normals ALWAYS (PBRT-v4: only with reverse_orientation ^ transform_swaps_handedness): the area
        samples of a small one-sided emitter show their dark side."""
    orc.orc_fn_triangle_pdf_with_context.restype = C.c_float
    fa = lambda v: (C.c_float * len(v))(*[float(x) for x in v])
    tri = [(-1.0, 2.0, -1.2), (1.3, 2.0, -0.8), (0.2, 2.2, 1.5)]
    ctx_p, ctx_n, ctx_ns = (0.3, 0.0, 0.1), (0.0, 1.0, 0.0), (0.35, 0.9, 0.1)
    rng = np.random.default_rng(1)
    n_differ = 0
    for _ in range(200):
        u = rng.random(2)
        ref, strict = _tri_sample(orc, tri, ctx_p, ctx_n, ctx_ns, u, 0), _tri_sample(orc, tri, ctx_p, ctx_n, ctx_ns, u, 1)
        assert ref is not None and strict is not None
        assert ref[2] == strict[2]  # the same density either way (the warp's factor at the warped u times 1 / solid angle) ...
        n_differ += not np.allclose(ref[0], strict[0], atol=1e-6)  # ... for different points
        for (p, _, pdf), is_strict in ((ref, False), (strict, True)):
            wi = (p - np.array(ctx_p)) / np.linalg.norm(p - np.array(ctx_p))
            again = orc.orc_fn_triangle_pdf_with_context(fa(tri[0]), fa(tri[1]), fa(tri[2]), fa(ctx_p), fa(ctx_n), fa(ctx_ns), fa(wi))
            if is_strict:
                assert again == pytest.approx(pdf, rel=2e-3), (u, pdf, again)  # the density of the direction that was sampled
    assert n_differ > 190
    # without a shading normal at the reference point there is no warp — the points still differ: (c) sample_spherical_triangle's barycentrics of the sampled direction
    # are divided by e1 . e1 in the reference (sampling.rs:477; PBRT-v4: s1 . e1), so its point is not where that direction meets the triangle
    a, b = _tri_sample(orc, tri, ctx_p, (0, 0, 0), (0, 0, 0), (0.3, 0.6), 0), _tri_sample(orc, tri, ctx_p, (0, 0, 0), (0, 0, 0), (0.3, 0.6), 1)
    assert a[2] == b[2] and not np.allclose(a[0], b[0], atol=1e-4)
    # (b) a tiny triangle (below 3e-4 sr: area sampling): the geometric normal of p0 p1 p2 is +y here; the reference hands out -y, PBRT-v4 +y
    tiny = [(0.0, 30.0, 0.0), (0.0, 30.0, 0.3), (0.3, 30.0, 0.0)]
    geo = np.cross(np.array(tiny[1]) - np.array(tiny[0]), np.array(tiny[2]) - np.array(tiny[0]))
    geo /= np.linalg.norm(geo)
    a, b = _tri_sample(orc, tiny, (0, 0, 0), (0, 1, 0), (0, 1, 0), (0.3, 0.6), 0), _tri_sample(orc, tiny, (0, 0, 0), (0, 1, 0), (0, 1, 0), (0.3, 0.6), 1)
    assert np.allclose(a[1], -geo) and np.allclose(b[1], geo) and np.array_equal(a[0], b[0]) and a[2] == b[2]


def test_small_one_sided_emitter_lights_the_scene_only_with_the_quirks_off(lib):
    """The image-level consequence of (b): a floor under a FINELY tessellated one-sided emitter without normals (every triangle below 3e-4 sr from the floor) gets no
    next-event estimation in the reference — the power heuristic still discounts the BSDF samples that find the emitter, so most of the light is lost; quirks off, it is lit."""
    def floor_mean(quirks):
        b = scn.SceneBuilder()
        b.set_film(16, 16)
        rfw = b.set_camera_look_at(lib, (0.0, 1.2, 4.0), (0.0, 0.0, 0.3), (0, 1, 0), 30.0)
        p, vi = scenes._quad((-40, 0, -40), (-40, 0, 40), (40, 0, 40), (40, 0, -40))
        b.add_mesh(scenes._to_render(p, rfw), vi, b.material_diffuse(0.5))
        n = 24  # a 0.5 x 0.5 emitter at height 2 as 24 x 24 x 2 triangles, facing down
        g = np.linspace(-0.25, 0.25, n + 1)
        vx, vz = np.meshgrid(g, g)
        verts = np.stack([vx.ravel(), np.full(vx.size, 2.0), vz.ravel()], 1).astype(np.float32)
        idx = lambda i, j: i * (n + 1) + j
        tris = []
        for i in range(n):
            for j in range(n):
                tris += [[idx(i, j), idx(i, j + 1), idx(i + 1, j + 1)], [idx(i, j), idx(i + 1, j + 1), idx(i + 1, j)]]
        b.add_mesh(scenes._to_render(verts, rfw), np.array(tris, np.uint32), b.material_diffuse(0.0), emission=scenes.blackbody_dense(6500.0), emission_scale=5.0)
        desc, _ = b.build(lib)
        o = oracle_py.Oracle(desc)
        f, _ = o.render(render.make_params(seed=2, spp=64, max_depth=1, reference_quirks=quirks), n_threads=8)
        o.close()
        return render.film_to_rgb(f)[8:, :].mean()
    on, off = floor_mean(True), floor_mean(False)
    assert off > 0 and on < 0.25 * off, (on, off)


def test_s1_against_an_estimator_that_never_samples_lights(lib):
    """BASELINE's C1 scene (S1: a sphere under a 2 x 2 quad light, seen under a LARGE solid angle from the sphere) three ways: the path integrator reference-exact, the
    same with the quirks off, and SimplePathIntegrator with sample_lights = false (integrator.rs:573-733: light is found only by BSDF-sampled rays hitting it — no light
    sampling code at all, so none of its deviations). The third is the yardstick: quirks off agrees with it, the reference-exact image is about a fifth darker — what
    sample_spherical_triangle's `e1 . e1` divisor (sampling.rs:477) and the shadowed warp (triangle.rs:639-641) cost the reference's own showcase of its C1 configuration.
    The default stays reference-exact (the drop-in contract); the switch is for a host that wants the physics."""
    sc = scenes.sphere_light(lib, 24, 24)
    o = oracle_py.Oracle(sc.desc)

    def mean(**kw):
        f, _ = o.render(render.make_params(seed=1, max_depth=5, **kw), n_threads=8)
        return float(render.film_to_rgb(f).mean())
    exact, off = mean(spp=192), mean(spp=192, reference_quirks=False)
    yardstick = mean(spp=3072, integrator="simplepath", sample_lights=False, sample_bsdf=True)
    o.close()
    assert off == pytest.approx(yardstick, rel=0.03), (off, yardstick)
    assert 0.70 * yardstick < exact < 0.90 * yardstick, (exact, yardstick)


@pytest.mark.parametrize("kind", ["skew_patch", "curved_patch", "rect_patch", "sphere"])
def test_emitter_kinds_against_an_estimator_that_never_samples_lights(lib, kind):
    """Every area-light shape over a floor and a sphere, the path integrator against SimplePathIntegrator with sample_lights = false (no light-sampling code in it). A
    rectangular patch and a sphere agree reference-exact already (the sphere's 2.90 pi only moves MIS weights). A NON-rectangular patch — planar or curved — is area-sampled
    through BilinearPatch::sample, whose edge points are interpolated along the wrong parameters (bilinear_patch.rs:549-553; pdf(): :627-628): reference-exact such an
    emitter delivers 40-50 % of its light; with the quirks off (PBRT-v4's lerp(v, p00, p01), lerp(v, p10, p11): shm/patch.h) it agrees with the yardstick."""
    b = scn.SceneBuilder()
    b.set_film(16, 16)
    rfw = b.set_camera_look_at(lib, (0.0, 1.5, 4.5), (0.0, 0.3, 0.0), (0, 1, 0), 35.0)
    p, vi = scenes._quad((-6, 0, -6), (-6, 0, 6), (6, 0, 6), (6, 0, -6))
    b.add_mesh(scenes._to_render(p, rfw), vi, b.material_diffuse(0.6))
    rfo = np.eye(4, dtype=np.float32)
    rfo[:3, 3] = scenes._to_render(np.array([[0.8, 0.5, 0.0]], np.float32), rfw)[0]
    b.add_sphere(0.5, b.material_diffuse(0.7), render_from_object=rfo)
    black, em = b.material_diffuse(0.0), dict(emission=scenes.blackbody_dense(6500.0), emission_scale=4.0)
    if kind == "sphere":
        r2 = np.eye(4, dtype=np.float32)
        r2[:3, 3] = scenes._to_render(np.array([[-0.8, 1.6, 0.3]], np.float32), rfw)[0]
        b.add_sphere(0.4, black, render_from_object=r2, **em)
    else:
        q = {"rect_patch": [(-1.2, 1.8, -0.5), (-0.2, 1.8, -0.5), (-1.2, 1.8, 0.5), (-0.2, 1.8, 0.5)],
             "skew_patch": [(-1.2, 1.8, -0.5), (-0.1, 1.8, -0.3), (-1.0, 1.8, 0.6), (-0.4, 1.8, 0.4)],
             "curved_patch": [(-1.2, 1.8, -0.5), (-0.2, 1.6, -0.5), (-1.2, 1.5, 0.5), (-0.2, 1.9, 0.5)]}[kind]
        b.add_patch_mesh(scenes._to_render(np.array(q, np.float32), rfw), [[0, 1, 2, 3]], black, two_sided=(kind == "curved_patch"), **em)
    desc, _ = b.build(lib)
    o = oracle_py.Oracle(desc)

    def mean(**kw):
        f, _ = o.render(render.make_params(seed=1, max_depth=3, **kw), n_threads=8)
        return float(render.film_to_rgb(f).mean())
    yardstick = mean(spp=4096, integrator="simplepath", sample_lights=False, sample_bsdf=True)
    exact, off = mean(spp=256), mean(spp=256, reference_quirks=False)
    o.close()
    assert off == pytest.approx(yardstick, rel=0.03), (kind, off, yardstick)
    if kind in ("skew_patch", "curved_patch"):
        assert 0.3 * yardstick < exact < 0.6 * yardstick, (kind, exact, yardstick)
    else:
        assert exact == pytest.approx(yardstick, rel=0.03), (kind, exact, yardstick)


@pytest.mark.parametrize("name", ["crown_proxy", "textured_cornell"])
def test_feature_scenes_against_an_estimator_that_never_samples_lights(lib, name):
    """The same yardstick on whole feature scenes — glass and gold under a triangle emitter (C4's class), the textured Cornell box with its coated ceiling: with the
    quirks off the path integrator is within 1.5 % of BSDF sampling alone; reference-exact it is 3-4 % darker (the triangle emitters' sampling, as in C1 / C2 / S3)."""
    sc, depth = {"crown_proxy": (lambda: scenes.crown_proxy(lib, 15, 21, level=1, n_glass=6, n_gold=2), 8),
                 "textured_cornell": (lambda: scenes.cornell_box(lib, 20, 20, textured=True), 5)}[name]
    sc = sc()
    o = oracle_py.Oracle(sc.desc)

    def mean(**kw):
        f, _ = o.render(render.make_params(seed=1, max_depth=depth, **kw), n_threads=8)
        r = render.film_to_rgb(f)
        return float(r[np.isfinite(r).all(axis=-1)].mean())
    exact, off = mean(spp=512), mean(spp=512, reference_quirks=False)
    yardstick = mean(spp=4096, integrator="simplepath", sample_lights=False, sample_bsdf=True, reference_quirks=False)
    o.close()
    assert off == pytest.approx(yardstick, rel=0.015), (off, yardstick)
    assert 0.93 * yardstick < exact < 0.985 * yardstick, (exact, yardstick)
