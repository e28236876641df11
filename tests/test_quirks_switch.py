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
    quirks off the same sample is finite, and the other 63 pixels of the tile do not change (the guard only fires on such samples)."""
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
    same = (a.view(np.uint64) == b.view(np.uint64)).all(axis=-1)
    assert same.sum() == 63 and not same[y - y0, x - x0]


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
    """S1 (sphere + quad light) has no sphere EMITTER, so Sphere::pdf_with_context is not reached and the film is the same; three spheres
    under a uniform sky lit through SimplePath's uniform sampling change with uniform_hemisphere_pdf (1/4pi -> 1/2pi halves those terms)."""
    s1 = scenes.sphere_light(lib, 32, 32)
    o = oracle_py.Oracle(s1.desc)
    a, _ = o.render(render.make_params(seed=1, spp=4), n_threads=4)
    b, _ = o.render(render.make_params(seed=1, spp=4, reference_quirks=False), n_threads=4)
    o.close()
    assert np.isfinite(b["rgb_sum"]).all() and np.array_equal(a, b)
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
