"""The committed golden vectors replayed ON THE DEVICE (round 5; run with `pytest -m gpu`): the bodies of the CPU leaf tests — tests/test_oracle_golden.py (the reference's
own in-source known answers: aggregate.rs:575-702, bxdf.rs:1839-1903, float.rs, sampling.rs, interval.rs, transform.rs, vecmath ...; bitwise float32 re-evaluations;
transcendentals against float64), tests/test_leaf_golden.py (the independent re-evaluations of the rough BxDFs, triangle interaction, sphere sampling, area light, film,
camera) and tests/test_layered_golden.py (the 128 LayeredBxDF vectors) — run with a device-backed object (tests/device_leaves.py -> shm_debug_eval_leaf -> shm/probe.h
compiled by hipcc for gfx950) in the place of the oracle's library. Same vectors, same comparisons, same tolerances: what the driver's GPU run sees is the device code
against the committed expected values, not the device against the CPU oracle. The tests that need a whole scene object of the oracle (four of 46) stay CPU-only."""
import hashlib
import inspect
import json

import numpy as np
import pytest

import test_layered_golden as TY
import test_leaf_golden as TL
import test_oracle_golden as TO
from device_leaves import DeviceLeaves

pytestmark = pytest.mark.gpu

# every leaf test whose body needs nothing but leaf entry points (an `Oracle(...)` scene object is what excludes the others)
ORACLE_GOLDEN = ["test_atan2_hypot_round", "test_cie_y_integral_monte_carlo", "test_coordinate_system", "test_dielectric_sample_f_reference_vector",
                 "test_difference_of_products", "test_dot_cross", "test_fresnel_complex_against_complex128", "test_fresnel_dielectric", "test_gamma",
                 "test_intersect_p_cached", "test_intersect_triangle", "test_interval_reference_known_answers", "test_math_reference_known_answers", "test_next_float",
                 "test_offset_ray_origin", "test_rotate_from_to_reference_known_answers", "test_sample_discrete_reference_known_answers", "test_sampler_stream_properties",
                 "test_tr_d_reference_value", "test_transform_reference_known_answers", "test_triangle_light_sampling_consistency",
                 "test_triangle_sample_reference_properties", "test_trowbridge_reitz", "test_vecmath_reference_known_answers", "test_visible_wavelengths_pdf_bounds"]
LEAF_GOLDEN = ["test_area_light_l_bitwise", "test_conductor_rough_f", "test_conductor_rough_pdf_bitwise", "test_conductor_rough_sample_f", "test_dielectric_rough_f_pdf_bitwise",
               "test_dielectric_rough_sample_f", "test_film_add_sample_bitwise", "test_perspective_camera_ray_differential", "test_sphere_sample_and_pdf_with_context",
               "test_tr_sample_wm", "test_triangle_interaction_bitwise"]


@pytest.fixture(scope="module")
def dev(gpu_lib):
    return DeviceLeaves(gpu_lib, 0)


@pytest.fixture(scope="module")
def golden_vectors():
    from pathlib import Path
    return json.loads((Path(__file__).resolve().parent / "golden" / "golden.json").read_text())


def _call(fn, dev, lib, golden_vectors):
    kw = {}
    for name in inspect.signature(fn).parameters:
        kw[name] = {"orc": dev, "olib": dev, "lib": lib, "golden": golden_vectors}[name]
    fn(**kw)


@pytest.mark.parametrize("name", ORACLE_GOLDEN)
def test_reference_and_float32_vectors_on_device(dev, gpu_lib, golden_vectors, name):
    _call(getattr(TO, name), dev, gpu_lib, golden_vectors)


def test_equal_area_mappings_on_device(dev):
    """ImageInfinitelight's equal-area octahedral mapping (math.rs:456-525) — where the ENV_LIGHT kernels' look-up, sample and pdf start — on the device against the float64
    restatements of tests/test_image_light.py, the body of that file's CPU test."""
    import test_image_light as TI
    TI.test_equal_area_mappings(dev)


@pytest.mark.parametrize("name,ref,lo,hi,tol", TO.test_transcendentals.pytestmark[0].args[1])
def test_transcendentals_on_device(dev, name, ref, lo, hi, tol):
    TO.test_transcendentals(dev, name, ref, lo, hi, tol)


@pytest.mark.parametrize("name", LEAF_GOLDEN)
def test_independent_leaf_vectors_on_device(dev, gpu_lib, golden_vectors, name):
    _call(getattr(TL, name), dev, gpu_lib, golden_vectors)


def test_layered_vectors_on_device(dev):
    """all 80 f / pdf vectors and all 48 sample_f vectors of tests/golden/golden_layered.json (bxdf.rs:883-1620) through the device's LayeredBxDF"""
    for i in range(len(TY.GOLD["f_pdf"])):
        TY.test_layered_f_and_pdf(dev, i)
    for i in range(len(TY.GOLD["sample_f"])):
        TY.test_layered_sample_f(dev, i)


def test_bilinear_patch_and_sphere_intersection_on_device(dev):
    """BilinearPatch::intersect / Sphere::intersect as the traversal kernels' parked round calls them (k_trace5<., GEN>): closed forms on the device."""
    import ctypes as C
    from oracle_py import fa
    out = (C.c_float * 3)()
    # a unit square in z = 0 hit from above at (u, v) = (0.25, 0.75): t = 2
    pts = [0, 0, 0, 1, 0, 0, 0, 1, 0, 1, 1, 0]
    assert dev.orc_fn_blp_intersect(fa(*pts), fa(0.25, 0.75, 2.0), fa(0, 0, -1), float("inf"), out) == 1
    assert abs(out[0] - 0.25) < 1e-6 and abs(out[1] - 0.75) < 1e-6 and abs(out[2] - 2.0) < 1e-6
    assert dev.orc_fn_blp_intersect(fa(*pts), fa(0.25, 0.75, 2.0), fa(0, 0, -1), 1.5, out) == 0   # beyond t_max
    assert dev.orc_fn_blp_intersect(fa(*pts), fa(1.25, 0.75, 2.0), fa(0, 0, -1), float("inf"), out) == 0  # beside the patch
    # the reference's sphere known answer (aggregate.rs:575-629): a unit sphere at the origin from z = 5 along -z: t = 4
    s = TL.make_sphere((0.0, 0.0, 0.0), 1.0)
    r, w = dev._run("sphere_intersect", [0, 0, 0x40A00000, 0, 0, 0xBF800000, 0x7F800000] + __import__("device_leaves")._struct_words(s), 5)
    assert r == 1 and w.view(np.float32)[0] == np.float32(4.0)


def test_gpu_films_match_the_committed_hashes(gpu_lib):
    """tests/golden/films.json pins the f64 film of 15 small scenes that together reach every feature of the path (sha256; tests/test_golden_films.py checks the oracle
    against it on the CPU): the SAME hashes checked against the GPU's films — the device against committed fixtures, without the oracle in between."""
    import test_golden_films as TF
    from shimmer_amd import render, scenes
    want = json.loads(TF.FILMS.read_text())
    cases = TF.cases(scenes, gpu_lib)
    assert set(want) == set(cases)
    for name, (make_scene, kw) in cases.items():
        sc = make_scene()
        r = render.Renderer(gpu_lib, sc.desc, 0)
        try:
            film, stats = r.render(render.make_params(**kw))
        finally:
            r.close()
        assert int(stats["rays_closest"]) == want[name]["rays_closest"] and int(stats["rays_any"]) == want[name]["rays_any"], name
        assert hashlib.sha256(np.ascontiguousarray(film).tobytes()).hexdigest() == want[name]["sha256"], name
