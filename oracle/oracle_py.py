"""ctypes loader for the CPU oracle (oracle/oracle.cpp).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
The product package (shimmer_amd/) never does.
"""
import ctypes as C
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT.parent))
from shimmer_amd import abi  # noqa: E402  (struct layouts of include/shimmer_hip.h only)
from shimmer_amd.render import FILM_DTYPE, HIT_DTYPE  # noqa: E402
from shimmer_amd.scene import tiles_for, wave_schedule  # noqa: E402

LIB_PATH = ROOT / "_build" / "liboracle.so"
_lib = None

F = C.c_float
FP = C.POINTER(C.c_float)


def load():
    global _lib
    if _lib is not None:
        return _lib
    import os
    path = os.environ.get("ORACLE_LIB") or str(LIB_PATH)  # tests/test_sanitizers.py points this at the ASan / UBSan build
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", str(ROOT)])
    lib = C.CDLL(path)
    lib.orc_last_error.restype = C.c_char_p
    lib.orc_scene_create.argtypes = [C.POINTER(abi.ShmSceneDesc), C.POINTER(C.c_void_p)]
    lib.orc_scene_destroy.argtypes = [C.c_void_p]
    lib.orc_scene_destroy.restype = None
    lib.orc_trace_closest.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(abi.ShmStats)]
    lib.orc_trace_any.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(abi.ShmStats)]
    lib.orc_render_wave.argtypes = [C.c_void_p, C.POINTER(abi.ShmRenderParams), C.POINTER(abi.ShmTile), C.c_uint32, C.c_int32, C.c_int32,
                                    C.c_int, C.c_void_p, C.POINTER(abi.ShmStats)]
    lib.orc_render.argtypes = [C.c_void_p, C.POINTER(abi.ShmRenderParams), C.POINTER(abi.ShmTile), C.c_uint32, C.c_int, C.c_void_p,
                               C.POINTER(abi.ShmStats)]
    for name in ["next_float_up", "next_float_down", "sin", "cos", "asin", "acos", "exp", "log", "atanh", "cosh", "round",
                 "sample_visible_wavelengths", "visible_wavelengths_pdf"]:
        fn = getattr(lib, "orc_fn_" + name)
        fn.restype, fn.argtypes = F, [F]
    for name in ["atan2", "hypot", "fresnel_dielectric", "power_heuristic"]:
        fn = getattr(lib, "orc_fn_" + name)
        fn.restype, fn.argtypes = F, [F, F]
    lib.orc_fn_gamma.restype, lib.orc_fn_gamma.argtypes = F, [C.c_int]
    lib.orc_fn_difference_of_products.restype, lib.orc_fn_difference_of_products.argtypes = F, [F, F, F, F]
    lib.orc_fn_lerp.restype, lib.orc_fn_lerp.argtypes = F, [F, F, F]
    lib.orc_fn_poly3.restype, lib.orc_fn_poly3.argtypes = F, [F, F, F, F]
    lib.orc_fn_fresnel_complex.restype, lib.orc_fn_fresnel_complex.argtypes = F, [F, F, F]
    lib.orc_fn_dot.restype, lib.orc_fn_dot.argtypes = F, [FP, FP]
    lib.orc_fn_cross.restype, lib.orc_fn_cross.argtypes = None, [FP, FP, FP]
    lib.orc_fn_coordinate_system.restype, lib.orc_fn_coordinate_system.argtypes = None, [FP, FP]
    lib.orc_fn_intersect_p_cached.restype, lib.orc_fn_intersect_p_cached.argtypes = C.c_int, [FP, FP, FP, FP, F]
    lib.orc_fn_intersect_triangle.restype, lib.orc_fn_intersect_triangle.argtypes = C.c_int, [FP, FP, F, FP, FP, FP, FP]
    lib.orc_fn_tr_d.restype, lib.orc_fn_tr_d.argtypes = F, [F, F, FP]
    lib.orc_fn_tr_g.restype, lib.orc_fn_tr_g.argtypes = F, [F, F, FP, FP]
    lib.orc_fn_tr_lambda.restype, lib.orc_fn_tr_lambda.argtypes = F, [F, F, FP]
    lib.orc_fn_tr_sample_wm.restype, lib.orc_fn_tr_sample_wm.argtypes = None, [F, F, FP, FP, FP]
    lib.orc_fn_bxdf_sample_f.restype = C.c_int
    lib.orc_fn_bxdf_sample_f.argtypes = [C.c_int, FP, FP, F, F, F, FP, F, FP, FP]
    lib.orc_fn_bxdf_f_pdf.restype, lib.orc_fn_bxdf_f_pdf.argtypes = None, [C.c_int, FP, FP, F, F, F, FP, FP, FP]
    IP = C.POINTER(C.c_int)
    lib.orc_fn_layered_f_pdf.restype, lib.orc_fn_layered_f_pdf.argtypes = None, [C.c_int, FP, IP, FP, FP, FP]
    lib.orc_fn_layered_sample_f.restype, lib.orc_fn_layered_sample_f.argtypes = C.c_int, [C.c_int, FP, IP, FP, F, FP, FP]
    lib.orc_fn_layered_sample_f_steps.restype, lib.orc_fn_layered_sample_f_steps.argtypes = C.c_int, [C.c_int, FP, IP, FP, F, FP, C.c_int, FP, IP]
    lib.orc_fn_henyey_greenstein.restype, lib.orc_fn_henyey_greenstein.argtypes = F, [F, F]
    lib.orc_fn_sample_henyey_greenstein.restype, lib.orc_fn_sample_henyey_greenstein.argtypes = None, [FP, F, FP, FP]
    lib.orc_fn_sample_exponential.restype, lib.orc_fn_sample_exponential.argtypes = F, [F, F]
    lib.orc_fn_blp_info.restype, lib.orc_fn_blp_info.argtypes = None, [FP, FP]
    lib.orc_fn_blp_interaction_attr.restype, lib.orc_fn_blp_interaction_attr.argtypes = None, [FP, C.c_int, FP, FP, F, F, FP, FP]
    lib.orc_fn_invert_bilinear.restype, lib.orc_fn_invert_bilinear.argtypes = None, [FP, FP, FP]
    lib.orc_fn_interval_op.restype, lib.orc_fn_interval_op.argtypes = None, [C.c_int, F, F, F, F, FP]
    lib.orc_fn_det3.restype, lib.orc_fn_det3.argtypes = F, [FP]
    lib.orc_fn_log2.restype, lib.orc_fn_log2.argtypes = F, [F]
    lib.orc_fn_texture_map.restype, lib.orc_fn_texture_map.argtypes = None, [C.c_void_p, C.c_uint32, FP, FP]
    lib.orc_fn_texture_filter.restype, lib.orc_fn_texture_filter.argtypes = None, [C.c_void_p, C.c_uint32, FP, FP, FP, FP]
    lib.orc_fn_rgb2spec_fetch.restype, lib.orc_fn_rgb2spec_fetch.argtypes = None, [C.c_void_p, FP, FP]
    lib.orc_fn_image_texture_evaluate.restype, lib.orc_fn_image_texture_evaluate.argtypes = None, [C.c_void_p, C.c_uint32, FP, FP, FP]
    lib.orc_fn_approximate_dp_dxy.restype, lib.orc_fn_approximate_dp_dxy.argtypes = None, [C.c_void_p, FP, FP, C.c_int, C.c_int, FP]
    lib.orc_fn_camera_hit_differentials.restype = C.c_int
    lib.orc_fn_camera_hit_differentials.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int, FP]
    lib.orc_fn_spawn_ray_differentials.restype = C.c_int
    lib.orc_fn_spawn_ray_differentials.argtypes = [FP] * 8 + [C.c_uint32, F, FP]
    lib.orc_fn_equal_area_square_to_sphere.restype, lib.orc_fn_equal_area_square_to_sphere.argtypes = None, [FP, FP]
    lib.orc_fn_equal_area_sphere_to_square.restype, lib.orc_fn_equal_area_sphere_to_square.argtypes = None, [FP, FP]
    lib.orc_fn_light_sample_li.restype, lib.orc_fn_light_sample_li.argtypes = C.c_int, [C.c_void_p, C.c_uint32, FP, C.c_int, FP, FP]
    lib.orc_fn_image_light_pdf.restype, lib.orc_fn_image_light_pdf.argtypes = F, [C.c_void_p, C.c_uint32, FP, C.c_int]
    lib.orc_fn_infinite_light_le.restype, lib.orc_fn_infinite_light_le.argtypes = None, [C.c_void_p, C.c_uint32, FP, FP, FP]
    lib.orc_fn_image_light_distribution.restype = C.c_int
    lib.orc_fn_image_light_distribution.argtypes = [C.c_void_p, C.c_uint32, C.c_int, FP, FP, FP]
    lib.orc_fn_float_texture_evaluate.restype, lib.orc_fn_float_texture_evaluate.argtypes = F, [C.c_void_p, C.c_uint32, FP]
    lib.orc_fn_bump_or_normal_map.restype, lib.orc_fn_bump_or_normal_map.argtypes = None, [C.c_void_p, C.c_int, C.c_uint32, FP, FP, FP, FP]
    lib.orc_fn_spectrum_texture_evaluate.restype = None
    lib.orc_fn_spectrum_texture_evaluate.argtypes = [C.c_void_p, C.POINTER(abi.ShmSpectrum), FP, FP, FP]
    lib.orc_fn_transform_apply.restype, lib.orc_fn_transform_apply.argtypes = None, [C.c_int, C.c_int, FP, FP, FP, FP]
    lib.orc_fn_vecmath.restype, lib.orc_fn_vecmath.argtypes = None, [FP, FP, FP]
    lib.orc_fn_scene_radius.restype, lib.orc_fn_scene_radius.argtypes = F, [C.c_void_p]
    lib.orc_fn_rotate_from_to.restype, lib.orc_fn_rotate_from_to.argtypes = None, [FP, FP, FP, FP]
    lib.orc_fn_blp_intersect.restype, lib.orc_fn_blp_intersect.argtypes = C.c_int, [FP, FP, FP, F, FP]
    lib.orc_fn_blp_interaction.restype, lib.orc_fn_blp_interaction.argtypes = None, [FP, C.c_int, F, F, FP, FP]
    lib.orc_fn_blp_sample_with_context.restype, lib.orc_fn_blp_sample_with_context.argtypes = C.c_int, [FP, C.c_int, FP, FP, FP, FP, FP]
    lib.orc_fn_blp_pdf_with_context.restype, lib.orc_fn_blp_pdf_with_context.argtypes = F, [FP, C.c_int, FP, FP, FP, FP]
    lib.orc_fn_spherical_quad_area.restype, lib.orc_fn_spherical_quad_area.argtypes = F, [FP, FP, FP, FP]
    lib.orc_fn_sample_spherical_rectangle.restype, lib.orc_fn_sample_spherical_rectangle.argtypes = None, [FP, FP, FP, FP, FP, FP]
    lib.orc_fn_invert_spherical_rectangle_sample.restype, lib.orc_fn_invert_spherical_rectangle_sample.argtypes = None, [FP, FP, FP, FP, FP, FP]
    lib.orc_fn_quadratic.restype, lib.orc_fn_quadratic.argtypes = C.c_int, [F, F, F, FP]
    lib.orc_fn_sample_discrete.restype, lib.orc_fn_sample_discrete.argtypes = C.c_int, [FP, C.c_int, F, FP, FP]
    lib.orc_fn_sample_cosine_hemisphere.restype, lib.orc_fn_sample_cosine_hemisphere.argtypes = None, [FP, FP]
    lib.orc_fn_sampler_stream.restype, lib.orc_fn_sampler_stream.argtypes = F, [C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, FP]
    lib.orc_fn_offset_ray_origin.restype, lib.orc_fn_offset_ray_origin.argtypes = None, [FP, FP, FP, FP, FP]
    lib.orc_fn_triangle_sample_with_context.restype = C.c_int
    lib.orc_fn_triangle_sample_with_context.argtypes = [FP] * 8
    lib.orc_fn_triangle_pdf_with_context.restype, lib.orc_fn_triangle_pdf_with_context.argtypes = F, [FP] * 7
    lib.orc_fn_hit_interaction.restype = C.c_int
    lib.orc_fn_hit_interaction.argtypes = [C.c_void_p, C.c_void_p, FP, C.c_void_p]
    lib.orc_fn_camera_ray.restype, lib.orc_fn_camera_ray.argtypes = None, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint64, FP]
    lib.orc_fn_film_sample_rgb.restype, lib.orc_fn_film_sample_rgb.argtypes = None, [C.c_void_p, FP, FP, FP, FP]
    lib.orc_fn_spectrum_get.restype, lib.orc_fn_spectrum_get.argtypes = F, [C.c_void_p, C.POINTER(abi.ShmSpectrum), F]
    lib.orc_fn_spectrum_sample.restype, lib.orc_fn_spectrum_sample.argtypes = None, [C.c_void_p, C.POINTER(abi.ShmSpectrum), FP, FP]
    lib.orc_render_reference_stream.argtypes = [C.c_void_p, C.POINTER(abi.ShmRenderParams), C.POINTER(abi.ShmTile), C.c_uint32, C.c_void_p, C.POINTER(abi.ShmStats),
                                                C.POINTER(C.c_uint64)]
    lib.orc_fn_xoshiro256pp.restype, lib.orc_fn_xoshiro256pp.argtypes = None, [C.POINTER(C.c_uint64), C.c_int, C.POINTER(C.c_uint64)]
    lib.orc_fn_splitmix64.restype, lib.orc_fn_splitmix64.argtypes = None, [C.c_uint64, C.c_int, C.POINTER(C.c_uint64)]
    lib.orc_fn_reference_stream_f32.restype, lib.orc_fn_reference_stream_f32.argtypes = None, [C.c_uint64, C.c_int, FP, C.POINTER(C.c_uint64)]
    _lib = lib
    return lib


def fa(*vals):
    """float array helper for the orc_fn_* entry points."""
    return (C.c_float * len(vals))(*[float(v) for v in vals])


class Oracle:
    def __init__(self, desc):
        self.lib = load()
        self.desc = desc
        self.handle = C.c_void_p()
        rc = self.lib.orc_scene_create(C.byref(desc), C.byref(self.handle))
        if rc != 0:
            raise RuntimeError(f"orc_scene_create failed ({rc}): {self.lib.orc_last_error().decode()}")
        pb = desc.film.pixel_bounds
        self.pixel_bounds = (pb[0], pb[1], pb[2], pb[3])
        self.width, self.height = pb[2] - pb[0], pb[3] - pb[1]
        self._tiles = None

    def close(self):
        if self.handle:
            self.lib.orc_scene_destroy(self.handle)
            self.handle = C.c_void_p()

    def tiles(self, host_lib):
        if self._tiles is None:
            self._tiles = tiles_for(host_lib, self.pixel_bounds)
        return self._tiles

    def render(self, params, n_threads=1, tiles=None, n_tiles=None, waves=None, film=None):
        if tiles is None:
            from shimmer_amd import abi as _abi
            tiles, n_tiles = self.tiles(_abi.load_library())
        if film is None:
            film = np.zeros((self.height, self.width), dtype=FILM_DTYPE)
        stats = abi.ShmStats()
        for (ws, we) in (waves if waves is not None else wave_schedule(params.samples_per_pixel)):
            rc = self.lib.orc_render_wave(self.handle, C.byref(params), tiles, n_tiles, ws, we, n_threads, film.ctypes.data_as(C.c_void_p), C.byref(stats))
            if rc != 0:
                raise RuntimeError(f"orc_render_wave failed ({rc}): {self.lib.orc_last_error().decode()}")
        return film, stats.as_dict()

    def render_reference_stream(self, params, tiles=None, n_tiles=None):
        """ImageTileIntegrator::render on the reference's OWN sampler stream, as `RAYON_NUM_THREADS=1 shimmer scene.pbrt --seed S` draws it (oracle.cpp,
        orc_render_reference_stream; scenes without LayeredBxDF / MixMaterial). Returns (film, stats, number of u64 draws)."""
        if tiles is None:
            from shimmer_amd import abi as _abi
            tiles, n_tiles = self.tiles(_abi.load_library())
        film = np.zeros((self.height, self.width), dtype=FILM_DTYPE)
        stats, draws = abi.ShmStats(), C.c_uint64()
        rc = self.lib.orc_render_reference_stream(self.handle, C.byref(params), tiles, n_tiles, film.ctypes.data_as(C.c_void_p), C.byref(stats), C.byref(draws))
        if rc != 0:
            raise RuntimeError(f"orc_render_reference_stream failed ({rc}): {self.lib.orc_last_error().decode()}")
        return film, stats.as_dict(), draws.value

    def trace(self, rays, any_hit=False):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = rays.shape[0]
        stats = abi.ShmStats()
        if any_hit:
            out = np.zeros(n, np.uint8)
            self.lib.orc_trace_any(self.handle, rays.ctypes.data_as(C.c_void_p), n, out.ctypes.data_as(C.c_void_p), C.byref(stats))
        else:
            out = np.zeros(n, dtype=HIT_DTYPE)
            self.lib.orc_trace_closest(self.handle, rays.ctypes.data_as(C.c_void_p), n, out.ctypes.data_as(C.c_void_p), C.byref(stats))
        return out, stats.as_dict()
