"""ctypes mirror of include/shimmer_hip.h (the C ABI of libshimmer_hip.so).

Plumbing only: struct layouts, library loading and argument marshalling. No arithmetic of the hot path
lives in Python; if the HIP library is missing this module raises — there is no CPU fallback.
"""
import ctypes as C
import os
from pathlib import Path

ROOT = Path(__file__).resolve().parent
LIB_PATH = ROOT / "csrc" / "libshimmer_hip.so"

SHM_ABI_VERSION = 8
SHM_OK = 0
SHM_DIST_ID_BYTES = 128
SHM_SHAPE_TRIANGLE, SHM_SHAPE_SPHERE, SHM_SHAPE_BILINEAR_PATCH, SHM_SHAPE_INSTANCE = 0, 1, 2, 3
SHM_SPECTRUM_CONSTANT, SHM_SPECTRUM_DENSE, SHM_SPECTRUM_PIECEWISE_LINEAR = 0, 1, 2
SHM_SPECTRUM_RGB_ALBEDO, SHM_SPECTRUM_RGB_UNBOUNDED, SHM_SPECTRUM_RGB_ILLUMINANT = 3, 4, 5
SHM_MATERIAL_DIFFUSE, SHM_MATERIAL_CONDUCTOR, SHM_MATERIAL_DIELECTRIC, SHM_MATERIAL_THIN_DIELECTRIC = 0, 1, 2, 3
SHM_MATERIAL_COATED_DIFFUSE, SHM_MATERIAL_COATED_CONDUCTOR, SHM_MATERIAL_MIX = 4, 5, 6
SHM_LIGHT_POINT, SHM_LIGHT_DIFFUSE_AREA, SHM_LIGHT_UNIFORM_INFINITE, SHM_LIGHT_IMAGE_INFINITE = 0, 1, 2, 3
SHM_SPECTRUM_IMAGE_TEXTURE, SHM_SPECTRUM_TEXTURE_NODE = 6, 7
SHM_SPECTEX_LEAF, SHM_SPECTEX_SCALED, SHM_SPECTEX_MIX, SHM_SPECTEX_DIRECTION_MIX = 0, 1, 2, 3
SHM_TEXMAP_UV, SHM_TEXMAP_SPHERICAL, SHM_TEXMAP_CYLINDRICAL, SHM_TEXMAP_PLANAR = 0, 1, 2, 3
SHM_TEXFILTER_POINT, SHM_TEXFILTER_BILINEAR, SHM_TEXFILTER_TRILINEAR, SHM_TEXFILTER_EWA = 0, 1, 2, 3
SHM_WRAP_BLACK, SHM_WRAP_CLAMP, SHM_WRAP_REPEAT, SHM_WRAP_OCTAHEDRAL_SPHERE = 0, 1, 2, 3
SHM_SPECTRUM_TYPE_ALBEDO, SHM_SPECTRUM_TYPE_UNBOUNDED, SHM_SPECTRUM_TYPE_ILLUMINANT = 0, 1, 2
SHM_FLOATTEX_CONSTANT, SHM_FLOATTEX_SCALED, SHM_FLOATTEX_MIX, SHM_FLOATTEX_DIRECTION_MIX, SHM_FLOATTEX_IMAGE = 0, 1, 2, 3, 4
(SHM_FLOATSLOT_DISPLACEMENT, SHM_FLOATSLOT_U_ROUGHNESS, SHM_FLOATSLOT_V_ROUGHNESS, SHM_FLOATSLOT_U2_ROUGHNESS,
 SHM_FLOATSLOT_V2_ROUGHNESS, SHM_FLOATSLOT_THICKNESS, SHM_FLOATSLOT_G, SHM_FLOATSLOT_MIX_AMOUNT) = range(8)
SHM_CAMERA_PERSPECTIVE, SHM_CAMERA_ORTHOGRAPHIC = 0, 1
SHM_INTEGRATOR_PATH, SHM_INTEGRATOR_SIMPLE_PATH, SHM_INTEGRATOR_RANDOM_WALK = 0, 1, 2

c_float_p = C.POINTER(C.c_float)
c_u32_p = C.POINTER(C.c_uint32)


class ShmBvhNode(C.Structure):
    _fields_ = [("bmin", C.c_float * 3), ("bmax", C.c_float * 3), ("offset", C.c_uint32), ("n_prims", C.c_uint16),
                ("axis", C.c_uint8), ("pad", C.c_uint8)]


class ShmTriangleMesh(C.Structure):
    _fields_ = [("n_triangles", C.c_uint32), ("n_vertices", C.c_uint32), ("vertex_indices", c_u32_p), ("p", c_float_p),
                ("n", c_float_p), ("s", c_float_p), ("uv", c_float_p), ("reverse_orientation", C.c_uint8),
                ("transform_swaps_handedness", C.c_uint8), ("pad", C.c_uint8 * 6)]


class ShmBilinearPatchMesh(C.Structure):
    _fields_ = [("n_patches", C.c_uint32), ("n_vertices", C.c_uint32), ("vertex_indices", c_u32_p), ("p", c_float_p),
                ("n", c_float_p), ("uv", c_float_p), ("reverse_orientation", C.c_uint8),
                ("transform_swaps_handedness", C.c_uint8), ("pad", C.c_uint8 * 6)]


class ShmSphere(C.Structure):
    _fields_ = [("radius", C.c_float), ("z_min", C.c_float), ("z_max", C.c_float), ("theta_z_min", C.c_float),
                ("theta_z_max", C.c_float), ("phi_max", C.c_float), ("render_from_object", C.c_float * 16),
                ("object_from_render", C.c_float * 16), ("reverse_orientation", C.c_uint8),
                ("transform_swaps_handedness", C.c_uint8), ("pad", C.c_uint8 * 6)]


class ShmPrimitive(C.Structure):
    _fields_ = [("shape_kind", C.c_uint32), ("shape_index", C.c_uint32), ("material", C.c_uint32), ("area_light", C.c_int32)]


class ShmSpectrum(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("c", C.c_float), ("offset", C.c_uint32), ("n", C.c_uint32),
                ("lambda_min", C.c_int32), ("rgb_c", C.c_float * 3)]


class ShmMaterial(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("has_displacement", C.c_uint32), ("displacement", C.c_float),
                ("remap_roughness", C.c_uint32), ("u_roughness", C.c_float), ("v_roughness", C.c_float),
                ("u2_roughness", C.c_float), ("v2_roughness", C.c_float), ("thickness", C.c_float), ("g", C.c_float),
                ("max_depth", C.c_int32), ("n_samples", C.c_int32), ("conductor_from_reflectance", C.c_uint32),
                ("mix_material", C.c_uint32 * 2), ("mix_amount", C.c_float), ("a", ShmSpectrum), ("b", ShmSpectrum), ("c", ShmSpectrum), ("d", ShmSpectrum),
                ("float_tex", C.c_uint32 * 8), ("normal_map", C.c_uint32), ("pad", C.c_uint32 * 3)]


class ShmInstance(C.Structure):
    _fields_ = [("render_from_primitive", C.c_float * 16), ("primitive_from_render", C.c_float * 16), ("root_node", C.c_uint32),
                ("pad", C.c_uint32 * 3)]


class ShmSpectrumTexture(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("a", C.c_uint32), ("b", C.c_uint32), ("f", C.c_uint32), ("dir", C.c_float * 3), ("pad", C.c_uint32),
                ("leaf", ShmSpectrum)]


class ShmFloatTexture(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("value", C.c_float), ("a", C.c_uint32), ("b", C.c_uint32), ("c", C.c_uint32),
                ("dir", C.c_float * 3), ("image", C.c_uint32), ("pad", C.c_uint32 * 3)]


class ShmLight(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("primitive", C.c_uint32), ("scale", C.c_float), ("two_sided", C.c_uint32),
                ("position", C.c_float * 3), ("area", C.c_float), ("spectrum", ShmSpectrum)]


class ShmCamera(C.Structure):
    _fields_ = [("camera_from_raster", C.c_float * 16), ("render_from_camera", C.c_float * 16), ("dx_camera", C.c_float * 3),
                ("dy_camera", C.c_float * 3), ("lens_radius", C.c_float), ("focal_distance", C.c_float),
                ("shutter_open", C.c_float), ("shutter_close", C.c_float), ("kind", C.c_uint32), ("pad", C.c_uint32),
                ("camera_from_render", C.c_float * 16), ("min_pos_differential_x", C.c_float * 3),
                ("min_pos_differential_y", C.c_float * 3), ("min_dir_differential_x", C.c_float * 3),
                ("min_dir_differential_y", C.c_float * 3)]


class ShmImageLevel(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("texel_offset", C.c_uint32), ("pad", C.c_uint32)]


class ShmImageTexture(C.Structure):
    _fields_ = [("mapping", C.c_uint32), ("su", C.c_float), ("sv", C.c_float), ("du", C.c_float), ("dv", C.c_float),
                ("vs", C.c_float * 3), ("vt", C.c_float * 3), ("texture_from_render", C.c_float * 16), ("filter", C.c_uint32),
                ("max_anisotropy", C.c_float), ("wrap", C.c_uint32), ("scale", C.c_float), ("invert", C.c_uint8),
                ("spectrum_type", C.c_uint8), ("n_channels", C.c_uint8), ("has_color_space", C.c_uint8),
                ("first_level", C.c_uint32), ("n_levels", C.c_uint32)]


class ShmImageInfiniteLight(C.Structure):
    _fields_ = [("render_from_light", C.c_float * 16), ("light_from_render", C.c_float * 16), ("image_level", C.c_uint32),
                ("pad", C.c_uint32)]


class ShmPlyMesh(C.Structure):
    _fields_ = [("n_vertices", C.c_uint32), ("n_tri_indices", C.c_uint32), ("n_quad_indices", C.c_uint32), ("n_face_indices", C.c_uint32),
                ("p", c_float_p), ("n", c_float_p), ("uv", c_float_p), ("tri_indices", C.POINTER(C.c_int32)),
                ("quad_indices", C.POINTER(C.c_int32)), ("face_indices", C.POINTER(C.c_int32))]


class ShmColorSpace(C.Structure):
    _fields_ = [("rgb2spec_res", C.c_uint32), ("pad", C.c_uint32), ("rgb2spec_scale", c_float_p), ("rgb2spec_data", c_float_p),
                ("illuminant", c_float_p)]


class ShmFilm(C.Structure):
    _fields_ = [("pixel_bounds", C.c_int32 * 4), ("full_resolution", C.c_int32 * 2), ("filter_radius", C.c_float * 2),
                ("imaging_ratio", C.c_float), ("max_component_value", C.c_float), ("sensor_r_bar", c_float_p),
                ("sensor_g_bar", c_float_p), ("sensor_b_bar", c_float_p)]


class ShmSceneDesc(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("n_nodes", C.c_uint32), ("nodes", C.POINTER(ShmBvhNode)),
                ("n_primitives", C.c_uint32), ("primitives", C.POINTER(ShmPrimitive)), ("n_meshes", C.c_uint32),
                ("meshes", C.POINTER(ShmTriangleMesh)), ("n_spheres", C.c_uint32), ("spheres", C.POINTER(ShmSphere)),
                ("n_materials", C.c_uint32), ("materials", C.POINTER(ShmMaterial)), ("n_lights", C.c_uint32),
                ("lights", C.POINTER(ShmLight)), ("n_spectrum_floats", C.c_uint32), ("spectrum_data", c_float_p),
                ("camera", ShmCamera), ("film", ShmFilm), ("n_patch_meshes", C.c_uint32), ("pad", C.c_uint32),
                ("patch_meshes", C.POINTER(ShmBilinearPatchMesh)),
                ("n_image_textures", C.c_uint32), ("n_image_levels", C.c_uint32), ("image_textures", C.POINTER(ShmImageTexture)),
                ("image_levels", C.POINTER(ShmImageLevel)), ("n_texel_floats", C.c_uint64), ("texel_data", c_float_p),
                ("color_space", ShmColorSpace), ("ewa_filter_lut", c_float_p), ("n_image_lights", C.c_uint32), ("n_float_textures", C.c_uint32),
                ("image_lights", C.POINTER(ShmImageInfiniteLight)), ("float_textures", C.POINTER(ShmFloatTexture)),
                ("n_spectrum_textures", C.c_uint32), ("n_instances", C.c_uint32), ("spectrum_textures", C.POINTER(ShmSpectrumTexture)),
                ("instances", C.POINTER(ShmInstance))]


class ShmRenderParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("samples_per_pixel", C.c_int32), ("max_depth", C.c_int32), ("regularize", C.c_uint8),
                ("disable_pixel_jitter", C.c_uint8), ("disable_wavelength_jitter", C.c_uint8), ("force_diffuse", C.c_uint8),
                ("integrator", C.c_uint8), ("sample_lights", C.c_uint8), ("sample_bsdf", C.c_uint8), ("disable_texture_filtering", C.c_uint8),
                ("disable_reference_quirks", C.c_uint8), ("pad", C.c_uint8 * 7)]


class ShmTile(C.Structure):
    _fields_ = [("x0", C.c_int32), ("y0", C.c_int32), ("x1", C.c_int32), ("y1", C.c_int32)]


class ShmFilmPixel(C.Structure):
    _fields_ = [("rgb_sum", C.c_double * 3), ("weight_sum", C.c_double)]


class ShmStats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("rays_closest", C.c_uint64), ("rays_any", C.c_uint64), ("nodes_closest", C.c_uint64),
                ("tris_closest", C.c_uint64), ("nodes_any", C.c_uint64), ("tris_any", C.c_uint64), ("ms_total", C.c_double),
                ("ms_trace_closest", C.c_double), ("ms_trace_any", C.c_double), ("ms_shade", C.c_double),
                ("launches_closest", C.c_uint32), ("launches_any", C.c_uint32), ("ms_gather", C.c_double), ("gather_bytes", C.c_uint64)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class ShmRay(C.Structure):
    _fields_ = [("o", C.c_float * 3), ("d", C.c_float * 3), ("t_max", C.c_float), ("pad", C.c_float)]


class ShmHit(C.Structure):
    _fields_ = [("prim", C.c_int32), ("t", C.c_float), ("b0", C.c_float), ("b1", C.c_float), ("b2", C.c_float),
                ("phi", C.c_float), ("instance", C.c_uint32), ("pad", C.c_uint32)]


assert C.sizeof(ShmMaterial) == 64 + 4 * 32 + 48 and C.sizeof(ShmFloatTexture) == 48 and C.sizeof(ShmBvhNode) == 32 and C.sizeof(ShmRay) == 32 and C.sizeof(ShmHit) == 32 and C.sizeof(ShmFilmPixel) == 32

class ShmPbrtScene(C.Structure):
    _fields_ = [("desc", ShmSceneDesc), ("params", ShmRenderParams), ("integrator", C.c_char * 32), ("output_filename", C.c_char * 256), ("owner", C.c_void_p),
                ("output_rgb_from_sensor_rgb", C.c_float * 9)]


class ShmCameraParams(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("render_space", C.c_uint32), ("world_from_camera", C.c_float * 16), ("fov_deg", C.c_float),
                ("full_resolution", C.c_int32 * 2), ("lens_radius", C.c_float), ("focal_distance", C.c_float), ("frame_aspect_ratio", C.c_float),
                ("has_screen_window", C.c_uint32), ("screen_window", C.c_float * 4)]


class ShmLoadedImage(C.Structure):
    _fields_ = [("n_levels", C.c_uint32), ("n_channels", C.c_uint32), ("file_channels", C.c_uint32), ("has_color_space", C.c_uint32),
                ("n_texel_floats", C.c_uint64), ("levels", C.POINTER(ShmImageLevel)), ("texels", c_float_p)]


class ShmDistInfo(C.Structure):
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("rccl_ranks", C.c_int32), ("rccl_device", C.c_int32), ("rccl_version", C.c_int32),
                ("hip_runtime_version", C.c_int32), ("rows_per_block", C.c_int32), ("n_my_tiles", C.c_uint32), ("librccl_path", C.c_char * 256),
                ("libamdhip_path", C.c_char * 256)]

    def as_dict(self):
        return {n: (getattr(self, n).decode() if n.endswith("_path") else getattr(self, n)) for n, _ in self._fields_}


SHM_REDUCE_SUM, SHM_REDUCE_MAX, SHM_REDUCE_MIN = 0, 1, 2

# Every symbol include/shimmer_hip.h declares, with its signature (tests check the .so exports all of them).
EXPORTS = {
    "shm_scene_create": (C.c_int, [C.POINTER(ShmSceneDesc), C.c_int, C.POINTER(C.c_void_p)]),
    "shm_scene_destroy": (None, [C.c_void_p]),
    "shm_film_clear": (C.c_int, [C.c_void_p]),
    "shm_render_wave": (C.c_int, [C.c_void_p, C.POINTER(ShmRenderParams), C.POINTER(ShmTile), C.c_uint32, C.c_int32, C.c_int32,
                                  C.POINTER(ShmStats)]),
    "shm_film_read": (C.c_int, [C.c_void_p, C.c_void_p]),
    "shm_film_device_ptr": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    "shm_render_device": (C.c_int, [C.c_void_p, C.POINTER(ShmRenderParams), C.POINTER(ShmTile), C.c_uint32, C.POINTER(ShmStats)]),
    "shm_render": (C.c_int, [C.c_void_p, C.POINTER(ShmRenderParams), C.POINTER(ShmTile), C.c_uint32, C.c_void_p, C.POINTER(ShmStats)]),
    "shm_trace_closest": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(ShmStats)]),
    "shm_trace_any": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(ShmStats)]),
    "shm_trace_closest_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.POINTER(ShmStats)]),
    "shm_trace_any_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.POINTER(ShmStats)]),
    "shm_last_error": (C.c_char_p, []),
    "shm_device_count": (C.c_int, []),
    "shm_bvh_build": (C.c_int, [c_float_p, C.c_uint32, C.c_int, C.POINTER(ShmBvhNode), c_u32_p, c_u32_p]),
    "shm_bounds3_probe": (C.c_int, [c_float_p, c_float_p, c_float_p, c_float_p]),
    "shm_tile_bounds": (C.c_int, [C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.POINTER(ShmTile), c_u32_p]),
    "shm_camera_perspective": (C.c_int, [c_float_p, C.c_float, C.POINTER(C.c_int32), C.c_float, C.c_float, C.POINTER(ShmCamera), c_float_p]),
    "shm_integrator_render": (C.c_int, [C.c_char_p, C.POINTER(ShmSceneDesc), C.c_int, C.c_int32, C.c_int, C.c_int, C.c_int, C.c_int32, C.c_int32, C.c_int, C.c_int,
                                        C.c_void_p, C.POINTER(ShmStats), C.POINTER(C.c_int32)]),
    "shm_camera_orthographic": (C.c_int, [c_float_p, C.POINTER(C.c_int32), C.c_float, C.c_float, C.POINTER(ShmCamera), c_float_p]),
    "shm_film_get_image": (C.c_int, [C.c_void_p, C.c_uint64, c_float_p, C.c_int, c_float_p]),
    "shm_write_pfm": (C.c_int, [C.c_char_p, c_float_p, C.c_int32, C.c_int32]),
    "shm_ply_read": (C.c_int, [C.c_char_p, C.c_void_p]),
    "shm_ply_free": (None, [C.c_void_p]),
    "shm_shard_tiles": (C.c_int, [C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_int32, c_u32_p, c_u32_p]),
    "shm_dist_unique_id": (C.c_int, [C.c_void_p]),
    "shm_dist_init": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "shm_dist_finalize": (C.c_int, [C.c_void_p]),
    "shm_render_sharded": (C.c_int, [C.c_void_p, C.POINTER(ShmRenderParams), C.POINTER(ShmStats)]),
    "shm_dist_selftest": (C.c_int, [C.c_void_p]),
    "shm_dist_barrier": (C.c_int, [C.c_void_p]),
    "shm_dist_allreduce_f64": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_uint32, C.c_int32]),
    "shm_dist_allgather_f64": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_uint32, C.POINTER(C.c_double)]),
    "shm_dist_info": (C.c_int, [C.c_void_p, C.POINTER(ShmDistInfo)]),
    "shm_device_synchronize": (C.c_int, [C.c_int32]),
    "shm_scene_load_pbrt": (C.c_int, [C.c_char_p, C.POINTER(C.POINTER(ShmPbrtScene))]),
    "shm_scene_parse_pbrt": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(C.POINTER(ShmPbrtScene))]),
    "shm_pbrt_free": (None, [C.POINTER(ShmPbrtScene)]),
    "shm_pbrt_tokenize": (C.c_int, [C.c_char_p, C.c_char_p, C.c_uint64, c_u32_p]),
    "shm_pbrt_parse_params": (C.c_int, [C.c_char_p, C.c_char_p, C.c_uint64]),
    "shm_blackbody_dense": (C.c_int, [C.c_float, c_float_p]),
    "shm_look_at": (C.c_int, [c_float_p, c_float_p, c_float_p, c_float_p]),
    "shm_image_load_png": (C.c_int, [C.c_char_p, C.c_char_p, C.c_uint32, C.c_int, C.POINTER(ShmLoadedImage)]),
    "shm_image_free": (None, [C.POINTER(ShmLoadedImage)]),
    "shm_camera_create": (C.c_int, [C.POINTER(ShmCameraParams), C.POINTER(ShmCamera), c_float_p]),
    "shm_render_multi": (C.c_int, [C.POINTER(ShmSceneDesc), C.POINTER(C.c_int32), C.c_int32, C.POINTER(ShmRenderParams), C.c_void_p, C.POINTER(ShmStats)]),
}

_lib = None


class ShimmerHipError(RuntimeError):
    pass


def load_library(path=None):
    """Load libshimmer_hip.so (built in-tree by __graft_entry__.build()). Raises if it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    if path is None and os.environ.get("SHM_LIB"):  # development hook: A/B a differently compiled build of the same source
        path = os.environ["SHM_LIB"]
    p = Path(path) if path else LIB_PATH
    if not p.exists():
        raise ShimmerHipError(f"{p} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950). There is no CPU fallback for the render path.")
    lib = C.CDLL(str(p), mode=getattr(os, "RTLD_NOW", 2))
    host_only = os.environ.get("SHM_HOST_ONLY") == "1"  # tests/test_sanitizers.py: the host mirror linked on its own (no device entry points)
    for name, (res, args) in EXPORTS.items():
        if host_only and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


PROBE_LIB_PATH = LIB_PATH.parent / "libshimmer_hip_probe.so"
_probe = None


def load_probe_library():
    """The TEST library libshimmer_hip_probe.so (include/shimmer_hip_probe.h): shm_debug_eval_leaf, one leaf function of the shared arithmetic on the device. Only the
    `-m gpu` suite loads it (tests/device_leaves.py); it is not part of the product library."""
    global _probe
    if _probe is None:
        if not PROBE_LIB_PATH.exists():
            raise ShimmerHipError(f"{PROBE_LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = C.CDLL(str(PROBE_LIB_PATH), mode=getattr(os, "RTLD_NOW", 2))
        lib.shm_debug_eval_leaf.restype = C.c_int
        lib.shm_debug_eval_leaf.argtypes = [C.c_int, C.c_int, c_u32_p, C.c_uint32, c_u32_p, C.c_uint32, C.POINTER(C.c_int)]
        lib.shm_probe_last_error.restype = C.c_char_p
        _probe = lib
    return _probe


def check(lib, rc, what):
    if rc != SHM_OK:
        msg = lib.shm_last_error()
        raise ShimmerHipError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")
