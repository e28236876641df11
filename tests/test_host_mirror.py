"""Host-side mirror of the reference's scene-construction steps + C-ABI surface (no GPU needed)."""
import ctypes as C
import re
import struct
from pathlib import Path

import numpy as np
import pytest

from shimmer_amd import abi, render, scene as scn, scenes

ROOT = Path(__file__).resolve().parents[1]


def test_library_exports_every_declared_symbol(lib):
    """Every function include/shimmer_hip.h declares is exported by libshimmer_hip.so, and nothing the header
    declares is missing from the ctypes table (no compute calls here)."""
    header = (ROOT / "include" / "shimmer_hip.h").read_text()
    declared = set(re.findall(r"SHM_API\s+[\w\s\*]+?\b(shm_[a-z_0-9]+)\s*\(", header))
    assert declared == set(abi.EXPORTS), declared ^ set(abi.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.shm_last_error() is not None


def test_probe_library_exports_its_header_and_the_product_does_not(lib):
    """include/shimmer_hip_probe.h is the TEST library's ABI (libshimmer_hip_probe.so: the device leaf probe, round 6): it exports every function that header declares,
    and the product library exports none of them (no compute call here: loading a HIP library needs no device)."""
    header = (ROOT / "include" / "shimmer_hip_probe.h").read_text()
    declared = set(re.findall(r"SHM_API\s+[\w\s\*]+?\b(shm_[a-z_0-9]+)\s*\(", header))
    assert declared == {"shm_debug_eval_leaf", "shm_probe_last_error"}, declared
    probe = abi.load_probe_library()
    for name in declared:
        assert getattr(probe, name) is not None
        assert not hasattr(lib, name), name
    assert not (declared & set(abi.EXPORTS))


def test_documents_name_only_entry_points_the_headers_declare():
    """INTEGRATION.md (the binding a maintainer of the reference would write), README.md and DESIGN.md may only name `shm_*` entry points that include/*.h declares:
    the documents cannot drift from the boundary."""
    import re
    root = Path(__file__).resolve().parent.parent
    declared = set()
    for h in ("shimmer_hip.h", "shimmer_hip_probe.h"):
        declared |= set(re.findall(r"\bshm_[a-z0-9_]+\b", (root / "include" / h).read_text()))
    for doc in ("INTEGRATION.md", "README.md", "DESIGN.md"):
        named = set(re.findall(r"\bshm_[a-z0-9_]+\b", (root / doc).read_text()))
        unknown = sorted(n for n in named - declared if not n.endswith("_"))  # ("shm_dist_*": a family, written with the star)
        assert not unknown, (doc, unknown)


def test_struct_layouts_match_header(tmp_path):
    assert C.sizeof(abi.ShmBvhNode) == 32 and C.sizeof(abi.ShmPrimitive) == 16 and C.sizeof(abi.ShmSpectrum) == 32
    assert C.sizeof(abi.ShmRay) == 32 and C.sizeof(abi.ShmHit) == 32 and C.sizeof(abi.ShmFilmPixel) == 32 and C.sizeof(abi.ShmTile) == 16
    assert C.sizeof(abi.ShmMaterial) == 64 + 4 * 32 + 48 and C.sizeof(abi.ShmLight) == 32 + 32 and C.sizeof(abi.ShmRenderParams) == 32
    # every struct of include/shimmer_hip.h as the C compiler lays it out (size, and the offset of the last field) against ctypes
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no C compiler")
    names = ["ShmBvhNode", "ShmTriangleMesh", "ShmBilinearPatchMesh", "ShmSphere", "ShmPrimitive", "ShmSpectrum", "ShmFloatTexture", "ShmSpectrumTexture", "ShmPlyMesh",
             "ShmMaterial", "ShmLight", "ShmImageLevel", "ShmImageTexture", "ShmColorSpace", "ShmImageInfiniteLight", "ShmCamera", "ShmFilm",
             "ShmSceneDesc", "ShmRenderParams", "ShmTile", "ShmFilmPixel", "ShmStats", "ShmRay", "ShmHit", "ShmDistInfo"]
    last = {n: getattr(abi, n)._fields_[-1][0] for n in names}
    src = "#include <stdio.h>\n#include <stddef.h>\n#include \"shimmer_hip.h\"\nint main(void) {\n" + "".join(
        f'  printf("{n} %zu %zu\\n", sizeof({n}), offsetof({n}, {last[n]}));\n' for n in names) + "  return 0;\n}\n"
    (tmp_path / "probe.c").write_text(src)
    subprocess.run(["gcc", "-I", str(ROOT / "include"), str(tmp_path / "probe.c"), "-o", str(tmp_path / "probe")], check=True)
    for line in subprocess.run([str(tmp_path / "probe")], check=True, capture_output=True, text=True).stdout.splitlines():
        n, size, off = line.split()
        t = getattr(abi, n)
        assert C.sizeof(t) == int(size), n
        assert getattr(t, last[n]).offset == int(off), n


def test_no_device_is_a_loud_error_not_a_fallback(lib):
    """On a machine without a GPU the render entry points must fail with SHM_ERR_NO_DEVICE."""
    if lib.shm_device_count() > 0:
        pytest.skip("a GPU is present")
    sc = scenes.cornell_box(lib, 16, 16)
    h = C.c_void_p()
    rc = lib.shm_scene_create(C.byref(sc.desc), 0, C.byref(h))
    assert rc == -4 and not h.value and b"no CPU fallback" in lib.shm_last_error()


def py_tiles(pb, tw=8, th=8):
    """Tile::tile restated literally (tile.rs:21-104)."""
    w, h = pb[2] - pb[0], pb[3] - pb[1]
    nh, rh, nv, rv = w // tw, w % tw, h // th, h % th
    out = []
    for ty in range(nv):
        for tx in range(nh):
            out.append((pb[0] + tx * tw, pb[1] + ty * th, pb[0] + tx * tw + tw, pb[1] + ty * th + th))
        if rh > 0:
            out.append((pb[0] + nh * tw, pb[1] + ty * th, pb[0] + nh * tw + rh, pb[1] + ty * th + th))
    if rv > 0:
        for tx in range(nh):
            out.append((pb[0] + tx * tw, pb[1] + nv * th, pb[0] + tx * tw + tw, pb[1] + nv * th + rv))
    if rh > 0 and rv > 0:
        out.append((pb[0] + nh * tw, pb[1] + nv * th, pb[0] + nh * tw + rh, pb[1] + nv * th + rv))
    return out


@pytest.mark.parametrize("pb", [(0, 0, 128, 128), (0, 0, 13, 7), (3, 5, 36, 22), (0, 0, 8, 8), (0, 0, 3, 3), (10, 10, 1034, 1034)])
def test_tile_bounds(lib, pb):
    tiles, n = scn.tiles_for(lib, pb)
    got = [(t.x0, t.y0, t.x1, t.y1) for t in tiles[:n]]
    assert got == py_tiles(pb)
    cover = np.zeros((pb[3] - pb[1], pb[2] - pb[0]), np.int32)
    for x0, y0, x1, y1 in got:
        cover[y0 - pb[1]:y1 - pb[1], x0 - pb[0]:x1 - pb[0]] += 1
    assert (cover == 1).all()  # exclusive ownership: the film-write argument of integrator.rs:277-286


def test_wave_schedule():
    """integrator.rs:231-233, 306-308."""
    assert scn.wave_schedule(1) == [(0, 1)]
    assert scn.wave_schedule(4) == [(0, 1), (1, 2), (2, 4)]
    sizes = [e - s for s, e in scn.wave_schedule(256)]
    assert sizes == [1, 1, 2, 4, 8, 16, 32, 64, 64, 64] and sum(sizes) == 256
    assert sum(e - s for s, e in scn.wave_schedule(1024)) == 1024 and len(scn.wave_schedule(1024)) == 22


def check_bvh(desc, info):
    nodes = np.frombuffer(C.string_at(desc.nodes, desc.n_nodes * 32), dtype=np.dtype(
        [("bmin", "<f4", 3), ("bmax", "<f4", 3), ("offset", "<u4"), ("n_prims", "<u2"), ("axis", "u1"), ("pad", "u1")]))
    n = desc.n_primitives
    pb = info["bounds"][info["order"]]  # leaf-order primitive bounds
    seen = np.zeros(n, np.int32)
    stack = [0]
    visited = 0
    depth_max = 0
    depth = {0: 0}
    while stack:
        i = stack.pop()
        visited += 1
        nd = nodes[i]
        if nd["n_prims"] > 0:
            sl = slice(int(nd["offset"]), int(nd["offset"]) + int(nd["n_prims"]))
            seen[sl] += 1
            assert (pb[sl, :3] >= nd["bmin"]).all() and (pb[sl, 3:] <= nd["bmax"]).all()
            assert np.array_equal(pb[sl, :3].min(0), nd["bmin"]) and np.array_equal(pb[sl, 3:].max(0), nd["bmax"])
            depth_max = max(depth_max, depth[i])
        else:
            a, b = i + 1, int(nd["offset"])
            assert b > a and nd["axis"] <= 2
            for c in (a, b):
                assert (nodes[c]["bmin"] >= nd["bmin"]).all() and (nodes[c]["bmax"] <= nd["bmax"]).all()
                depth[c] = depth[i] + 1
            assert np.array_equal(np.minimum(nodes[a]["bmin"], nodes[b]["bmin"]), nd["bmin"])
            assert np.array_equal(np.maximum(nodes[a]["bmax"], nodes[b]["bmax"]), nd["bmax"])
            stack += [b, a]
    assert visited == desc.n_nodes and (seen == 1).all()
    assert sorted(info["order"].tolist()) == list(range(n))
    return nodes, depth_max


def test_bvh_build_structure(lib):
    """BvhAggregate::new (aggregate.rs:207-467): every primitive in exactly one leaf, DFS order (first child = i+1),
    parent bounds = union of children, leaves only for single prims or degenerate centroid bounds."""
    sc = scenes.ganesha_proxy(lib, 32, 32, n=12)
    nodes, depth = check_bvh(sc.desc, sc.info)
    assert depth < 64
    multi = nodes[nodes["n_prims"] > 1]
    # multi-primitive leaves only arise from coincident centroids (the room's quads): aggregate.rs:345
    assert len(multi) <= 16
    sc2 = scenes.cornell_box(lib, 16, 16)
    check_bvh(sc2.desc, sc2.info)
    assert sc2.desc.n_primitives == 32


def test_bvh_single_primitive_bounds(lib):
    """aggregate.rs:575-598 single_primitive_bvh: the BVH bounds equal the primitive's bounds."""
    sc = scenes.three_spheres(lib, offsets=(0.0,))
    assert sc.desc.n_nodes == 1
    nd = sc.desc.nodes[0]
    assert list(nd.bmin) == [-1.0, -1.0, -1.0] and list(nd.bmax) == [1.0, 1.0, 1.0] and nd.n_prims == 1


def test_bvh_equal_counts_and_errors(lib):
    rng = np.random.default_rng(3)
    lo = rng.random((257, 3)).astype(np.float32)
    bounds = np.ascontiguousarray(np.concatenate([lo, lo + 0.01], axis=1).astype(np.float32))
    nodes = (abi.ShmBvhNode * (2 * 257))()
    order = np.zeros(257, np.uint32)
    nn = C.c_uint32()
    for method in (0, 1):
        assert lib.shm_bvh_build(bounds.ctypes.data_as(abi.c_float_p), 257, method, nodes, C.byref(nn), order.ctypes.data_as(abi.c_u32_p)) == 0
        assert nn.value == 2 * 257 - 1 and sorted(order.tolist()) == list(range(257))
    assert lib.shm_bvh_build(bounds.ctypes.data_as(abi.c_float_p), 0, 0, nodes, C.byref(nn), order.ctypes.data_as(abi.c_u32_p)) == -1
    assert lib.shm_bvh_build(bounds.ctypes.data_as(abi.c_float_p), 257, 7, nodes, C.byref(nn), order.ctypes.data_as(abi.c_u32_p)) == -1
    bad = bounds.copy()
    bad[5, 1] = np.nan  # the reference panics "Unexpected NaN" (aggregate.rs:373)
    assert lib.shm_bvh_build(bad.ctypes.data_as(abi.c_float_p), 257, 0, nodes, C.byref(nn), order.ctypes.data_as(abi.c_u32_p)) == -1


@pytest.mark.parametrize("method", [0, 1])
def test_bvh_build_on_host_threads_equals_the_serial_build(lib, method, monkeypatch):
    """aggregate.rs:389-394 leaves build_recursive serial ("This can be done in parallel ..."); shm_bvh_build runs the same recursion over disjoint sub-ranges on host
    threads from 65 536 primitives on (host_mirror.cpp, BuildTask). Nodes and primitive order must be the serial build's byte for byte — with coincident centroids
    (leaves of several primitives: the order inside them is the partition's), flat boxes and signed zeros in the bounds (the piecewise folds keep a zero's sign)."""
    rng = np.random.default_rng(11)
    n = 150_000
    lo = (rng.random((n, 3)) * 2.0 - 1.0).astype(np.float32)
    ext = (rng.random((n, 3)) * 0.01).astype(np.float32)
    lo[:3000] = lo[3000:6000]  # coincident boxes: the degenerate-centroid-bounds exit (aggregate.rs:345)
    ext[:3000] = ext[3000:6000]
    ext[6000:9000, 1] = 0.0  # flat boxes
    lo[9000:9400, 0] = 0.0   # +0 / -0 among the minima of one axis
    lo[9400:9800, 0] = -0.0
    bounds = np.ascontiguousarray(np.concatenate([lo, lo + ext], axis=1).astype(np.float32))

    def build():
        nodes = (abi.ShmBvhNode * (2 * n))()
        order = np.zeros(n, np.uint32)
        nn = C.c_uint32()
        assert lib.shm_bvh_build(bounds.ctypes.data_as(abi.c_float_p), n, method, nodes, C.byref(nn), order.ctypes.data_as(abi.c_u32_p)) == 0
        return nn.value, bytes(memoryview(nodes))[:nn.value * C.sizeof(abi.ShmBvhNode)], order.tobytes()

    monkeypatch.setenv("SHM_BVH_THREADS", "1")
    serial = build()
    for threads in ("2", "5", "16"):
        monkeypatch.setenv("SHM_BVH_THREADS", threads)
        got = build()
        assert got[0] == serial[0] and got[2] == serial[2], threads
        assert got[1] == serial[1], threads
    assert sorted(np.frombuffer(serial[2], np.uint32).tolist()) == list(range(n))


def test_camera_perspective(lib):
    """PerspectiveCamera::new: the centre of the raster maps to the optical axis, corners to +-tan(fov/2) on the short
    side (camera.rs:848-963), camera-world rendering space (camera.rs:507-523)."""
    b = scn.SceneBuilder()
    b.set_film(200, 100)
    rfw = b.set_camera_look_at(lib, (1, 2, 3), (1, 2, 2), (0, 1, 0), 60.0)
    cam = b.camera
    m = np.array(list(cam.camera_from_raster), np.float64).reshape(4, 4)

    def xf(p):
        q = m @ np.array([*p, 1.0])
        return q[:3] / q[3]

    c = xf((100, 50, 0))
    assert abs(c[0]) < 1e-6 and abs(c[1]) < 1e-6
    t = np.tan(np.deg2rad(30.0))
    top = xf((100, 0, 0))
    left = xf((0, 50, 0))
    assert abs(top[1] / top[2] - t) < 1e-5 and abs(left[0] / left[2] + 2 * t) < 1e-5
    assert np.allclose(rfw[:3, 3], [-1, -2, -3]) and np.allclose(rfw[:3, :3], np.eye(3))
    rfc = np.array(list(cam.render_from_camera), np.float64).reshape(4, 4)
    assert np.allclose(rfc[:3, 3], 0, atol=1e-6)  # camera at the render-space origin
    assert np.allclose(rfc[:3, 2], [0, 0, -1], atol=1e-6)  # looking down -z in world = +z in camera space
    dx = np.array(list(cam.dx_camera))
    assert dx[0] > 0 and abs(dx[1]) < 1e-9


def test_write_pfm_roundtrip(lib, tmp_path):
    """Image::write_pfm (image.rs:1333-1377): 'PF', dims, scale -1 (little endian), rows bottom-up."""
    img = np.arange(5 * 3 * 3, dtype=np.float32).reshape(3, 5, 3)
    path = tmp_path / "x.pfm"
    assert lib.shm_write_pfm(str(path).encode(), img.ctypes.data_as(abi.c_float_p), 5, 3) == 0
    raw = path.read_bytes()
    header, body = raw[:raw.index(b"-1.0\n") + 5], raw[raw.index(b"-1.0\n") + 5:]
    assert header == b"PF\n5 3\n-1.0\n" and len(body) == 5 * 3 * 3 * 4
    back = np.frombuffer(body, "<f4").reshape(3, 5, 3)[::-1]
    assert np.array_equal(back, img)


def test_scene_validation_errors(orc, lib):
    """flatten_scene rejects malformed descriptions with codes instead of panicking (the reference panics)."""
    import oracle_py
    sc = scenes.cornell_box(lib, 16, 16)
    d = sc.desc
    h = C.c_void_p()
    for field, bad in (("abi_version", 99), ("n_nodes", 0), ("n_primitives", 0), ("n_materials", 0)):
        keep = getattr(d, field)
        setattr(d, field, bad)
        assert orc.orc_scene_create(C.byref(d), C.byref(h)) == -1
        setattr(d, field, keep)
    keep = d.materials[0].kind
    d.materials[0].kind = 9  # e.g. a coated material: SURVEY §8f row, not in the contract
    assert orc.orc_scene_create(C.byref(d), C.byref(h)) == -2
    d.materials[0].kind = keep
    keep = d.primitives[0].material
    d.primitives[0].material = 0xffffffff  # the reference's `material: None` (medium interface, skip_intersection): named and rejected
    assert orc.orc_scene_create(C.byref(d), C.byref(h)) == -2
    d.primitives[0].material = keep
    keep = d.nodes[0].offset
    d.nodes[0].offset = 10 ** 6
    assert orc.orc_scene_create(C.byref(d), C.byref(h)) == -1
    d.nodes[0].offset = keep
    o = oracle_py.Oracle(d)  # intact again
    o.close()


def test_film_get_image(lib, tmp_path):
    """RgbFilm::get_pixel_rgb / get_image (film.rs:647-707, 720-738) and write_pfm: sum / weight in f32, zero-weight pixels
    untouched, the 3x3 output matrix applied as mul_mat_vec does (accumulating from 0 in column order), the f16 clamp as
    the reference writes it, and the PFM file read back row-flipped."""
    rng = np.random.default_rng(2)
    film = np.zeros((5, 7), dtype=render.FILM_DTYPE)
    film["rgb_sum"] = rng.uniform(0, 50, (5, 7, 3))
    film["weight_sum"] = rng.integers(1, 9, (5, 7)).astype(np.float64)
    film["weight_sum"][0, 0] = 0.0
    film["rgb_sum"][1, 1] = (1e6, 3e6, 2.0)   # r and g above the f16 range
    film["rgb_sum"][1, 2] = (1.0, 3e6, 2.0)   # only g above it
    film["weight_sum"][1, 1] = film["weight_sum"][1, 2] = 1.0
    m = render.SRGB_FROM_XYZ
    img = render.film_get_image(lib, film, m)
    rgb = film["rgb_sum"].astype(np.float32)
    w = film["weight_sum"].astype(np.float32)
    nz = w != 0
    rgb[nz] = rgb[nz] / w[nz][:, None]
    want = np.zeros_like(rgb)
    for r in range(3):
        acc = np.zeros(rgb.shape[:2], np.float32)
        for c in range(3):
            acc = (acc + m[r, c] * rgb[..., c]).astype(np.float32)
        want[..., r] = acc
    assert np.array_equal(img.view(np.uint32), want.view(np.uint32))
    ident = render.film_get_image(lib, film)
    assert np.array_equal(ident, render.film_to_rgb(film))
    half = render.film_get_image(lib, film, None, write_fp16=True)
    assert half[1, 1, 0] == 65504.0 and half[1, 1, 1] == 3e6  # (sic) g > max clamps r, g itself is left alone
    assert half[1, 2, 0] == 65504.0 and half[1, 2, 1] == 3e6 and half[1, 2, 2] == 2.0
    path = tmp_path / "out.pfm"
    abi.check(lib, lib.shm_write_pfm(str(path).encode(), img.ctypes.data_as(abi.c_float_p), 7, 5), "shm_write_pfm")
    raw = path.read_bytes()
    header, body = raw[:raw.index(b"-1.0\n") + 5], raw[raw.index(b"-1.0\n") + 5:]
    assert header == b"PF\n7 5\n-1.0\n"
    back = np.frombuffer(body, "<f4").reshape(5, 7, 3)[::-1]
    assert np.array_equal(back, img)


def test_integrator_mirror_errors(lib):
    """create_integrator (integrator.rs:16-42) through the C++ host mirror: an unknown name is the reference's
    "Unknown integrator" panic turned into an error code + message; without a GPU the three integrators the reference knows
    fail with the no-device error instead of falling back."""
    sc = scenes.cornell_box(lib, 16, 16)
    film = np.zeros((16, 16), dtype=render.FILM_DTYPE)
    args = (C.byref(sc.desc), 0, 5, 0, 1, 1, 2, 0, 0, 0, film.ctypes.data_as(C.c_void_p), None, None)
    assert lib.shm_integrator_render(b"bdpt", *args) == -2 and b"Unknown integrator bdpt" in lib.shm_last_error()
    if lib.shm_device_count() == 0:
        for name in (b"path", b"simplepath", b"randomwalk"):
            assert lib.shm_integrator_render(name, *args) == -2 and b"no CPU fallback" in lib.shm_last_error()


def test_orthographic_camera(lib):
    """OrthographicCamera (camera.rs:658-840) through the host mirror + the oracle's camera-ray generator: every ray runs along
    the view direction, origins span the screen window ([-aspect, aspect] x [-1, 1] around the camera position, y down in
    raster space), one pixel apart by dx_camera / dy_camera.
    The integrators call generate_ray_differential (integrator.rs:351), and OrthographicCamera's (camera.rs:769-792) returns its
    ray in CAMERA space: it never applies render_from_camera, unlike generate_ray (:747-767). That is what the reference renders,
    so it is what is restated: correct for a camera whose axes are the world's, and the second half of this test pins the
    behaviour for a rotated one."""
    import oracle_py
    from oracle_py import fa  # noqa: F401

    def rays_of(pos, look_at):
        b = scn.SceneBuilder()
        b.set_film(64, 32)
        rfw = b.set_camera_look_at(lib, pos, look_at, (0, 1, 0), 30.0, orthographic=True)
        assert b.camera.kind == abi.SHM_CAMERA_ORTHOGRAPHIC
        m = b.material_diffuse(0.5)
        p = np.array([(-9, -9, 1), (9, -9, 1), (0, 9, 1)], np.float32)
        pr = (np.c_[p, np.ones(3, np.float32)] @ rfw.T)[:, :3]
        b.add_mesh(pr, [[0, 1, 2]], m)
        desc, _ = b.build(lib)
        o = oracle_py.Oracle(desc)
        rays = {}
        for (px, py) in ((0, 0), (63, 0), (0, 31), (32, 16), (33, 16), (32, 17)):
            out = (C.c_float * 14)()
            o.lib.orc_fn_camera_ray(o.handle, px, py, 0, 0, out)
            rays[(px, py)] = np.array(out[:6], np.float64)
        o.close()
        return b, rays

    b, rays = rays_of((1.0, 2.0, -5.0), (1.0, 2.0, 0.0))  # looking down +z with y up: camera axes == world axes
    for r in rays.values():
        assert np.allclose(r[3:], (0, 0, 1), atol=1e-6)
    # render space = world translated by -camera position: the film centre is the origin, x spans [-2, 2], y [-1, 1]
    assert abs(rays[(0, 0)][0] + 2.0) < 0.08 and abs(rays[(63, 0)][0] - 2.0) < 0.08
    assert rays[(0, 0)][1] > 0.9 and rays[(0, 31)][1] < -0.9  # raster y grows downwards
    dx = rays[(33, 16)][:3] - rays[(32, 16)][:3]
    dy = rays[(32, 17)][:3] - rays[(32, 16)][:3]
    assert np.allclose(np.abs(dx), np.abs(np.array(list(b.camera.dx_camera))), atol=0.08)  # jittered samples: one pixel +- the jitter
    assert abs(abs(dy[1]) - abs(b.camera.dy_camera[1])) < 0.08
    assert np.allclose(b.camera.min_pos_differential_x[:], b.camera.dx_camera[:]) and np.allclose(b.camera.min_dir_differential_y[:], 0)  # camera.rs:733-736
    # a camera turned to look down -z: the rays are the SAME camera-space rays (still +z), not rotated into render space
    _, turned = rays_of((1.0, 2.0, 5.0), (1.0, 2.0, 0.0))
    for k in rays:
        assert np.array_equal(turned[k], rays[k])
