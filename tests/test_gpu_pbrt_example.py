"""examples/render_pbrt.c — a C-only caller (no Python, no torch in the process) of the whole chain: shm_scene_load_pbrt ->
shm_render_multi -> shm_film_get_image -> shm_write_pfm. Built with gcc against the in-tree library, run on S1 written as .pbrt text; its
PFM must equal, bit for bit, the image of the same scene built by the Python generator and rendered through the ctypes binding."""
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

import ctypes as C

import oracle_py
from shimmer_amd import abi, render, scenes
from test_pbrt_loader import S1_TEXT

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def test_c_example_renders_pbrt_file(gpu_lib, tmp_path):
    if not shutil.which("gcc"):
        pytest.skip("no host C compiler")
    exe = tmp_path / "render_pbrt"
    libdir = ROOT / "shimmer_amd" / "csrc"
    subprocess.run(["gcc", "-O2", "-I", str(ROOT / "include"), str(ROOT / "examples" / "render_pbrt.c"), "-L", str(libdir), "-lshimmer_hip",
                    f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], check=True)
    scene = tmp_path / "s1.pbrt"
    scene.write_text(S1_TEXT)
    out = tmp_path / "s1.pfm"
    r = subprocess.run([str(exe), str(scene), str(out), "--spp", "8", "--devices", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "48x40, 8 spp, integrator path, 1 device(s)" in r.stdout
    head, body = out.read_bytes().split(b"-1.0\n", 1)
    assert head == b"PF\n48 40\n"
    img = np.frombuffer(body, "<f4").reshape(40, 48, 3)[::-1]
    sc = scenes.sphere_light(gpu_lib, 48, 40)
    rr = render.Renderer(gpu_lib, sc.desc, device=0)
    film, _ = rr.render(render.make_params(seed=0, spp=8, max_depth=5))
    rr.close()
    # the example converts with the matrix the loader hands over (RgbFilm::new's output_rgb_from_sensor_rgb)
    loaded = C.POINTER(abi.ShmPbrtScene)()
    abi.check(gpu_lib, gpu_lib.shm_scene_parse_pbrt(S1_TEXT.encode(), None, C.byref(loaded)), "shm_scene_parse_pbrt")
    matrix = np.array(list(loaded.contents.output_rgb_from_sensor_rgb), np.float32).reshape(3, 3)
    gpu_lib.shm_pbrt_free(loaded)
    assert np.allclose(matrix, render.SRGB_FROM_XYZ, atol=2e-3)
    want = render.film_get_image(gpu_lib, film, matrix)
    assert np.array_equal(img, want)
    # a scene file the loader rejects: the message with file:line reaches the caller, exit code 1
    bad = tmp_path / "bad.pbrt"
    bad.write_text('WorldBegin\nShape "sphere"\nShape "curve"\n')
    r = subprocess.run([str(exe), str(bad)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "bad.pbrt:3" in r.stderr


TEXTURED_TEXT = """
Film "rgb" "integer xresolution" 40 "integer yresolution" 32
LookAt 0 1.2 -4  0 0.2 0  0 1 0
Camera "perspective" "float fov" 45
Sampler "independent" "integer pixelsamples" 8
Integrator "path" "integer maxdepth" 4
WorldBegin
AttributeBegin
  Rotate 40 0 1 0
  LightSource "infinite" "string filename" "env.png" "float scale" 1.5
AttributeEnd
Texture "wood" "spectrum" "imagemap" "string filename" "albedo.png" "string filter" "{filter}" "float uscale" 3 "float vscale" 3
Texture "bumps" "float" "imagemap" "string filename" "bump.png" "string encoding" "linear" "float scale" 0.03
Texture "blend" "spectrum" "mix" "texture tex1" "wood" "rgb tex2" [ 0.8 0.15 0.1 ] "float amount" 0.3
Material "diffuse" "texture reflectance" "blend" "texture displacement" "bumps"
Shape "trianglemesh" "point3 P" [ -3 -0.5 -3  3 -0.5 -3  3 -0.5 3  -3 -0.5 3 ] "integer indices" [ 0 2 1 0 3 2 ] "point2 uv" [ 0 0 1 0 1 1 0 1 ]
Material "conductor" "texture eta" "wood" "rgb k" [ 2 3 4 ] "float roughness" 0.15
Shape "sphere" "float radius" 0.6
Material "coateddiffuse" "rgb reflectance" [ 0.2 0.5 0.7 ] "string normalmap" "nmap.png" "float roughness" 0.05
Translate 1.4 0 0.3
Shape "sphere" "float radius" 0.5
AttributeBegin
  AreaLightSource "diffuse" "rgb L" [ 1 0.9 0.7 ] "float scale" 6
  Translate -2.4 1.5 0
  Shape "sphere" "float radius" 0.3
AttributeEnd
"""


@pytest.mark.parametrize("filter", ["bilinear", "ewa"])
def test_png_textured_pbrt_scene_renders_bit_exact(gpu_lib, tmp_path, filter):
    """A .pbrt scene with "rgb" spectra, PNG image textures (spectrum + float), a normal map and an environment-map light, loaded by the C++
    front end and rendered by the HIP path: the film equals the oracle's bit for bit (the staged textured kernels read exactly the level /
    texel / colour-space tables the loader built)."""
    from test_image_io import write_png
    rng = np.random.default_rng(21)
    write_png(tmp_path / "albedo.png", rng.integers(0, 256, size=(16, 16, 3)), 8, 2, filters=(1, 2))
    write_png(tmp_path / "bump.png", rng.integers(0, 256, size=(8, 8, 1)), 8, 0)
    write_png(tmp_path / "nmap.png", (127 + rng.integers(-20, 20, size=(4, 4, 3))).clip(0, 255), 8, 2)
    env = rng.integers(0, 80, size=(16, 16, 3))
    env[2:5, 9:12] = 255
    write_png(tmp_path / "env.png", env, 8, 2)
    out = C.POINTER(abi.ShmPbrtScene)()
    abi.check(gpu_lib, gpu_lib.shm_scene_parse_pbrt(TEXTURED_TEXT.format(filter=filter).encode(), str(tmp_path).encode(), C.byref(out)), "shm_scene_parse_pbrt")
    try:
        s = out.contents
        assert s.desc.n_image_textures == 4 and s.desc.n_image_lights == 1 and s.params.samples_per_pixel == 8
        rr = render.Renderer(gpu_lib, s.desc, device=0)
        film, st = rr.render(s.params)
        rr.close()
        o = oracle_py.Oracle(s.desc)
        want, ost = o.render(s.params, n_threads=8)
        o.close()
        assert np.isfinite(film["rgb_sum"]).all() and film["rgb_sum"].sum() > 0
        assert np.array_equal(film["rgb_sum"], want["rgb_sum"]) and np.array_equal(film["weight_sum"], want["weight_sum"])
        for k in ("paths", "rays_closest", "rays_any", "nodes_closest", "tris_closest", "nodes_any", "tris_any"):
            assert st[k] == ost[k], k
    finally:
        gpu_lib.shm_pbrt_free(out)


def test_example_scene_files_render_bit_exact(gpu_lib):
    """examples/scenes/*.pbrt through the loader and the HIP path at a few samples: the oracle's film, bit for bit."""
    for f in sorted((ROOT / "examples" / "scenes").glob("*.pbrt")):
        out = C.POINTER(abi.ShmPbrtScene)()
        abi.check(gpu_lib, gpu_lib.shm_scene_load_pbrt(str(f).encode(), C.byref(out)), f.name)
        try:
            s = out.contents
            s.params.samples_per_pixel = 3
            rr = render.Renderer(gpu_lib, s.desc, device=0)
            film, st = rr.render(s.params)
            rr.close()
            o = oracle_py.Oracle(s.desc)
            want, ost = o.render(s.params, n_threads=8)
            o.close()
            assert np.array_equal(film.view(np.uint64), want.view(np.uint64)), f.name
            for k in ("paths", "rays_closest", "rays_any", "nodes_closest", "tris_closest", "nodes_any", "tris_any"):
                assert st[k] == ost[k], (f.name, k)
        finally:
            gpu_lib.shm_pbrt_free(out)
