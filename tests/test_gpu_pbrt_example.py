"""examples/render_pbrt.c — a C-only caller (no Python, no torch in the process) of the whole chain: shm_scene_load_pbrt ->
shm_render_multi -> shm_film_get_image -> shm_write_pfm. Built with gcc against the in-tree library, run on S1 written as .pbrt text; its
PFM must equal, bit for bit, the image of the same scene built by the Python generator and rendered through the ctypes binding."""
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

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
    want = render.film_get_image(gpu_lib, film, render.SRGB_FROM_XYZ)
    assert np.array_equal(img, want)
    # a scene file the loader rejects: the message with file:line reaches the caller, exit code 1
    bad = tmp_path / "bad.pbrt"
    bad.write_text('WorldBegin\nShape "sphere"\nShape "curve"\n')
    r = subprocess.run([str(exe), str(bad)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "bad.pbrt:3" in r.stderr
