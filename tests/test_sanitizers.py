"""AddressSanitizer + UndefinedBehaviorSanitizer over the CPU side (the reference has no sanitizer runs; GPU ASan is not
available on this pool): the oracle, the shared arithmetic headers, the scene marshalling (`host/flatten.h`, shared by the product
and the oracle) the product's host mirror (`host/host_mirror.cpp`: BVH build, tiling, cameras, PLY reader, PFM) and its PBRT-v4 scene loader
(`host/pbrt_loader.cpp`) are rebuilt
with `-fsanitize=address,undefined -fno-sanitize-recover=undefined` (`make -C oracle asan`) and the CPU test-suite is run against
those builds in a child process. Any report aborts the child."""
import os
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def test_cpu_suite_under_asan_and_ubsan():
    if os.environ.get("SHM_HOST_ONLY") == "1":
        pytest.skip("already inside the sanitizer run")
    if not shutil.which("g++") or not shutil.which("make"):
        pytest.skip("no host toolchain")
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not libasan or not Path(libasan).exists():
        pytest.skip("libasan not installed")
    # libstdc++ is preloaded with the sanitizer runtime: the interpreter does not link it, and ASan's __cxa_throw interceptor needs the
    # real symbol at start-up (the PBRT loader and the integrator mirror report errors with C++ exceptions caught inside the library)
    libstdcxx = subprocess.run(["gcc", "-print-file-name=libstdc++.so.6"], capture_output=True, text=True).stdout.strip()
    subprocess.run(["make", "-C", str(ROOT / "oracle"), "asan"], check=True, capture_output=True)
    env = dict(os.environ, LD_PRELOAD=libasan + (" " + libstdcxx if libstdcxx and Path(libstdcxx).exists() else ""), ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1",
               ORACLE_LIB=str(ROOT / "oracle" / "_build" / "liboracle_asan.so"), SHM_LIB=str(ROOT / "oracle" / "_build" / "libhostmirror_asan.so"),
               SHM_HOST_ONLY="1",  # (the loader's default place for the tables is beside the PRODUCT library: name them for this build)
               SHM_RGB2SPEC_SRGB=str(ROOT / "shimmer_amd" / "data" / "rgb2spec_srgb_res64.spec"),
               SHM_RGB2SPEC_REC2020=str(ROOT / "shimmer_amd" / "data" / "rgb2spec_rec2020_res64.spec"),
               SHM_RGB2SPEC_ACES2065_1=str(ROOT / "shimmer_amd" / "data" / "rgb2spec_aces2065_1_res64.spec"))
    # everything that runs without a device entry point (those live in shimmer_hip.hip, which only hipcc builds)
    files = ["test_oracle_golden.py", "test_textures.py", "test_image_light.py", "test_instancing.py", "test_ply.py", "test_layered.py",
             "test_bilinear_patch.py", "test_spectra.py", "test_color_spaces.py", "test_reference_loader_vectors.py", "test_bvh_independent.py", "test_quirks_switch.py", "test_fuzz_scenes.py", "test_golden_films.py", "test_oracle_render.py", "test_host_mirror.py", "test_pbrt_loader.py", "test_leaf_golden.py", "test_image_io.py"]
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "not gpu", "-p", "no:cacheprovider",
           "-k", "not exports_every_declared and not no_device and not integrator_mirror_errors"] + [str(ROOT / "tests" / f) for f in files]
    r = subprocess.run(cmd, cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=1500)
    out = r.stdout + r.stderr
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert r.returncode == 0, out[-4000:]
    assert " passed" in out
