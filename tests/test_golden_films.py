"""Regression pins for whole renders: sha256 of the oracle's f64 film for a set of small scenes that together reach every
feature of the path (tests/golden/films.json). The GPU parity tests compare the HIP path with the oracle bit for bit, so these
hashes pin BOTH against unintended arithmetic changes in the shared headers from one round to the next. They are not reference
values (the reference cannot be run here and its sample stream is not reproducible): regenerate deliberately with
`python tests/test_golden_films.py --regen` when a restatement is corrected, and say so in the commit."""
import hashlib
import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
FILMS = ROOT / "tests" / "golden" / "films.json"


def cases(scenes, lib):
    env = scenes.environment_image(16)
    return {
        "sphere_light_path": (lambda: scenes.sphere_light(lib, 32, 32), dict(spp=4, max_depth=5, seed=0)),
        "cornell_path": (lambda: scenes.cornell_box(lib, 32, 32), dict(spp=4, max_depth=5, seed=1)),
        "cornell_coated_path": (lambda: scenes.cornell_box(lib, 24, 24, coated=True), dict(spp=2, max_depth=5, seed=2)),
        "cornell_mix_regularize": (lambda: scenes.cornell_box(lib, 24, 24, mix=True), dict(spp=2, max_depth=5, seed=3, regularize=True)),
        "cornell_patches_path": (lambda: scenes.cornell_box(lib, 24, 24, patches=True), dict(spp=2, max_depth=5, seed=4)),
        "cornell_textured_path": (lambda: scenes.cornell_box(lib, 32, 32, textured=True), dict(spp=4, max_depth=6, seed=5)),
        "cornell_textured_simplepath": (lambda: scenes.cornell_box(lib, 24, 24, textured=True), dict(spp=2, max_depth=4, seed=6, integrator="simplepath")),
        "cornell_textured_randomwalk": (lambda: scenes.cornell_box(lib, 24, 24, textured=True), dict(spp=2, max_depth=4, seed=7, integrator="randomwalk")),
        "crown_proxy_depth16": (lambda: scenes.crown_proxy(lib, 20, 28, level=1, n_glass=6, n_gold=2), dict(spp=2, max_depth=16, seed=8)),
        "environment_path": (lambda: scenes.three_spheres(lib, 32, 24, camera=(0.75, 0.5, 9.0), environment=env), dict(spp=4, max_depth=4, seed=9)),
        "environment_simplepath": (lambda: scenes.three_spheres(lib, 24, 16, camera=(0.75, 0.5, 9.0), environment=env), dict(spp=2, max_depth=4, seed=10, integrator="simplepath")),
        "instanced_path": (lambda: scenes.instanced_scene(lib, 32, 24), dict(spp=4, max_depth=4, seed=14)),
        "random_scene_3": (lambda: scenes.random_scene(lib, 3), dict(spp=2, max_depth=6, seed=11)),
        "random_scene_2_ortho_nojitter": (lambda: scenes.random_scene(lib, 2), dict(spp=2, max_depth=5, seed=12, disable_pixel_jitter=True, disable_wavelength_jitter=True)),
        "random_scene_12_textured": (lambda: scenes.random_scene(lib, 12), dict(spp=2, max_depth=6, seed=13)),
    }


def film_hash(lib, make_scene, kw):
    import oracle_py
    from shimmer_amd import render
    sc = make_scene()
    o = oracle_py.Oracle(sc.desc)
    try:
        film, stats = o.render(render.make_params(**kw), n_threads=8)
    finally:
        o.close()
    h = hashlib.sha256(np.ascontiguousarray(film).tobytes()).hexdigest()
    return {"sha256": h, "rays_closest": int(stats["rays_closest"]), "rays_any": int(stats["rays_any"]),
            "nodes_closest": int(stats["nodes_closest"]), "mean_rgb_sum": float(film["rgb_sum"].mean())}


def test_golden_films(lib):
    from shimmer_amd import scenes
    want = json.loads(FILMS.read_text())
    got_cases = cases(scenes, lib)
    assert set(want) == set(got_cases)
    for name, (make_scene, kw) in got_cases.items():
        got = film_hash(lib, make_scene, kw)
        assert got["rays_closest"] == want[name]["rays_closest"] and got["rays_any"] == want[name]["rays_any"], name
        assert got["sha256"] == want[name]["sha256"], (name, got["mean_rgb_sum"], want[name]["mean_rgb_sum"])


if __name__ == "__main__":
    assert "--regen" in sys.argv, "usage: python tests/test_golden_films.py --regen"
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "oracle"))
    from shimmer_amd import abi, scenes
    lib = abi.load_library()
    out = {name: film_hash(lib, mk, kw) for name, (mk, kw) in cases(scenes, lib).items()}
    FILMS.write_text(json.dumps(out, indent=1) + "\n")
    print("wrote", FILMS, len(out), "films")
