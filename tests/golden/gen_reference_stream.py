#!/usr/bin/env python3
"""Writes tests/golden/reference_stream.json: the oracle's "reference stream" renders (oracle.cpp, orc_render_reference_stream — ImageTileIntegrator::render on the
reference's own Xoshiro256++ sampler stream, as `RAYON_NUM_THREADS=1 shimmer <scene>.pbrt --seed 0 --spp N` draws it) of the two example scenes a maintainer with
cargo can hand to the Rust binary: examples/scenes/sphere_light.pbrt (S1) and examples/scenes/cornell_box.pbrt (S2). Per (scene, spp): the sha256 of the f64 film
sums, the sha256 of the PFM the repo's own get_image + PFM writer make of it (what the binary writes: film.rs:647-738, image.rs:1333-1377), ray counts, and the number
of 64-bit draws the render took from the stream. UNVERIFIED against the binary (no Rust toolchain in this image): INTEGRATION.md, "Comparing whole images with the
reference binary". `--full` adds the scenes' own sample counts (64 spp: about two minutes single-threaded)."""
import ctypes as C, hashlib, json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from shimmer_amd import abi, render
import oracle_py


def reference_stream_record(lib, scene_file, spp, seed=0):
    out = C.POINTER(abi.ShmPbrtScene)()
    abi.check(lib, lib.shm_scene_load_pbrt(os.path.join(ROOT, "examples", "scenes", scene_file).encode(), C.byref(out)), scene_file)
    try:
        s = out.contents
        p = s.params
        p.samples_per_pixel, p.seed = spp, seed
        o = oracle_py.Oracle(s.desc)
        film, st, draws = o.render_reference_stream(p)
        o.close()
        m = np.array(list(s.output_rgb_from_sensor_rgb), np.float32).reshape(3, 3)
        img = render.film_get_image(lib, film, m)
        with tempfile.TemporaryDirectory() as d:
            f = os.path.join(d, "out.pfm")
            abi.check(lib, lib.shm_write_pfm(f.encode(), img.ctypes.data_as(abi.c_float_p), img.shape[1], img.shape[0]), "shm_write_pfm")
            pfm = open(f, "rb").read()
        return {"scene": scene_file, "spp": spp, "seed": seed, "width": int(img.shape[1]), "height": int(img.shape[0]),
                "film_sha256": hashlib.sha256(film.tobytes()).hexdigest(), "pfm_sha256": hashlib.sha256(pfm).hexdigest(),
                "paths": st["paths"], "rays_closest": st["rays_closest"], "rays_any": st["rays_any"], "u64_draws": draws,
                "mean_rgb": [float(x) for x in img.reshape(-1, 3).mean(0)]}
    finally:
        lib.shm_pbrt_free(out)


if __name__ == "__main__":
    lib = abi.load_library()
    jobs = [("sphere_light.pbrt", 1), ("sphere_light.pbrt", 4), ("cornell_box.pbrt", 1), ("cornell_box.pbrt", 4)]
    if "--full" in sys.argv:
        jobs += [("sphere_light.pbrt", 64), ("cornell_box.pbrt", 64)]
    path = os.path.join(ROOT, "tests", "golden", "reference_stream.json")
    old = {(r["scene"], r["spp"]): r for r in json.load(open(path))["renders"]} if os.path.exists(path) else {}
    for scene, spp in jobs:
        old[(scene, spp)] = reference_stream_record(lib, scene, spp)
        print(old[(scene, spp)], flush=True)
    doc = {"what": "oracle reference-stream renders (Xoshiro256++ seeded by SplitMix64 from --seed, one sequential stream over waves -> tiles -> x -> y -> sample); "
                   "UNVERIFIED against the Rust binary; see tests/golden/gen_reference_stream.py and INTEGRATION.md",
           "renders": [old[k] for k in sorted(old)]}
    json.dump(doc, open(path, "w"), indent=1)
    print("wrote", path)
