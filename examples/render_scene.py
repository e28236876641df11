#!/usr/bin/env python3
"""Render one of the repository's scenes on the GPU and write a PFM (and a tone-mapped PNG next to it).

    python examples/render_scene.py cornell --spp 256 --res 512 -o cornell.pfm
    python examples/render_scene.py textured | coated | patches | instanced | environment | ganesha | crown | fuzz:13

Everything goes through the C ABI of include/shimmer_hip.h (shimmer_amd/abi.py is the ctypes binding): scene description ->
shm_scene_create -> shm_render_device -> shm_film_read -> shm_film_get_image -> shm_write_pfm. Needs an MI355X: there is no CPU path.
"""
import argparse
import os
import struct
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shimmer_amd import abi, render, scenes  # noqa: E402


def make_scene(lib, name, w, h):
    if name == "cornell":
        return scenes.cornell_box(lib, w, h)
    if name == "textured":
        return scenes.cornell_box(lib, w, h, textured=True)
    if name == "coated":
        return scenes.cornell_box(lib, w, h, coated=True)
    if name == "patches":
        return scenes.cornell_box(lib, w, h, patches=True)
    if name == "instanced":
        return scenes.instanced_scene(lib, w, h)
    if name == "environment":
        return scenes.three_spheres(lib, w, h, camera=(0.75, 0.5, 9.0), environment=scenes.environment_image(64))
    if name == "ganesha":
        return scenes.ganesha_proxy(lib, w, h)
    if name == "crown":
        return scenes.crown_proxy(lib, w, h)
    if name.startswith("fuzz:"):
        return scenes.random_scene(lib, int(name.split(":")[1]), w, h)
    raise SystemExit(f"unknown scene {name}")


def write_png(path, rgb8):
    h, w, _ = rgb8.shape
    raw = b"".join(b"\x00" + rgb8[y].tobytes() for y in range(h))

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("scene")
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--max-depth", type=int, default=5)
    ap.add_argument("--integrator", default="path", choices=["path", "simplepath", "randomwalk"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--exposure", type=float, default=1.0)
    ap.add_argument("--quirks-off", action="store_true",
                    help="ShmRenderParams::disable_reference_quirks: PBRT-v4's forms of the reference's deviations (emitter sampling, instancing, ...: DESIGN.md section 2) "
                         "instead of the reference-exact default")
    ap.add_argument("-o", "--output", default="")
    args = ap.parse_args()
    lib = abi.load_library()
    if lib.shm_device_count() < 1:
        raise SystemExit("no HIP device visible (there is no CPU fallback)")
    w, h = args.res, args.height or args.res
    sc = make_scene(lib, args.scene, w, h)
    r = render.Renderer(lib, sc.desc, 0)
    p = render.make_params(seed=args.seed, spp=args.spp, max_depth=args.max_depth, integrator=args.integrator, reference_quirks=not args.quirks_off)
    r.clear()
    t0 = time.perf_counter()
    st = r.render_device(p)
    dt = time.perf_counter() - t0
    film = r.read_film()
    rays = st["rays_closest"] + st["rays_any"]
    print(f"{sc.name}: {w}x{h} x {args.spp} spp, {rays / 1e6:.1f} Mrays in {dt * 1e3:.1f} ms = {rays / dt / 1e6:.0f} Mray/s")
    img = render.film_get_image(lib, film, render.SRGB_FROM_XYZ)  # RgbFilm::get_image with an sRGB output matrix
    out = args.output or f"{args.scene.replace(':', '_')}.pfm"
    img = np.ascontiguousarray(img, np.float32)
    abi.check(lib, lib.shm_write_pfm(out.encode(), img.ctypes.data_as(abi.c_float_p), w, h), "shm_write_pfm")
    ldr = np.clip(img * args.exposure, 0.0, 1.0)
    ldr = np.where(ldr <= 0.0031308, 12.92 * ldr, 1.055 * np.power(ldr, 1 / 2.4) - 0.055)
    write_png(os.path.splitext(out)[0] + ".png", (ldr * 255 + 0.5).astype(np.uint8))
    print("wrote", out, "and", os.path.splitext(out)[0] + ".png")
    r.close()


if __name__ == "__main__":
    main()
