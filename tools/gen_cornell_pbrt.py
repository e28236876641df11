#!/usr/bin/env python3
"""Writes examples/scenes/cornell_box.pbrt: S2 (BASELINE.json configs[1]; shimmer_amd/scenes.py cornell_box) as PBRT-v4 text, world space — 32 triangles:
five walls x 2, two boxes x 5 faces x 2, the ceiling emitter x 2. The file is what a maintainer hands to the Rust binary ("comparing whole
images with the reference binary"); this repo loads it through its own restatement of the reference's loader (host/pbrt_loader.cpp)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from shimmer_amd import scenes as S


def rot_y(p, deg, centre):
    a = np.deg2rad(deg)
    c, s = np.cos(a), np.sin(a)
    q = p - centre
    out = np.stack([c * q[:, 0] + s * q[:, 2], q[:, 1], -s * q[:, 0] + c * q[:, 2]], axis=1)
    return (out + centre).astype(np.float32)


def mesh(name, p, vi):
    pts = "  ".join(" ".join(repr(float(np.float32(c))) for c in v) for v in p)
    idx = "  ".join(" ".join(str(int(i)) for i in t) for t in vi)
    return f'  NamedMaterial "{name}"\n  Shape "trianglemesh" "point3 P" [ {pts} ]\n    "integer indices" [ {idx} ]\n'


def inward(q):
    p, vi = q
    return p, vi[:, ::-1].copy()


floor = inward(S._quad((-1, 0, -1), (1, 0, -1), (1, 0, 1), (-1, 0, 1)))
ceil_ = S._quad((-1, 2, -1), (1, 2, -1), (1, 2, 1), (-1, 2, 1))
back = S._quad((-1, 0, -1), (-1, 2, -1), (1, 2, -1), (1, 0, -1))
left = S._quad((-1, 0, -1), (-1, 0, 1), (-1, 2, 1), (-1, 2, -1))
right = inward(S._quad((1, 0, -1), (1, 0, 1), (1, 2, 1), (1, 2, -1)))
tall = S._box((-0.75, 0.0, -0.65), (-0.15, 1.2, -0.05), faces="xXYzZ")
tall = (rot_y(tall[0], 18.0, np.array([-0.45, 0, -0.35], np.float32)), tall[1])
short = S._box((0.1, 0.0, 0.0), (0.7, 0.6, 0.6), faces="xXYzZ")
short = (rot_y(short[0], -17.0, np.array([0.4, 0, 0.3], np.float32)), short[1])
light = S._quad((-0.3, 1.98, -0.3), (0.3, 1.98, -0.3), (0.3, 1.98, 0.3), (-0.3, 1.98, 0.3))
cb = S._merge([ceil_, back])

text = f'''# S2 (BASELINE configs[1]): the 32-triangle Cornell-style box — shimmer_amd/scenes.py cornell_box, written by tools/gen_cornell_pbrt.py
LookAt 0 1 3.4   0 1 0   0 1 0
Camera "perspective" "float fov" [ 39 ]
Film "rgb" "integer xresolution" [ 512 ] "integer yresolution" [ 512 ] "string filename" "s2.pfm"
Sampler "independent" "integer pixelsamples" 64
Integrator "path" "integer maxdepth" [ 5 ]
WorldBegin
MakeNamedMaterial "white" "string type" "diffuse" "float reflectance" 0.75
MakeNamedMaterial "red" "string type" "diffuse" "spectrum reflectance" [ 359 0.05 831 0.75 ]
MakeNamedMaterial "green" "string type" "diffuse" "spectrum reflectance" [ 359 0.6 831 0.08 ]
MakeNamedMaterial "black" "string type" "diffuse" "float reflectance" 0
AttributeBegin
{mesh("white", *cb)}{mesh("white", *floor)}{mesh("red", *left)}{mesh("green", *right)}{mesh("white", *tall)}{mesh("white", *short)}AttributeEnd
AttributeBegin
  AreaLightSource "diffuse" "blackbody L" [ 6500 ] "float scale" 20
{mesh("black", *light)}AttributeEnd
'''
out = os.path.join(ROOT, "examples", "scenes", "cornell_box.pbrt")
open(out, "w").write(text)
print("wrote", out, len(text), "bytes")
