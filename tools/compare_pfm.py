#!/usr/bin/env python3
"""Compares two PFM images (image.rs:1333-1377 / shm_write_pfm: "PF", width height, negative scale = little-endian, rows bottom-up):
sha256 of each file, per-pixel L-infinity over RGB, the number of pixels that differ.    python tools/compare_pfm.py a.pfm b.pfm"""
import hashlib, sys
import numpy as np


def read_pfm(path):
    raw = open(path, "rb").read()
    parts, pos = [], 0
    while len(parts) < 4:  # "PF", width, height, scale: whitespace-separated header tokens, one whitespace byte before the data
        while raw[pos:pos + 1].isspace():
            pos += 1
        end = pos
        while not raw[end:end + 1].isspace():
            end += 1
        parts.append(raw[pos:end].decode())
        pos = end
    pos += 1
    assert parts[0] == "PF", "a three-channel PFM is expected"
    w, h, scale = int(parts[1]), int(parts[2]), float(parts[3])
    img = np.frombuffer(raw, "<f4" if scale < 0 else ">f4", count=w * h * 3, offset=pos).reshape(h, w, 3)[::-1]
    return img.astype(np.float32), hashlib.sha256(raw).hexdigest()


if __name__ == "__main__":
    (a, ha), (b, hb) = read_pfm(sys.argv[1]), read_pfm(sys.argv[2])
    print(f"{sys.argv[1]}: {a.shape[1]}x{a.shape[0]} sha256 {ha}")
    print(f"{sys.argv[2]}: {b.shape[1]}x{b.shape[0]} sha256 {hb}")
    if a.shape != b.shape:
        sys.exit("different sizes")
    d = np.abs(a.astype(np.float64) - b.astype(np.float64)).max(axis=2)
    print(f"identical files: {ha == hb} | L_inf {d.max():.6e} | pixels that differ {int((d > 0).sum())} of {d.size} | north-star tolerance 1e-4: {'met' if d.max() < 1e-4 else 'NOT met'}")
