#!/usr/bin/env python3
"""Generate an sRGB RGB -> sigmoid-polynomial coefficient table in the rgb2spec `.spec` layout (res, scale[res],
data[3][res][res][res][3]) with the published Jakob & Hanika (2019) optimisation: for every table cell a Gauss-Newton fit of
the three coefficients so that the reflectance s(c0 l^2 + c1 l + c2) under D65 reproduces the cell's RGB (CIELAB residual).

The reference loads rgbtospec/srgb.spec (resolution 64) at run time (rgb_to_spectra.rs:27-31); the blobs are not in its
repository. A Rust host hands its own table through ShmSceneDesc::color_space; this script only provides one for the tests,
the examples and bench scenes of THIS repository (resolution 16 keeps the committed file small). Output:
shimmer_amd/data/rgb2spec_srgb.npz {res, scale, data}.
"""
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
T = dict(np.load(ROOT / "shimmer_amd" / "data" / "spectral_tables.npz"))
RES = 16

lam = np.arange(360.0, 831.0)
xyz_bar = np.stack([T["CIE_X"], T["CIE_Y"], T["CIE_Z"]]).astype(np.float64)
d65_l, d65_v = T["CIE_ILLUM_D6500"][0::2].astype(np.float64), T["CIE_ILLUM_D6500"][1::2].astype(np.float64)
illum = np.interp(lam, d65_l, d65_v)
w = np.ones_like(lam)
w[0] = w[-1] = 0.5  # trapezoid, 1 nm
XYZ_TO_SRGB = np.array([[3.240479, -1.537150, -0.498535], [-0.969256, 1.875991, 0.041556], [0.055648, -0.204043, 1.057311]])
SRGB_TO_XYZ = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]])
norm = 1.0 / (xyz_bar[1] * illum * w).sum()
rgb_tbl = XYZ_TO_SRGB @ (xyz_bar * illum * w * norm)  # (3, n): RGB response per wavelength
white = (xyz_bar * illum * w * norm).sum(axis=1)
lam_n = (lam - 360.0) / (830.0 - 360.0)


def lab(xyz):
    r = xyz / white
    f = np.where(r > 216.0 / 24389.0, np.cbrt(r), (24389.0 / 27.0 * r + 16.0) / 116.0)
    return np.stack([116.0 * f[..., 1] - 16.0, 500.0 * (f[..., 0] - f[..., 1]), 200.0 * (f[..., 1] - f[..., 2])], axis=-1)


def residual(c, rgb):
    x = (c[..., 0:1] * lam_n + c[..., 1:2]) * lam_n + c[..., 2:3]
    s = 0.5 + 0.5 * x / np.sqrt(1.0 + x * x)
    out = s @ rgb_tbl.T
    return lab(rgb @ SRGB_TO_XYZ.T) - lab(out @ SRGB_TO_XYZ.T)


def gauss_newton(rgb, c, iters=15):
    for _ in range(iters):
        r = residual(c, rgb)
        J = np.empty(c.shape + (3,))
        for k in range(3):
            e = np.zeros(3)
            e[k] = 1e-4
            J[..., :, k] = (residual(c + e, rgb) - residual(c - e, rgb)) / 2e-4
        JtJ = np.einsum("...ki,...kj->...ij", J, J) + 1e-12 * np.eye(3)
        Jtr = np.einsum("...ki,...k->...i", J, r)
        step = np.linalg.solve(JtJ, Jtr[..., None])[..., 0]
        c = np.clip(c - step, -200.0, 200.0)
        if np.abs(r).max() < 1e-6:
            break
    return c


def smoothstep(x):
    return x * x * (3.0 - 2.0 * x)


def main():
    scale = smoothstep(smoothstep(np.arange(RES) / (RES - 1.0)))
    data = np.zeros((3, RES, RES, RES, 3), np.float32)
    xy = np.arange(RES) / (RES - 1.0)
    X, Y = np.meshgrid(xy, xy, indexing="xy")  # X varies along the last (x) axis
    c0, c1 = 360.0, 1.0 / (830.0 - 360.0)
    for l in range(3):
        start = RES // 5
        for sweep in (range(start, RES), range(start, -1, -1)):
            c = np.zeros((RES, RES, 3))
            for k in sweep:
                b = scale[k]
                rgb = np.empty((RES, RES, 3))
                rgb[..., l] = b
                rgb[..., (l + 1) % 3] = X * b
                rgb[..., (l + 2) % 3] = Y * b
                c = gauss_newton(rgb, c)
                A, B, Cc = c[..., 0], c[..., 1], c[..., 2]
                data[l, k, :, :, 0] = A * c1 * c1
                data[l, k, :, :, 1] = B * c1 - 2.0 * A * c0 * c1 * c1
                data[l, k, :, :, 2] = Cc - B * c0 * c1 + A * (c0 * c1) ** 2
    out = ROOT / "shimmer_amd" / "data" / "rgb2spec_srgb.npz"
    np.savez_compressed(out, res=np.uint32(RES), scale=scale.astype(np.float32), data=data)
    print("wrote", out, data.shape, "finite:", np.isfinite(data).all())


if __name__ == "__main__":
    main()
