"""Image textures in the oracle = the shared headers (shm/texture.h): MIP-pyramid filtering, texture mappings,
the RGB -> sigmoid-coefficient lookup, ray differentials (camera, compute_differentials, approximate_dp_dxy, Igehy's specular
differentials) — each against an independent float64 re-evaluation of the cited formulas or a property of the construction.
The reference has no known answers for any of this (its mipmap / texture / camera modules carry no tests); the rgb2spec crate's
`fetch` is un-vendored (parity unpinned there, see texture.h)."""
import ctypes as C
import math

import numpy as np
import pytest

import oracle_py
from oracle_py import fa
from shimmer_amd import abi, render, scenes
from shimmer_amd.scene import generate_pyramid, tables

f32 = np.float32


@pytest.fixture(scope="module")
def env(lib):
    sc = scenes.cornell_box(lib, 48, 48, textured=True)
    o = oracle_py.Oracle(sc.desc)
    yield sc, o
    o.close()


def ulp_diff(a, b):
    a, b = np.float32(a), np.float32(b)
    ia, ib = np.int64(a.view(np.int32)), np.int64(b.view(np.int32))
    ia = ia if ia >= 0 else np.int64(-2**31) - ia
    ib = ib if ib >= 0 else np.int64(-2**31) - ib
    return abs(int(ia) - int(ib))


def test_log2(orc):
    for e in range(-40, 40):
        assert orc.orc_fn_log2(float(2.0 ** e)) == float(e)  # MIP level boundaries are exact
    rng = np.random.default_rng(3)
    for x in np.exp(rng.uniform(-20, 20, 400)).astype(np.float32):
        got, want = orc.orc_fn_log2(float(x)), math.log2(float(x))
        assert abs(got - want) <= 4e-7 * max(1.0, abs(want))
    assert orc.orc_fn_log2(0.0) == -np.inf and math.isnan(orc.orc_fn_log2(-1.0))


def test_generate_pyramid_is_the_box_filter():
    img = scenes.test_image(16, 3)
    lv = generate_pyramid(img)
    assert [l.shape[0] for l in lv] == [16, 8, 4, 2, 1]
    want = img.astype(np.float64).reshape(8, 2, 8, 2, 3).mean(axis=(1, 3))
    assert np.allclose(lv[1], want, atol=1e-6)
    assert np.allclose(lv[-1], img.astype(np.float64).mean(axis=(0, 1)), atol=1e-5)
    thin = generate_pyramid(np.arange(8, dtype=np.float32).reshape(1, 8) / 8)  # a 1-pixel-high image keeps halving in x only
    assert [l.shape for l in thin] == [(1, 8), (1, 4), (1, 2), (1, 1)]


# ---- an independent float64 MIPMap::filter -------------------------------------------------------------
def remap(x, y, w, h, wrap):
    if wrap == abi.SHM_WRAP_OCTAHEDRAL_SPHERE:
        if x < 0:
            x, y = -x, h - 1 - y
        elif x >= w:
            x, y = 2 * w - 1 - x, h - 1 - y
        if y < 0:
            x, y = w - 1 - x, -y
        elif y >= h:
            x, y = w - 1 - x, 2 * h - 1 - y
        x, y = (0 if w == 1 else x), (0 if h == 1 else y)
        return min(max(x, 0), w - 1), min(max(y, 0), h - 1)  # beyond one mirror the reference panics; the device clamps
    for _ in range(1):
        if not 0 <= x < w:
            if wrap == abi.SHM_WRAP_BLACK:
                return None
            x = min(max(x, 0), w - 1) if wrap == abi.SHM_WRAP_CLAMP else x % w
        if not 0 <= y < h:
            if wrap == abi.SHM_WRAP_BLACK:
                return None
            y = min(max(y, 0), h - 1) if wrap == abi.SHM_WRAP_CLAMP else y % h
    return x, y


def texel(level, x, y, wrap):
    h, w = level.shape[:2]
    r = remap(x, y, w, h, wrap)
    if r is None:
        return np.zeros(3)
    v = level[r[1], r[0]].astype(np.float64)
    return np.repeat(v, 3) if v.ndim == 0 else v


def bilerp(level, st, wrap):
    h, w = level.shape[:2]
    x, y = st[0] * w - 0.5, st[1] * h - 0.5
    xi, yi = math.floor(x), math.floor(y)
    dx, dy = x - xi, y - yi
    return ((1 - dx) * (1 - dy) * texel(level, xi, yi, wrap) + dx * (1 - dy) * texel(level, xi + 1, yi, wrap)
            + (1 - dx) * dy * texel(level, xi, yi + 1, wrap) + dx * dy * texel(level, xi + 1, yi + 1, wrap))


def ewa(levels, lut, ilevel, st, d0, d1, wrap):
    if ilevel >= len(levels):
        return texel(levels[-1], 0, 0, wrap)
    level = levels[ilevel]
    h, w = level.shape[:2]
    s, t = st[0] * w - 0.5, st[1] * h - 0.5
    d0, d1 = (d0[0] * w, d0[1] * h), (d1[0] * w, d1[1] * h)
    a = d0[1] ** 2 + d1[1] ** 2 + 1
    b = -2 * (d0[0] * d0[1] + d1[0] * d1[1])
    c = d0[0] ** 2 + d1[0] ** 2 + 1
    inv_f = 1 / (a * c - b * b * 0.25)
    a, b, c = a * inv_f, b * inv_f, c * inv_f
    det = -b * b + 4 * a * c
    us, vs = math.sqrt(max(0, det * c)), math.sqrt(max(0, a * det))
    s0, s1 = math.ceil(s - 2 / det * us), math.floor(s + 2 / det * us)
    t0, t1 = math.ceil(t - 2 / det * vs), math.floor(t + 2 / det * vs)
    tot, wsum = np.zeros(3), 0.0
    for it in range(t0, t1 + 1):
        for i_s in range(s0, s1 + 1):
            ss, tt = i_s - s, it - t
            r2 = a * ss * ss + b * ss * tt + c * tt * tt
            if r2 < 1:
                wgt = float(lut[min(int(r2 * 128), 127)])
                tot += wgt * texel(level, i_s, it, wrap)
                wsum += wgt
    return tot / wsum


def mip_filter(levels, lut, flt, wrap, max_aniso, st, d0, d1):
    n = len(levels)
    if flt == abi.SHM_TEXFILTER_EWA:
        if d0[0] ** 2 + d0[1] ** 2 < d1[0] ** 2 + d1[1] ** 2:
            d0, d1 = d1, d0
        longer, shorter = math.hypot(*d0), math.hypot(*d1)
        if shorter * max_aniso < longer and shorter > 0:
            sc = longer / (shorter * max_aniso)
            d1, shorter = (d1[0] * sc, d1[1] * sc), shorter * sc
        if shorter == 0:
            return bilerp(levels[0], st, wrap)
        lod = max(0.0, n - 1 + math.log2(shorter))
        il = math.floor(lod)
        return (1 - (lod - il)) * ewa(levels, lut, il, st, d0, d1, wrap) + (lod - il) * ewa(levels, lut, il + 1, st, d0, d1, wrap)
    width = 2 * max(abs(d0[0]), abs(d0[1]), abs(d1[0]), abs(d1[1]))
    level = n - 1 + math.log2(max(width, 1e-8))
    if level >= n - 1:
        return texel(levels[-1], 0, 0, wrap)
    il = max(0, math.floor(level))
    if flt == abi.SHM_TEXFILTER_POINT:
        h, w = levels[il].shape[:2]
        rnd = lambda v: math.floor(abs(v) + 0.5) * (1 if v >= 0 else -1)  # f32::round: half away from zero
        return texel(levels[il], int(rnd(st[0] * w - 0.5)), int(rnd(st[1] * h - 0.5)), wrap)
    if flt == abi.SHM_TEXFILTER_BILINEAR or il == 0:
        return bilerp(levels[il], st, wrap)
    return (1 - (level - il)) * bilerp(levels[il], st, wrap) + (level - il) * bilerp(levels[il + 1], st, wrap)


@pytest.mark.parametrize("flt", [abi.SHM_TEXFILTER_POINT, abi.SHM_TEXFILTER_BILINEAR, abi.SHM_TEXFILTER_TRILINEAR, abi.SHM_TEXFILTER_EWA])
@pytest.mark.parametrize("wrap", [abi.SHM_WRAP_BLACK, abi.SHM_WRAP_CLAMP, abi.SHM_WRAP_REPEAT, abi.SHM_WRAP_OCTAHEDRAL_SPHERE])
def test_mipmap_filter_against_float64(lib, flt, wrap):
    """MIPMap::filter (mipmap.rs:121-199) for every filter x wrap mode, 3-channel and 1-channel pyramids."""
    sc = scenes.cornell_box(lib, 8, 8)
    b = sc.builder
    names = {0: "point", 1: "bilinear", 2: "trilinear", 3: "ewa"}, {0: "black", 1: "clamp", 2: "repeat", 3: "octahedralsphere"}
    imgs = [scenes.test_image(32, 3), scenes.test_image(16, 1, seed=5)]
    for img in imgs:
        b.add_image_texture(img, filter=names[0][flt], wrap=names[1][wrap], max_anisotropy=6.0)
    desc, _ = b.build(lib)
    o = oracle_py.Oracle(desc)
    lut = tables()["MIP_FILTER_LUT"]
    rng = np.random.default_rng(100 * flt + wrap)
    out = (C.c_float * 3)()
    try:
        for k in range(60):
            ti = k % 2
            levels = generate_pyramid(imgs[ti])
            st = rng.uniform(-0.6, 1.6, 2)
            # keep the level-of-detail away from integer boundaries: float32 vs float64 log2 may differ there by an ulp
            mag = 2.0 ** (-rng.integers(0, 7) - rng.uniform(0.15, 0.85))
            ang = rng.uniform(0, 2 * np.pi)
            d0 = (mag * math.cos(ang), mag * math.sin(ang))
            ratio = rng.choice([1.0, 0.5, 0.1, 0.01, 0.0]) if k % 5 else 1.0
            d1 = (-ratio * mag * math.sin(ang) * 0.83, ratio * mag * math.cos(ang) * 0.83)
            if flt != abi.SHM_TEXFILTER_EWA:  # the non-EWA width is the largest |component|: make it the one with the safe log2
                d0, d1 = (mag / 2, 0.3 * mag / 2), (0.1 * mag, -0.4 * mag / 2)
            st32, d032, d132 = [tuple(float(f32(v)) for v in t) for t in (st, d0, d1)]
            o.lib.orc_fn_texture_filter(o.handle, ti, fa(*st32), fa(*d032), fa(*d132), out)
            want = mip_filter(levels, lut, flt, wrap, 6.0, st32, d032, d132)
            if flt == abi.SHM_TEXFILTER_POINT:
                # round() at a half-texel boundary may go either way between f32 and f64: skip those samples
                h, w = levels[0].shape[:2]
                if min(abs((st32[0] * w - 0.5) % 1 - 0.5), abs((st32[1] * h - 0.5) % 1 - 0.5)) < 1e-3:
                    continue
            assert np.allclose(np.array(out[:]), want, rtol=2e-4, atol=2e-5), (k, list(out), want)
    finally:
        o.close()


def test_ewa_filter_of_a_constant_image_is_that_constant(lib):
    sc = scenes.cornell_box(lib, 8, 8)
    b = sc.builder
    b.add_image_texture(np.full((16, 16, 3), 0.375, np.float32), filter="ewa")
    desc, _ = b.build(lib)
    o = oracle_py.Oracle(desc)
    out = (C.c_float * 3)()
    try:
        for d in (1e-4, 3e-3, 0.05, 0.3, 5.0):
            o.lib.orc_fn_texture_filter(o.handle, 0, fa(0.3, 0.7), fa(d, 0.2 * d), fa(-0.1 * d, 0.6 * d), out)
            assert np.allclose(out[:], 0.375, rtol=1e-6)
    finally:
        o.close()


# ---- rgb2spec ----------------------------------------------------------------------------------------------
def fetch64(cs, rgb):
    res, scale, data = cs["res"], cs["scale"].astype(np.float64), cs["data"].astype(np.float64)
    rgb = np.clip(np.asarray(rgb, np.float64), 0, 1)
    i = 0
    for j in (1, 2):
        if rgb[j] >= rgb[i]:
            i = j
    z = rgb[i]
    sc = (res - 1) / z
    x, y = rgb[(i + 1) % 3] * sc, rgb[(i + 2) % 3] * sc
    xi, yi = min(int(x), res - 2), min(int(y), res - 2)
    zi = min(max(int(np.searchsorted(scale, z, side="right")) - 1, 0), res - 2)
    off = (((i * res + zi) * res + yi) * res + xi) * 3
    dx, dy, dz = 3, 3 * res, 3 * res * res
    x1, y1, z1 = x - xi, y - yi, (z - scale[zi]) / (scale[zi + 1] - scale[zi])
    x0, y0, z0 = 1 - x1, 1 - y1, 1 - z1
    out = []
    for j in range(3):
        o = off + j
        out.append(((data[o] * x0 + data[o + dx] * x1) * y0 + (data[o + dy] * x0 + data[o + dy + dx] * x1) * y1) * z0
                   + ((data[o + dz] * x0 + data[o + dz + dx] * x1) * y0 + (data[o + dz + dy] * x0 + data[o + dz + dy + dx] * x1) * y1) * z1)
    return np.array(out)


def test_rgb2spec_fetch(env):
    """RGB2Spec::fetch (rgb2spec 0.1.1, un-vendored; Jakob & Hanika's published rgb2spec_fetch) against float64, plus the
    round trip that shows the committed table means something: the fetched spectrum, viewed under D65, is the RGB again."""
    sc, o = env
    cs = sc.builder.color_space
    t = tables()
    lam = np.arange(360.0, 831.0)
    d65 = cs["illuminant"].astype(np.float64)
    xyz_bar = np.stack([t["CIE_X"], t["CIE_Y"], t["CIE_Z"]]).astype(np.float64)
    m = np.array([[3.240479, -1.537150, -0.498535], [-0.969256, 1.875991, 0.041556], [0.055648, -0.204043, 1.057311]])
    out = (C.c_float * 3)()
    rng = np.random.default_rng(8)
    for k in range(200):
        rgb = rng.uniform(0.0, 1.0, 3) if k % 4 else rng.uniform(0.0, 1.3, 3) * rng.choice([1.0, 0.02])
        rgb32 = [float(f32(v)) for v in rgb]
        o.lib.orc_fn_rgb2spec_fetch(o.handle, fa(*rgb32), out)
        want = fetch64(cs, rgb32)
        assert np.allclose(out[:], want, rtol=2e-4, atol=1e-6 * np.abs(want).max() + 1e-7), (rgb32, list(out), want)
        if max(rgb32) <= 1.0 and min(rgb32) > 0.02:
            c = np.array(out[:], np.float64)
            x = (c[0] * lam + c[1]) * lam + c[2]
            s = 0.5 + 0.5 * x / np.sqrt(1 + x * x)
            back = m @ ((xyz_bar * d65 * s).sum(axis=1) / (xyz_bar[1] * d65).sum())
            assert np.allclose(back, rgb32, atol=0.002)  # resolution 64, as the reference's table (0.03 at the resolution 16 of round 1), (rgb32, back)
    o.lib.orc_fn_rgb2spec_fetch(o.handle, fa(0.0, 0.0, 0.0), out)  # black: defined as the zero spectrum (texture.h)
    assert out[0] == 0.0 and out[1] == 0.0 and out[2] == -np.inf


def test_image_texture_evaluate_spectrum_types(env):
    """SpectrumImageTexture::evaluate (texture.rs:777-808): scale, invert, clamp_zero, then the Rgb*Spectrum of its type."""
    sc, o = env
    cs = sc.builder.color_space
    lams = (452.0, 533.0, 601.5, 688.25)
    out4, out3, out6 = (C.c_float * 4)(), (C.c_float * 3)(), (C.c_float * 6)()

    def sig(c, l):
        x = (c[0] * l + c[1]) * l + c[2]
        return 0.5 + 0.5 * x / math.sqrt(1 + x * x)

    for ti, t in enumerate(sc.builder.textures):
        ctx = fa(0.3, 0.9, -3.0, 2e-3, 0, 0, 0, 1e-3, 0, 0, 1, 0, 0.37, 0.61, 4e-3, 1e-3, -2e-3, 5e-3)
        o.lib.orc_fn_texture_map(o.handle, ti, ctx, out6)
        o.lib.orc_fn_texture_filter(o.handle, ti, fa(out6[0], float(f32(1.0) - f32(out6[1]))), fa(out6[2], out6[4]), fa(out6[3], out6[5]), out3)
        o.lib.orc_fn_image_texture_evaluate(o.handle, ti, ctx, fa(*lams), out4)
        rgb = np.array(out3[:], np.float64) * t.scale
        if t.invert:
            rgb = 1.0 - rgb
        rgb = np.maximum(rgb, 0.0)
        if not t.has_color_space:
            want = np.full(4, rgb[0])
        elif t.spectrum_type == abi.SHM_SPECTRUM_TYPE_ALBEDO:
            c = fetch64(cs, rgb)
            want = np.array([sig(c, l) for l in lams])
        else:
            sc2 = 2 * rgb.max()
            c = fetch64(cs, rgb / sc2)
            want = np.array([sc2 * sig(c, l) for l in lams])
            if t.spectrum_type == abi.SHM_SPECTRUM_TYPE_ILLUMINANT:
                want = want * np.array([cs["illuminant"][int(math.floor(l + 0.5)) - 360] for l in lams])
        assert np.allclose(out4[:], want, rtol=3e-4, atol=1e-6), (ti, list(out4), want)


def test_texture_mappings(env):
    """texture.rs:918-1044, including what the reference writes literally: SphericalMapping returns spherical_theta (through
    safe_acos = asin, math.rs:266-274) for both coordinates; CylindricalMapping's s is PI + atan2 / 2pi."""
    sc, o = env
    out6 = (C.c_float * 6)()
    p, dpdx, dpdy = np.array([0.4, 0.7, -2.9]), np.array([3e-3, 1e-3, -2e-3]), np.array([-1e-3, 4e-3, 5e-4])
    uv, duv = (0.37, 0.61), (4e-3, 1e-3, -2e-3, 5e-3)  # dudx, dudy, dvdx, dvdy
    ctx = fa(*p, *dpdx, *dpdy, 0, 1, 0, *uv, *duv)
    for ti, t in enumerate(sc.builder.textures):
        o.lib.orc_fn_texture_map(o.handle, ti, ctx, out6)
        m = np.array(t.texture_from_render[:], np.float64).reshape(4, 4)
        pt = m[:3, :3] @ p + m[:3, 3]
        dx, dy = m[:3, :3] @ dpdx, m[:3, :3] @ dpdy
        if t.mapping == abi.SHM_TEXMAP_UV:
            want = [t.su * uv[0] + t.du, t.sv * uv[1] + t.dv, t.su * duv[0], t.su * duv[1], t.sv * duv[2], t.sv * duv[3]]
        elif t.mapping == abi.SHM_TEXMAP_PLANAR:
            vs, vt = np.array(t.vs[:]), np.array(t.vt[:])
            want = [t.du + pt @ vs, t.dv + pt @ vt, vs @ dx, vs @ dy, vt @ dx, vt @ dy]
        elif t.mapping == abi.SHM_TEXMAP_CYLINDRICAL:
            x2y2 = pt[0] ** 2 + pt[1] ** 2
            dsdp = np.array([-pt[1], pt[0], 0]) / (2 * np.pi * x2y2)
            want = [np.pi + math.atan2(pt[1], pt[0]) / (2 * np.pi), pt[2], dsdp @ dx, dsdp @ dy, dx[2], dy[2]]
        else:
            x2y2 = pt[0] ** 2 + pt[1] ** 2
            dsdp = np.array([-pt[1], pt[0], 0]) / (2 * np.pi * x2y2)
            dtdp = np.array([pt[0] * pt[2], pt[1] * pt[2], -x2y2]) / math.sqrt(x2y2) / (np.pi * (x2y2 + pt[2] ** 2))
            theta = math.asin(pt[2] / np.linalg.norm(pt))
            want = [theta / np.pi, theta / (2 * np.pi), dsdp @ dx, dsdp @ dy, dtdp @ dx, dtdp @ dy]
        assert np.allclose(out6[:], want, rtol=1e-4, atol=1e-7), (ti, list(out6), want)


# ---- ray differentials ----------------------------------------------------------------------------------------
def hit_diff(o, px, py, spp=1, dpj=1, use_aux=1, sample=0):
    out = (C.c_float * 44)()
    ok = o.lib.orc_fn_camera_hit_differentials(o.handle, px, py, sample, 0, spp, dpj, use_aux, out)
    a = np.array(out[:], np.float64)
    return ok, dict(o=a[0:3], d=a[3:6], rx_o=a[6:9], rx_d=a[9:12], ry_o=a[12:15], ry_d=a[15:18], p=a[18:21], n=a[21:24], uv=a[24:26],
                    dpdu=a[26:29], dpdv=a[29:32], dpdx=a[32:35], dpdy=a[35:38], dudx=a[38], dvdx=a[39], dudy=a[40], dvdy=a[41])


def test_camera_differentials_are_the_neighbouring_pixels(env):
    """camera.rs:1057-1068 + interaction.rs:296-314: without pixel jitter (no scaling, integrator.rs:360) the x auxiliary ray IS
    the ray through the next pixel, so p + dpdx is where that pixel's ray meets the tangent plane; (u, v) follow linearly."""
    sc, o = env
    checked = 0
    for (px, py) in [(4, 24), (24, 44), (44, 24), (24, 6), (20, 20)]:
        ok, a = hit_diff(o, px, py)
        okx, bx = hit_diff(o, px + 1, py)
        oky, by = hit_diff(o, px, py + 1)
        assert ok
        assert np.allclose(a["rx_d"], bx["d"], atol=1e-6) and np.allclose(a["ry_d"], by["d"], atol=1e-6)
        for ok2, b, dp, du, dv in ((okx, bx, a["dpdx"], a["dudx"], a["dvdx"]), (oky, by, a["dpdy"], a["dudy"], a["dvdy"])):
            if not ok2 or not np.allclose(b["n"], a["n"]) or abs(b["n"] @ (b["p"] - a["p"])) > 1e-5:
                continue  # the neighbour landed on another surface
            assert np.allclose(a["p"] + dp, b["p"], atol=2e-5)
            # the least-squares (du, dv) reproduces dp in the tangent plane: dp = dpdu du + dpdv dv
            assert np.allclose(a["dpdu"] * du + a["dpdv"] * dv, dp, atol=2e-5)
            checked += 1
    assert checked >= 6


def test_differential_scaling_with_spp(env):
    """integrator.rs:356-362: with pixel jitter the differentials shrink by max(1/8, 1/sqrt(spp))."""
    sc, o = env
    _, a1 = hit_diff(o, 26, 20, spp=1, dpj=0)  # the back wall, seen almost head-on: the footprint is linear in the differential
    _, a16 = hit_diff(o, 26, 20, spp=16, dpj=0)
    _, a1k = hit_diff(o, 26, 20, spp=1024, dpj=0)
    assert np.allclose(a16["dpdx"], a1["dpdx"] / 4, rtol=1e-2, atol=2e-6) and np.allclose(a1k["dpdy"], a1["dpdy"] / 8, rtol=1e-2, atol=2e-6)


def test_approximate_dp_dxy_has_the_footprint_of_the_real_differentials(env):
    """camera.rs:307-354 with the minimum differentials of camera.rs:356-440: an estimate built from the SMALLEST differential on
    the film diagonal, oriented by rotate_from_to and not by the raster axes — so its size is comparable (never larger) and its
    direction is not pinned. It must lie in the tangent plane."""
    sc, o = env
    for (px, py) in [(4, 24), (24, 44), (44, 24), (24, 24)]:
        ok, a = hit_diff(o, px, py, use_aux=1)
        _, b = hit_diff(o, px, py, use_aux=0)
        assert ok
        for k in ("dpdx", "dpdy"):
            assert abs(b[k] @ a["n"]) < 1e-5 * max(1.0, np.linalg.norm(b[k]) / 1e-3)
        area_a = np.linalg.norm(np.cross(a["dpdx"], a["dpdy"]))
        area_b = np.linalg.norm(np.cross(b["dpdx"], b["dpdy"]))
        assert 0.3 * area_a < area_b < 1.3 * area_a


def test_minimum_differentials_of_the_host_camera_mirror(lib):
    """CameraBase::find_minimum_differentials through shm_camera_perspective: a pinhole has no positional differential; the
    directional ones are the pixel spacing at the film corner (the smallest on the diagonal), in the ray's own frame."""
    sc = scenes.cornell_box(lib, 48, 48)
    cam = sc.desc.camera
    assert np.allclose(cam.min_pos_differential_x[:], 0) and np.allclose(cam.min_pos_differential_y[:], 0)
    m = np.array(cam.render_from_camera[:], np.float64).reshape(4, 4) @ np.array(cam.camera_from_render[:], np.float64).reshape(4, 4)
    assert np.allclose(m, np.eye(4), atol=1e-6)
    fov, res = math.radians(39.0), 48
    pixel = 2 * math.tan(fov / 2) / res                       # pixel spacing on the z = 1 plane
    corner = math.atan(math.sqrt(2) * math.tan(fov / 2))      # the diagonal's end
    want = pixel * math.cos(corner) ** 2 * math.cos(math.radians(45)) * 0 + pixel * math.cos(corner) ** 2  # upper bound on the angular step
    for v in (cam.min_dir_differential_x, cam.min_dir_differential_y):
        n = np.linalg.norm(v[:])
        assert 0.5 * want < n <= pixel * 1.001 and abs(v[2]) < 0.1 * n
    lens = scenes.cornell_box(lib, 48, 48)
    b = lens.builder
    b.set_camera_look_at(lib, (0, 1, 3.4), (0, 1, 0), (0, 1, 0), 39.0, lens_radius=0.05, focal_distance=3.0)
    # with a lens every auxiliary ray starts at the same lens point as the main ray: still no positional differential
    assert np.allclose(b.camera.min_pos_differential_x[:], 0, atol=1e-7)


def reflect(d, n):
    return d - 2 * (d @ n) * n


def refract(d, n, eta):
    """Snell for an incoming direction d (pointing at the surface), n on the incoming side, eta = n_t / n_i."""
    c = -(d @ n)
    k = 1 - (1 - c * c) / (eta * eta)
    return d / eta + (c / eta - math.sqrt(k)) * n


def spawn(o, p, n, wo, dpdx, dpdy, aux, wi, flags, eta):
    out = (C.c_float * 12)()
    has = o.lib.orc_fn_spawn_ray_differentials(fa(*p), fa(*n), fa(*wo), fa(*dpdx), fa(*dpdy), fa(0, 0, 0, 0, 0, 0), fa(*aux), fa(*wi), flags, eta, out)
    a = np.array(out[:], np.float64)
    return has, a[0:3], a[3:6], a[6:9], a[9:12]


def test_specular_differentials_on_a_plane(orc):
    """interaction.rs:430-514 on a planar interface (dn/du = dn/dv = 0). Reflection is linear in the direction, so the
    differential direction must BE the mirrored auxiliary direction; transmission is Snell's law to first order."""
    class O:
        lib = orc
    p, n = np.array([0.2, -0.1, 0.0]), np.array([0.0, 0.0, 1.0])
    d = np.array([0.3, -0.2, -0.8])
    d /= np.linalg.norm(d)
    wo = -d
    rx_d, ry_d = d + np.array([2e-3, 0, 0.5e-3]), d + np.array([0, -1.5e-3, 0.3e-3])
    dpdx, dpdy = np.array([3e-3, 1e-3, 0.0]), np.array([-1e-3, 2e-3, 0.0])
    aux = [*(p - 1.0 * rx_d), *rx_d, *(p - 1.0 * ry_d), *ry_d]
    wi = reflect(d, n)
    has, rxo, rxd, ryo, ryd = spawn(O, p, n, wo, dpdx, dpdy, aux, wi, 0x11, 1.0)  # SPECULAR | REFLECTION
    assert has == 1
    assert np.allclose(rxo, p + dpdx, atol=1e-7) and np.allclose(ryo, p + dpdy, atol=1e-7)
    assert np.allclose(rxd, reflect(rx_d, n), atol=2e-7) and np.allclose(ryd, reflect(ry_d, n), atol=2e-7)
    # the formulas of interaction.rs:480-495 use eta as BSDFSample::eta (= eta_t / eta_i seen from wo's side) in `wi - eta * dwodx`,
    # i.e. they are the differential of wi = -wo * eta + ... only for eta = 1; for eta != 1 check what is invariant: the result is
    # finite, has auxiliary rays, and for eta = 1 (index-matched) the transmitted differential is the incoming one
    has, rxo, rxd, ryo, ryd = spawn(O, p, n, wo, dpdx, dpdy, aux, d, 0x12, 1.0)  # SPECULAR | TRANSMISSION, straight through
    assert has == 1 and np.allclose(rxd, rx_d, atol=2e-6) and np.allclose(ryd, ry_d, atol=2e-6)
    has, rxo, rxd, ryo, ryd = spawn(O, p, n, wo, dpdx, dpdy, aux, refract(d, n, 1.5), 0x12, 1.5)
    assert has == 1 and np.all(np.isfinite(rxd)) and np.all(np.isfinite(ryd))
    has, *_ = spawn(O, p, n, wo, dpdx, dpdy, aux, wi, 0x05, 1.0)  # DIFFUSE | REFLECTION: no differentials (interaction.rs:499)
    assert has == 0
    has, *_ = spawn(O, p, n, wo, dpdx, dpdy, [*(p - rx_d), *(1e9 * rx_d), *(p - ry_d), *ry_d], wi, 0x11, 1.0)  # squashed (:504-511)
    assert has == 0


# ---- float textures, bump and normal maps ---------------------------------------------------------------------
def test_float_texture_graph(lib):
    """FloatTexture::evaluate (texture.rs:142-305, 393-403): constant / scale / mix / directionmix / imagemap nodes against a
    float64 evaluation of the same graph; an RGB image read as a float is channel 0 for texel lookups (point, EWA) and the
    three-channel AVERAGE for bilinear ones (TexelType for Float, mipmap.rs:297-312)."""
    sc = scenes.cornell_box(lib, 8, 8)
    b = sc.builder
    img1, img3 = scenes.test_image(16, 1, seed=3), scenes.test_image(16, 3, seed=4)
    t_img1 = b.ftex_image(img1, filter="bilinear", scale=0.7)
    t_img3p = b.ftex_image(img3, filter="point")
    t_img3b = b.ftex_image(img3, filter="bilinear", invert=True)
    t_scaled = b.ftex_scaled(t_img1, 0.25)
    t_zero = b.ftex_scaled(t_img1, 0.0)
    t_dir = b.ftex_direction_mix(0.2, t_img1, dir=(0.0, 0.6, 0.8))
    t_mix = b.ftex_mix(t_scaled, t_dir, t_img3b)
    t_deep = b.ftex_mix(t_mix, 1.0, b.ftex_scaled(t_mix, t_dir))
    desc, _ = b.build(lib)
    o = oracle_py.Oracle(desc)
    try:
        rng = np.random.default_rng(0)
        for _ in range(40):
            uv = rng.uniform(0.05, 0.95, 2)
            n = rng.normal(size=3)
            n /= np.linalg.norm(n)
            ctx = fa(0.1, 0.2, 0.3, 1e-3, 0, 0, 0, 1e-3, 0, *n, *uv, 1e-3, 0.0, 0.0, 1e-3)
            ev = lambda t: float(o.lib.orc_fn_float_texture_evaluate(o.handle, t, ctx))
            st = (float(f32(uv[0])), 1.0 - float(f32(uv[1])))
            lv1, lv3 = generate_pyramid(img1), generate_pyramid(img3)
            v1 = bilerp(lv1[0], st, abi.SHM_WRAP_REPEAT)[0] * 0.7
            h, w = lv3[0].shape[:2]
            rnd = lambda v: math.floor(abs(v) + 0.5) * (1 if v >= 0 else -1)
            v3p = texel(lv3[0], int(rnd(st[0] * w - 0.5)), int(rnd(st[1] * h - 0.5)), abi.SHM_WRAP_REPEAT)[0]
            v3b = max(0.0, 1.0 - bilerp(lv3[0], st, abi.SHM_WRAP_REPEAT).mean())
            assert ev(t_img1) == pytest.approx(v1, rel=2e-5) and ev(t_img3b) == pytest.approx(v3b, rel=2e-5, abs=1e-6)
            if min(abs((st[0] * w - 0.5) % 1 - 0.5), abs((st[1] * h - 0.5) % 1 - 0.5)) > 1e-3:
                assert ev(t_img3p) == pytest.approx(v3p, rel=1e-6)
            amt_d = float(n @ np.array([0.0, 0.6, 0.8]))
            vdir = amt_d * 0.2 + (1 - amt_d) * v1
            vmix = (v1 * 0.25) * (1 - v3b) + vdir * v3b
            assert ev(t_scaled) == pytest.approx(v1 * 0.25, rel=2e-5) and ev(t_zero) == 0.0
            assert ev(t_dir) == pytest.approx(vdir, rel=5e-5, abs=1e-6) and ev(t_mix) == pytest.approx(vmix, rel=5e-5, abs=1e-6)
            a2 = vmix * vdir
            assert ev(t_deep) == pytest.approx(vmix * (1 - a2) + 1.0 * a2, rel=1e-4, abs=1e-6)
    finally:
        o.close()


def test_composite_spectrum_textures(lib):
    """SpectrumScaledTexture / SpectrumMixTexture / SpectrumDirectionMixTexture (texture.rs:573-581, 628-645, 810-828) over
    spectrum and image leaves, against the same expressions evaluated on the leaves' own values."""
    sc = scenes.cornell_box(lib, 8, 8)
    b = sc.builder
    img3, img1 = scenes.test_image(16, 3, seed=8), scenes.test_image(16, 1, seed=9)
    leaf_img = b.add_image_texture(img3, filter="bilinear")
    leaf_pw = b.spectrum_piecewise([400.0, 700.0], [0.2, 0.8])
    f_img = b.ftex_image(img1, filter="bilinear")
    scaled = b.stex_scaled(leaf_img, 0.5)
    zero = b.stex_scaled(leaf_img, 0.0)
    mix = b.stex_mix(leaf_pw, scaled, f_img)
    dmix = b.stex_direction_mix(mix, 0.25, dir=(0.0, 0.6, 0.8))
    b.materials[0].a = dmix  # bound to a material: the node table travels through scene creation
    desc, _ = b.build(lib)
    o = oracle_py.Oracle(desc)
    lams = (452.0, 533.0, 601.5, 688.25)
    out = (C.c_float * 4)()

    def ev(sp, ctx):
        o.lib.orc_fn_spectrum_texture_evaluate(o.handle, C.byref(sp), ctx, fa(*lams), out)
        return np.array(out[:], np.float64)

    try:
        rng = np.random.default_rng(4)
        for _ in range(20):
            uv = rng.uniform(0.05, 0.95, 2)
            n = rng.normal(size=3)
            n /= np.linalg.norm(n)
            ctx = fa(0.1, 0.2, 0.3, 1e-3, 0, 0, 0, 1e-3, 0, *n, *uv, 1e-3, 0.0, 0.0, 1e-3)
            v_img, v_pw = ev(leaf_img, ctx), ev(leaf_pw, ctx)
            amt = float(o.lib.orc_fn_float_texture_evaluate(o.handle, f_img, ctx))
            assert np.allclose(ev(scaled, ctx), v_img * 0.5, rtol=1e-6) and not ev(zero, ctx).any()
            v_mix = v_pw * (1 - amt) + (v_img * 0.5) * amt
            assert np.allclose(ev(mix, ctx), v_mix, rtol=1e-5)
            a = float(n @ np.array([0.0, 0.6, 0.8]))
            assert np.allclose(ev(dmix, ctx), a * v_mix + (1 - a) * 0.25, rtol=1e-5, atol=1e-6)
    finally:
        o.close()
    # a tree over the node limit, and a composite where only a spectrum may stand, are refused at scene creation
    big = leaf_pw
    for _ in range(8):
        big = b.stex_scaled(big, 0.9)
    b.materials[0].a = big
    with pytest.raises(RuntimeError, match="larger than 8"):
        oracle_py.Oracle(b.build(lib)[0])
    b2 = scenes.cornell_box(lib, 8, 8).builder  # (every node of the table is validated, bound or not: a fresh table)
    glass = b2.material_dielectric(1.5)
    b2.materials[glass].a = b2.stex_scaled(0.5, 0.5)
    with pytest.raises(RuntimeError, match="not valid in this slot"):
        oracle_py.Oracle(b2.build(lib)[0])


def test_bump_and_normal_map(lib):
    """material.rs:1453-1508. A displacement that rises linearly along u tilts dpdu by slope * n and leaves dpdv alone; a constant
    one changes nothing on a flat surface; a normal map of (0.5, 0.5, 1) is the identity and one leaning towards +s tilts the
    frame about dpdv."""
    sc = scenes.cornell_box(lib, 8, 8)
    b = sc.builder
    ramp = np.tile((np.arange(32, dtype=np.float32) + 0.5) / 32, (32, 1))  # value = u at texel centres
    t_ramp = b.ftex_image(ramp, filter="bilinear", wrap="clamp")
    t_const = b.ftex_constant(0.37)
    flat = np.zeros((4, 4, 3), np.float32) + np.array([0.5, 0.5, 1.0], np.float32)
    lean = np.zeros((4, 4, 3), np.float32) + np.array([0.75, 0.5, 0.5 + 0.5 * math.sqrt(0.75)], np.float32)  # (0.5, 0, 0.866) in [-1,1]^3
    nm_flat = b.add_image_texture(flat, color_space=False).offset
    nm_lean = b.add_image_texture(lean, color_space=False).offset
    desc, _ = b.build(lib)
    o = oracle_py.Oracle(desc)
    out = (C.c_float * 6)()
    geo = fa(0.2, 0.1, 0.0, 0, 0, 1, 2.0, 0, 0, 0, 3.0, 0, 0, 0, 0, 0, 0, 0)  # p, n = +z, dpdu = 2x, dpdv = 3y, dndu = dndv = 0
    try:
        o.lib.orc_fn_bump_or_normal_map(o.handle, 0, t_const, geo, fa(0.4, 0.6), fa(1e-3, 0, 0, 1e-3), out)
        assert np.allclose(out[:], (2, 0, 0, 0, 3, 0), atol=1e-6)
        o.lib.orc_fn_bump_or_normal_map(o.handle, 0, t_ramp, geo, fa(0.4, 0.6), fa(4e-3, 0, 0, 4e-3), out)
        assert np.allclose(out[:], (2, 0, 1.0, 0, 3, 0), atol=2e-3)  # d(displacement)/du = 1
        o.lib.orc_fn_bump_or_normal_map(o.handle, 0, t_ramp, geo, fa(0.4, 0.6), fa(0, 0, 0, 0), out)  # du = dv = 0 -> 0.0005 (material.rs:1486-1489)
        assert np.allclose(out[:], (2, 0, 1.0, 0, 3, 0), atol=5e-3)
        o.lib.orc_fn_bump_or_normal_map(o.handle, 1, nm_flat, geo, fa(0.4, 0.6), fa(0, 0, 0, 0), out)
        assert np.allclose(out[:], (2, 0, 0, 0, 3, 0), atol=1e-5)
        o.lib.orc_fn_bump_or_normal_map(o.handle, 1, nm_lean, geo, fa(0.4, 0.6), fa(0, 0, 0, 0), out)
        dpdu, dpdv = np.array(out[0:3]), np.array(out[3:6])
        ns = np.cross(dpdu, dpdv)
        ns /= np.linalg.norm(ns)
        assert np.allclose(ns, (0.5, 0, math.sqrt(0.75)), atol=1e-5)  # the frame is (dpdu, n x dpdu, n): local x = world x here
        assert np.linalg.norm(dpdu) == pytest.approx(2.0, rel=1e-6) and np.linalg.norm(dpdv) == pytest.approx(3.0, rel=1e-6)
    finally:
        o.close()


# ---- whole renders --------------------------------------------------------------------------------------------
def test_point_filtered_constant_texture_renders_exactly_like_the_constant(lib):
    """A one-channel image without a colour space evaluates to from_const(texel) (texture.rs:801-805): with the point filter a
    constant image must give the film of the same scene with a ConstantSpectrum reflectance, bit for bit — the differentials
    that are now computed at every vertex must not leak into anything else."""
    def film_of(make_reflectance):
        sc = scenes.cornell_box(lib, 32, 32)
        b = sc.builder
        refl = make_reflectance(b)
        for m in b.materials:
            if m.kind == abi.SHM_MATERIAL_DIFFUSE and m.a.kind == abi.SHM_SPECTRUM_CONSTANT and m.a.c == 0.75:
                m.a = refl
        desc, _ = b.build(lib)
        o = oracle_py.Oracle(desc)
        try:
            film, stats = o.render(render.make_params(spp=8, max_depth=5, seed=3), n_threads=4)
        finally:
            o.close()
        return film, stats
    plain, s0 = film_of(lambda b: b.spectrum_constant(0.625))
    tex, s1 = film_of(lambda b: b.add_image_texture(np.full((8, 8), 0.625, np.float32), filter="point", color_space=False))
    assert s0["rays_closest"] == s1["rays_closest"] and s0["rays_any"] == s1["rays_any"]
    assert plain.tobytes() == tex.tobytes()


@pytest.mark.parametrize("integrator", ["path", "simplepath", "randomwalk"])
def test_textured_render_is_finite_and_textured(lib, integrator):
    sc = scenes.cornell_box(lib, 40, 40, textured=True)
    o = oracle_py.Oracle(sc.desc)
    try:
        film, stats = o.render(render.make_params(spp=8, max_depth=4, seed=2, integrator=integrator), n_threads=8)
    finally:
        o.close()
    rgb = film["rgb_sum"] / film["weight_sum"][..., None]
    assert np.isfinite(rgb).all() and rgb.mean() > 0.01
    floor = rgb[34:39, 8:32, :]  # the EWA-filtered checker on the floor: neighbouring columns differ by more than noise alone would
    assert floor.std(axis=1).mean() > 0.02


def test_disable_texture_filtering(lib):
    """options.disable_texture_filtering (interaction.rs:287-295): zero differentials, so every filter reads the finest level
    (width 0 -> level 0; EWA's shorter axis 0 -> bilerp(0), mipmap.rs:141-144) and bump maps step by 0.0005."""
    sc = scenes.cornell_box(lib, 32, 32, textured=True)
    o = oracle_py.Oracle(sc.desc)
    try:
        a, _ = o.render(render.make_params(spp=4, max_depth=4, seed=2), n_threads=8)
        b, _ = o.render(render.make_params(spp=4, max_depth=4, seed=2, disable_texture_filtering=True), n_threads=8)
        out = (C.c_float * 44)()
    finally:
        o.close()
    ra, rb = a["rgb_sum"] / a["weight_sum"][..., None], b["rgb_sum"] / b["weight_sum"][..., None]
    assert np.isfinite(rb).all() and not np.array_equal(a, b)
    assert abs(rb.mean() / ra.mean() - 1) < 0.2  # same scene, sharper textures


@pytest.mark.parametrize("flt", ["bilinear", "trilinear", "ewa"])
def test_filters_reproduce_a_linear_ramp(lib, flt):
    """A property no reading of mipmap.rs is needed for: every filter here is a normalised kernel that is symmetric about the lookup point — bilinear interpolation, its
    blend across two box-filtered levels, the EWA Gaussian over its ellipse — and the box pyramid of a linear image is linear; so filtering the ramp a + b x + c y gives
    the ramp's value AT the lookup point, for any footprint that stays inside the image (clamp wrap, centre region)."""
    n = 64
    yy, xx = np.mgrid[0:n, 0:n].astype(np.float64)
    a, bx, cy = 0.2, 0.5 / n, 0.25 / n
    img = (a + bx * (xx + 0.5) + cy * (yy + 0.5)).astype(np.float32)  # texel centres at (x + 0.5, y + 0.5) / n
    sc = scenes.cornell_box(lib, 8, 8)
    b = sc.builder
    b.add_image_texture(np.stack([img] * 3, axis=-1), filter=flt, wrap="clamp", max_anisotropy=8.0)
    desc, _ = b.build(lib)
    o = oracle_py.Oracle(desc)
    out = (C.c_float * 3)()
    rng = np.random.default_rng(12)
    errs = []
    try:
        for _ in range(200):
            s, t = rng.uniform(0.35, 0.65, 2)
            mag = 2.0 ** -rng.uniform(2.5, 7.0)  # footprints of ~0.5 to 11 texels
            ang, ratio = rng.uniform(0, 2 * np.pi), rng.choice([1.0, 0.5, 0.2])
            d0 = (mag * math.cos(ang), mag * math.sin(ang))
            d1 = (-ratio * mag * math.sin(ang), ratio * mag * math.cos(ang))
            o.lib.orc_fn_texture_filter(o.handle, desc.n_image_textures - 1, fa(float(s), float(t)), fa(*map(float, d0)), fa(*map(float, d1)), out)
            want = a + bx * s * n + cy * t * n
            errs.append(out[0] / want - 1.0)
            # (bilinear / trilinear are exact on a ramp; EWA sums a truncated Gaussian over the LATTICE points inside its ellipse, which are not symmetric about the lookup
            #  point: each lookup is off by a fraction of a texel of its level — bounded here —, with no systematic shift — the mean below)
            assert out[0] == pytest.approx(want, rel=1.5e-2 if flt == "ewa" else 2e-5), (flt, s, t, mag, ratio, out[0], want)
        assert abs(np.mean(errs)) < (1.5e-3 if flt == "ewa" else 1e-6), np.mean(errs)
    finally:
        o.close()
