"""The image side of the C++ scene front end (shimmer_amd/csrc/host/image_io.hpp behind shm_image_load_png; reference: image.rs:1140-1311
Image::read_png, color.rs:420-724 ColorEncoding, image.rs:699-802 generate_pyramid, image.rs:1007-1138 float_resize_up, mipmap.rs:42-99)
against INDEPENDENT restatements: PNG files are written here with Python's zlib (every scanline filter, both bit depths, the four colour
types the reference accepts, stored / fixed / dynamic deflate blocks, split IDAT chunks) and must decode to the samples that went in; the
sRGB encode / decode pair, the quantised MIP pyramid and the (reference-shaped) up-sampling are re-evaluated in numpy float32."""
import ctypes as C
import struct
import zlib

import numpy as np
import pytest

from shimmer_amd import abi
from shimmer_amd.scene import tables

f32 = np.float32


def _chunk(t, d):
    return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def write_png(path, arr, depth, ctype, filters=(0,), level=6, split=1, interlace=0, extra=b""):
    """A PNG writer that exercises the decoder: `filters` cycles over the scanlines, `split` cuts the zlib stream into several IDAT chunks."""
    h, w = arr.shape[:2]
    nch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    a = np.asarray(arr).reshape(h, w, nch)
    raw = a.astype(">u2").tobytes() if depth == 16 else a.astype(np.uint8).tobytes()
    bpp = nch * depth // 8
    stride = w * bpp
    out = bytearray()
    prev = bytes(stride)
    for y in range(h):
        row = raw[y * stride:(y + 1) * stride]
        f = filters[y % len(filters)]
        enc = bytearray(stride)
        for i in range(stride):
            A = row[i - bpp] if i >= bpp else 0
            B = prev[i]
            Cc = prev[i - bpp] if i >= bpp else 0
            enc[i] = (row[i] - [0, A, B, (A + B) >> 1, _paeth(A, B, Cc)][f]) & 255
        out.append(f)
        out += enc
        prev = row
    z = zlib.compress(bytes(out), level)
    parts = [z[i * len(z) // split:(i + 1) * len(z) // split] for i in range(split)]
    with open(path, "wb") as fp:
        fp.write(b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, interlace)) + extra
                 + b"".join(_chunk(b"IDAT", p) for p in parts) + _chunk(b"tEXt", b"k\0v") + _chunk(b"IEND", b""))


def load_png(lib, path, encoding="sRGB", wrap=abi.SHM_WRAP_REPEAT, pyramid=False):
    im = abi.ShmLoadedImage()
    rc = lib.shm_image_load_png(str(path).encode(), encoding.encode() if encoding else None, wrap, int(pyramid), C.byref(im))
    if rc != 0:
        raise abi.ShimmerHipError(f"{rc}: {lib.shm_last_error().decode()}")
    try:
        tex = np.ctypeslib.as_array(im.texels, shape=(im.n_texel_floats,)).copy()
        levels = []
        for i in range(im.n_levels):
            lv = im.levels[i]
            levels.append(tex[lv.texel_offset:lv.texel_offset + lv.width * lv.height * im.n_channels].reshape(lv.height, lv.width, im.n_channels))
        return levels, dict(n_channels=im.n_channels, file_channels=im.file_channels, has_color_space=im.has_color_space)
    finally:
        lib.shm_image_free(C.byref(im))


@pytest.fixture(scope="module")
def lib():
    return abi.load_library()


@pytest.mark.parametrize("depth,ctype,w,h,level", [(8, 2, 16, 8, 6), (8, 6, 5, 7, 9), (8, 0, 33, 3, 1), (16, 2, 8, 8, 6), (16, 6, 3, 5, 6), (8, 4, 4, 4, 0),
                                                 (8, 2, 64, 64, 9), (8, 0, 1, 1, 6)])
def test_png_decode(lib, tmp_path, depth, ctype, w, h, level):
    """Samples in = samples out (linear encoding: an 8-bit sample v is v / 255; a 16-bit RGB sample v is f16(v / 65535), image.rs:1246-1262)."""
    nch = {0: 1, 2: 3, 4: 2, 6: 4}[ctype]
    rng = np.random.default_rng(depth * 100 + ctype * 10 + w)
    arr = rng.integers(0, 256 if depth == 8 else 65536, size=(h, w, nch))
    if w == 64:  # compressible content: long matches and dynamic Huffman tables
        arr = (np.add.outer(np.arange(h), np.arange(w))[:, :, None] // 4 % 7 * 30 + np.arange(nch)).astype(np.int64)
    f = tmp_path / "t.png"
    write_png(f, arr, depth, ctype, filters=(0, 1, 2, 3, 4), level=level, split=3)
    levels, info = load_png(lib, f, "linear")
    assert len(levels) == 1 and info["file_channels"] == (1 if ctype in (0, 4) else nch) and info["has_color_space"] == (0 if ctype in (0, 4) else 1)
    keep = 3 if nch >= 3 else 1
    if depth == 8:
        want = arr[..., :keep].astype(np.float32) / f32(255.0)
    else:
        want = (arr[..., :keep].astype(np.float32) / f32(65535.0)).astype(np.float16).astype(np.float32)
    assert np.array_equal(levels[0], want)


def test_png_16bit_grey_is_read_as_little_endian_f16(lib, tmp_path):
    """The reference's quirk (image.rs:1184-1197): the big-endian 16-bit grey sample's two BYTES are taken as a little-endian f16."""
    vals = np.array([[0x003C, 0x0038, 0x0000, 0x0040]])  # bytes (00 3C) -> f16 0x3C00 = 1.0, (00 38) -> 0.5, 0, (00 40) -> 2.0
    f = tmp_path / "g16.png"
    write_png(f, vals, 16, 0)
    levels, info = load_png(lib, f, "linear")
    assert levels[0].ravel().tolist() == [1.0, 0.5, 0.0, 2.0] and info["n_channels"] == 1 and info["has_color_space"] == 0


def test_png_errors_are_codes(lib, tmp_path):
    arr = np.zeros((2, 2, 3), np.int64)
    im = abi.ShmLoadedImage()

    def rc_of(path):
        return lib.shm_image_load_png(str(path).encode(), None, abi.SHM_WRAP_REPEAT, 0, C.byref(im)), lib.shm_last_error().decode()

    good = tmp_path / "ok.png"
    write_png(good, arr, 8, 2)
    data = good.read_bytes()
    (tmp_path / "sig.png").write_bytes(b"NOPE" + data[4:])
    assert rc_of(tmp_path / "sig.png")[0] == -1 and "not a PNG" in rc_of(tmp_path / "sig.png")[1]
    (tmp_path / "cut.png").write_bytes(data[:40])
    assert rc_of(tmp_path / "cut.png")[0] == -1 and "truncated" in rc_of(tmp_path / "cut.png")[1]
    bad = bytearray(data)
    bad[45] ^= 0x10  # inside the IDAT payload
    (tmp_path / "crc.png").write_bytes(bytes(bad))
    assert rc_of(tmp_path / "crc.png")[0] == -1 and "checksum" in rc_of(tmp_path / "crc.png")[1]
    write_png(tmp_path / "idx.png", np.zeros((2, 2, 1), np.int64), 8, 3, extra=_chunk(b"PLTE", bytes(6)))
    assert rc_of(tmp_path / "idx.png")[0] == -1 and "Indexed" in rc_of(tmp_path / "idx.png")[1]  # image.rs:1296 panics
    write_png(tmp_path / "lace.png", arr, 8, 2, interlace=1)
    assert rc_of(tmp_path / "lace.png")[0] == -2
    assert rc_of(tmp_path / "missing.png")[0] == -1
    (tmp_path / "x.exr").write_bytes(b"")
    assert rc_of(tmp_path / "x.exr")[0] == -2 and "Unsupported file extension" in rc_of(tmp_path / "x.exr")[1]  # image.rs:1147


# ---- colour encodings (color.rs:527-724), restated -----------------------------------------------------------------------------------
def fma(a, b, c):
    return np.float32(np.float64(np.float32(a)) * np.float64(np.float32(b)) + np.float64(np.float32(c)))


def poly_estrin(x, c):
    """fast_polynomial's Estrin scheme with FMAs, as image_io.hpp documents its (unpinned) choice; 5 or 6 coefficients, lowest first."""
    x = f32(x)
    x2 = f32(x * x)
    x4 = f32(x2 * x2)
    lo = fma(fma(c[3], x, c[2]), x2, fma(c[1], x, c[0]))
    return fma(c[4], x4, lo) if len(c) == 5 else fma(fma(c[5], x, c[4]), x4, lo)


def linear_to_srgb(v):
    v = f32(v)
    if v <= f32(0.0031308):
        return f32(f32(12.92) * v)
    s = np.sqrt(v, dtype=np.float32)
    p = [f32(x) for x in (-0.0016829072605308378, 0.03453868659826638, 0.7642611304733891, 2.0041169284241644, 0.7551545191665577, -0.016202083165206348)]
    q = [f32(x) for x in (4.178892964897981e-7, -0.00004375359692957097, 0.03467195408529984, 0.6085338522168684, 1.8970238036421054, 1.0)]
    return f32(f32(poly_estrin(s, p) / poly_estrin(s, q)) * v)


def linear_to_srgb8(v):
    if v <= 0.0:
        return 0
    if v >= 1.0:
        return 255
    t = f32(f32(255.0) * linear_to_srgb(v))
    return int(min(max(np.floor(np.float64(t) + 0.5), 0), 255))  # Float::round: half away from zero (t >= 0 here)


def test_srgb_encoding_matches_the_standard_and_round_trips(lib, tmp_path):
    lut = tables()["SRGB_TO_LINEAR_LUT"]
    x = np.arange(256) / 255.0
    assert np.allclose(lut, np.where(x <= 0.04045, x / 12.92, ((x + 0.055) / 1.055) ** 2.4), atol=3e-7)  # IEC 61966-2-1
    # every byte survives decode -> encode (what select_channels / the pyramid's level 0 do to an 8-bit image)
    assert [linear_to_srgb8(lut[b]) for b in range(256)] == list(range(256))
    ramp = np.arange(256).reshape(16, 16, 1)
    f = tmp_path / "ramp.png"
    write_png(f, np.repeat(ramp, 3, axis=2), 8, 2)
    levels, _ = load_png(lib, f, "sRGB", pyramid=True)
    assert np.array_equal(levels[0][..., 0], lut[ramp[..., 0]]) and len(levels) == 5 and levels[-1].shape == (1, 1, 3)
    levels_lin, _ = load_png(lib, f, "linear", pyramid=True)
    assert np.array_equal(levels_lin[0][..., 1], ramp[..., 0].astype(np.float32) / f32(255.0))
    g, _ = load_png(lib, f, "gamma 2.2")
    assert np.allclose(g[0][..., 2], (ramp[..., 0] / 255.0) ** 2.2, rtol=2e-6, atol=1e-9)
    im = abi.ShmLoadedImage()
    assert lib.shm_image_load_png(str(f).encode(), b"gamma", abi.SHM_WRAP_REPEAT, 0, C.byref(im)) == -1 and "gamma <value>" in lib.shm_last_error().decode()


def pyramid_restated(img8, lut, encode):
    """Image::generate_pyramid for an 8-bit image whose sides are powers of two: f32 box filtering down the chain, every STORED level
    re-encoded to 8 bits and decoded through the table (the level's pixel format is the source's, image.rs:730, 773-778)."""
    cur = lut[img8].astype(np.float32)
    levels = []
    while True:
        enc = np.vectorize(encode)(cur)
        levels.append(lut[enc].astype(np.float32))
        h, w = cur.shape[:2]
        if h == 1 and w == 1:
            return levels
        nh, nw = max(1, (h + 1) // 2), max(1, (w + 1) // 2)
        y0, x0 = 2 * np.arange(nh), 2 * np.arange(nw)
        y1, x1 = (y0 + 1 if h > 1 else y0), (x0 + 1 if w > 1 else x0)
        a, b, c, d = cur[y0][:, x0], cur[y0][:, x1], cur[y1][:, x0], cur[y1][:, x1]
        cur = (f32(0.25) * (((a + b).astype(np.float32) + c).astype(np.float32) + d).astype(np.float32)).astype(np.float32)


@pytest.mark.parametrize("w,h,ctype", [(16, 8, 2), (8, 32, 0), (4, 4, 6)])
def test_pyramid_levels_are_requantised(lib, tmp_path, w, h, ctype):
    nch = {0: 1, 2: 3, 6: 4}[ctype]
    rng = np.random.default_rng(w * h)
    arr = rng.integers(0, 256, size=(h, w, nch))
    if ctype == 6:
        arr[..., 3] = 255  # an all-ones alpha: MIPMap::create_from_file keeps R, G, B (mipmap.rs:57-75)
    f = tmp_path / "p.png"
    write_png(f, arr, 8, ctype, filters=(4,))
    levels, info = load_png(lib, f, "sRGB", wrap=abi.SHM_WRAP_CLAMP, pyramid=True)
    keep = 3 if nch >= 3 else 1
    want = pyramid_restated(arr[..., :keep], tables()["SRGB_TO_LINEAR_LUT"], linear_to_srgb8)
    assert info["file_channels"] == (3 if ctype == 6 else nch) and len(levels) == len(want) == 1 + int(np.log2(max(w, h)))
    for got, exp in zip(levels, want):
        assert got.shape == exp.shape and np.array_equal(got, exp)
    if ctype == 6:  # a real alpha channel stays in the pyramid (file_channels 4); R, G, B are what texel_rgb reads
        arr[0, 0, 3] = 7
        write_png(f, arr, 8, ctype)
        _, info = load_png(lib, f, "sRGB", pyramid=True)
        assert info["file_channels"] == 4 and info["n_channels"] == 3


def test_resize_to_power_of_two_as_the_reference_computes_it(lib, tmp_path):
    """float_resize_up (image.rs:1007-1111) with resample_weights as written there (image.rs:1113-1138): the four taps of an output pixel all
    evaluate the window at first_pixel + 0.5, so after normalisation each weight is 1/4 (to an ulp) — a 4-tap box over source pixels
    first_pixel .. first_pixel + 3, wrapped by the texture's wrap mode, clamped at zero. Restated with that closed form."""
    w, h = 12, 6
    rng = np.random.default_rng(5)
    arr = rng.integers(0, 256, size=(h, w, 3))
    f = tmp_path / "np2.png"
    write_png(f, arr, 8, 2)
    src = arr.astype(np.float64) / 255.0
    for wrap, idx in ((abi.SHM_WRAP_REPEAT, lambda i, n: i % n), (abi.SHM_WRAP_CLAMP, lambda i, n: min(max(i, 0), n - 1))):
        levels, _ = load_png(lib, f, "linear", wrap=wrap, pyramid=True)
        assert levels[0].shape == (8, 16, 3) and len(levels) == 5
        nw, nh = 16, 8
        fx = [max(int(np.floor((i + 0.5) * w / nw - 2.0 + 0.5)), 0) for i in range(nw)]
        fy = [max(int(np.floor((i + 0.5) * h / nh - 2.0 + 0.5)), 0) for i in range(nh)]
        want = np.zeros((nh, nw, 3))
        for y in range(nh):
            for x in range(nw):
                acc = np.zeros(3)
                for dy in range(4):
                    for dx in range(4):
                        acc += src[idx(fy[y] + dy, h), idx(fx[x] + dx, w)]
                want[y, x] = acc / 16.0
        # level 0 is stored in the source's format: 8 bits, linear encoding (value * 255 + 0.5, truncated)
        assert np.abs(levels[0] - np.floor(want * 255.0 + 0.5) / 255.0).max() <= 1.0 / 255.0 + 1e-6
        assert np.abs(levels[0] - want).max() <= 0.5 / 255.0 + 1e-5
    im = abi.ShmLoadedImage()
    # "black" cannot be resized (image.rs:835 asserts on it), and BOTH sides must grow (image.rs:1009-1010)
    assert lib.shm_image_load_png(str(f).encode(), b"linear", abi.SHM_WRAP_BLACK, 1, C.byref(im)) == -1 and "black" in lib.shm_last_error().decode()
    write_png(f, rng.integers(0, 256, size=(8, 12, 3)), 8, 2)
    assert lib.shm_image_load_png(str(f).encode(), b"linear", abi.SHM_WRAP_REPEAT, 1, C.byref(im)) == -1 and "BOTH sides" in lib.shm_last_error().decode()
