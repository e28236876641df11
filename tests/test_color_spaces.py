"""The reference's colour-space vectors (color.rs:1007-1340, 14 tests; VERDICT r02 missing #2 / #3) replayed against the product: the three
named colour spaces (colorspace.rs:117-164) through the scene front end — ColorSpace directive, "rgb" parameters converted with THAT
space's generated rgb2spec table and illuminant, the film's output matrix — checked with an independent numpy integration against the CIE
tables. The reference's random triples come from rand's StdRng, whose stream cannot be replayed here; the same counts and ranges are drawn
from numpy's generator instead (the properties are distribution-free: every triple must round-trip within the reference's epsilon)."""
import ctypes as C

import numpy as np
import pytest

from shimmer_amd import abi, scene as scn

T = scn.tables()
LAM = np.arange(360, 831, dtype=np.float64)
SPACES = {  # name in a scene file, primaries, illuminant table, (lo, span) of the reference's round-trip triples (color.rs:1157, 1191, 1225)
    "srgb": ("CIE_ILLUM_D6500", 0.0, 1.0),
    "rec2020": ("CIE_ILLUM_D6500", 0.1, 0.7),
    "aces2065-1": ("ACES_ILLUM_D60", 0.3, 0.4),
}


def illuminant(key):
    """Spectrum::get_named_spectrum(...): PiecewiseLinearSpectrum::from_interleaved(table, normalize = true), at 1 nm."""
    l, v = T[key][0::2].astype(np.float64), T[key][1::2].astype(np.float64)
    d = np.interp(LAM, l, v)
    return d * (float(T["CIE_Y_INTEGRAL"]) / (d * T["CIE_Y"]).sum())


def xyz_from_spectrum(dense):
    """XYZ::from_spectrum (color.rs:222-232): inner products with the matching functions over 360..=830, / CIE_Y_INTEGRAL."""
    return np.array([(T[k].astype(np.float64) * dense).sum() for k in ("CIE_X", "CIE_Y", "CIE_Z")]) / float(T["CIE_Y_INTEGRAL"])


def load(lib, text):
    out = C.POINTER(abi.ShmPbrtScene)()
    rc = lib.shm_scene_parse_pbrt(text.encode(), None, C.byref(out))
    assert rc == 0, lib.shm_last_error().decode()
    return out


def rgb_from_xyz(lib, cs):
    """RgbFilm::new's output matrix for the cie1931 sensor without white balance IS the film colour space's rgb_from_xyz (film.rs:524)."""
    out = load(lib, f'ColorSpace "{cs}"\nFilm "rgb"\nWorldBegin\nShape "sphere"')
    m = np.array(list(out.contents.output_rgb_from_sensor_rgb), np.float64).reshape(3, 3)
    lib.shm_pbrt_free(out)
    return m


def sigmoid(c, lam):
    """RgbSigmoidPolynomial::get (color.rs:353-383)."""
    x = (c[0] * lam + c[1]) * lam + c[2]
    return np.where(np.isinf(x), (x > 0).astype(np.float64), 0.5 + x / (2.0 * np.sqrt(1.0 + x * x)))


def test_srgb_matrix_columns(lib):
    """color.rs:1030-1049 `srgb`: rgb_from_xyz applied to the x, y, z basis vectors (epsilon 0.001)."""
    m = rgb_from_xyz(lib, "srgb")
    want = np.array([[3.2406, -1.5372, -0.4986], [-0.9689, 1.8758, 0.0415], [0.0557, -0.2040, 1.0570]])
    assert np.abs(m - want).max() < 1e-3


@pytest.mark.parametrize("cs", list(SPACES))
def test_rgb_xyz_round_trip_and_illuminant_white(lib, cs):
    """color.rs:1014-1028 `rgb_xyz` (to_rgb(to_xyz((1, 1, 1))) = (1, 1, 1)) and :1051-1088 `std_illum_whites_*` (the colour space's own
    illuminant is its white: every channel within 0.99..1.01)."""
    m = rgb_from_xyz(lib, cs)
    assert np.allclose(m @ (np.linalg.inv(m) @ np.ones(3)), 1.0, atol=1e-6)
    white = m @ xyz_from_spectrum(illuminant(SPACES[cs][0]))
    assert (white > 0.99).all() and (white < 1.01).all(), white
    # ... and the two spaces that share D65 are still different spaces; ACES's white is D60, not D65
    if cs != "srgb":
        assert not np.allclose(m, rgb_from_xyz(lib, "srgb"), atol=1e-2)
    if cs == "aces2065-1":
        d65 = m @ xyz_from_spectrum(illuminant("CIE_ILLUM_D6500"))
        assert np.abs(d65 - 1.0).max() > 0.02


def _materials(lib, cs, rgbs, directive='Material "diffuse" "rgb reflectance"'):
    body = "\n".join(f'{directive} [ {r:.9g} {g:.9g} {b:.9g} ]\nShape "sphere"' for r, g, b in rgbs)
    return load(lib, f'ColorSpace "{cs}"\nWorldBegin\n{body}')


@pytest.mark.parametrize("cs", list(SPACES))
def test_rgb_albedo_spectrum_round_trip(lib, cs):
    """color.rs:1150-1248 rgb_albedo_spectrum_round_trip_{rgb, rec2020, aces}: RgbAlbedoSpectrum(cs, rgb) x cs.illuminant -> XYZ -> cs.to_rgb
    gives rgb back within 0.01, 100 triples per space in the reference's ranges. Also :1122-1148 rgb_albedo_spectrum_max_value: the
    polynomial's analytic maximum (RgbSigmoidPolynomial::max_value, color.rs:366-383) equals the maximum over a 1/16 nm sweep."""
    key, lo, span = SPACES[cs]
    rng = np.random.default_rng(0)
    rgbs = (lo + span * rng.random((100, 3))).astype(np.float32)
    out = _materials(lib, cs, rgbs)
    d = out.contents.desc
    m, illum = rgb_from_xyz(lib, cs), illuminant(key)
    worst = 0.0
    fine = np.arange(360.0, 830.0 + 1e-9, 1.0 / 16.0)
    for i, rgb in enumerate(rgbs):
        s = d.materials[i + 1].a   # (material 0 is the default slot)
        assert s.kind == abi.SHM_SPECTRUM_RGB_ALBEDO
        c = np.array(list(s.rgb_c), np.float64)
        back = m @ xyz_from_spectrum(sigmoid(c, LAM) * illum)
        worst = max(worst, np.abs(back - rgb).max())
        # max_value: the larger of the end points and the vertex -c1 / (2 c0) when it lies inside (color.rs:366-383)
        cand = [360.0, 830.0] + ([-c[1] / (2 * c[0])] if c[0] != 0 and 360.0 <= -c[1] / (2 * c[0]) <= 830.0 else [])
        analytic = max(float(sigmoid(c, np.array([x]))[0]) for x in cand)
        swept = sigmoid(c, fine).max()
        assert abs(swept - analytic) / swept < 1e-4
    lib.shm_pbrt_free(out)
    assert worst < 0.01, worst


@pytest.mark.parametrize("cs", list(SPACES))
def test_rgb_illuminant_spectrum_round_trip(lib, cs):
    """color.rs:1250-1340 rgb_illum_spectrum_round_trip_{rgb, rec2020, aces}: RgbIlluminantSpectrum(cs, rgb) — scale x sigmoid x the colour
    space's illuminant — integrates back to rgb within 0.01. Read from the scene: an area light's "rgb L" arrives as that spectrum sampled at
    1 nm (DiffuseAreaLight::new, light.rs:525-527)."""
    key, lo, span = SPACES[cs]
    rng = np.random.default_rng(1)
    rgbs = (lo + span * rng.random((100, 3))).astype(np.float32)
    body = "\n".join(f'AttributeBegin\nAreaLightSource "diffuse" "rgb L" [ {r:.9g} {g:.9g} {b:.9g} ]\nShape "sphere"\nAttributeEnd' for r, g, b in rgbs)
    out = load(lib, f'ColorSpace "{cs}"\nWorldBegin\n{body}')
    d = out.contents.desc
    assert d.n_lights == 100
    m = rgb_from_xyz(lib, cs)
    worst = 0.0
    for i, rgb in enumerate(rgbs):
        l = d.lights[i]
        assert l.kind == abi.SHM_LIGHT_DIFFUSE_AREA and l.spectrum.kind == abi.SHM_SPECTRUM_DENSE
        dense = np.array([d.spectrum_data[l.spectrum.offset + k] for k in range(471)], np.float64)
        worst = max(worst, np.abs(m @ xyz_from_spectrum(dense) - rgb).max())
    lib.shm_pbrt_free(out)
    assert worst < 0.01, worst


@pytest.mark.parametrize("cs", list(SPACES))
def test_rgb_unbounded_spectrum_max_value(lib, cs):
    """color.rs:1090-1120 rgb_unbounded_spectrum_max_value: rgb in [0, 10)^3; scale x max of the polynomial == the maximum over a 1/16 nm
    sweep (relative 1e-4). A conductor's "rgb eta" is read as RgbUnboundedSpectrum (paramdict.rs:605-656)."""
    rng = np.random.default_rng(2)
    rgbs = (10.0 * rng.random((100, 3))).astype(np.float32)
    out = _materials(lib, cs, rgbs, 'Material "conductor" "rgb k" [ 1 1 1 ] "rgb eta"')
    d = out.contents.desc
    fine = np.arange(360.0, 830.0 + 1e-9, 1.0 / 16.0)
    for i, rgb in enumerate(rgbs):
        s = d.materials[i + 1].a
        assert s.kind == abi.SHM_SPECTRUM_RGB_UNBOUNDED and s.c == pytest.approx(2.0 * rgb.max(), rel=1e-6)
        c = np.array(list(s.rgb_c), np.float64)
        cand = [360.0, 830.0] + ([-c[1] / (2 * c[0])] if c[0] != 0 and 360.0 <= -c[1] / (2 * c[0]) <= 830.0 else [])
        analytic = s.c * max(float(sigmoid(c, np.array([x]))[0]) for x in cand)
        swept = (s.c * sigmoid(c, fine)).max()
        assert abs(swept - analytic) / swept < 1e-4
    lib.shm_pbrt_free(out)


def test_color_space_is_attribute_state_and_travels_with_the_parameter(lib):
    """scene.rs:1561-1564 (the directive sets the graphics state's colour space: saved and restored by AttributeBegin / End), :1725-1729 (an
    Attribute's parameters keep the colour space of their declaration), colorspace.rs:123-132 (names match in any case; others panic)."""
    text = '''
WorldBegin
Material "diffuse" "rgb reflectance" [ 0.7 0.2 0.1 ]
Shape "sphere"
AttributeBegin
  ColorSpace "Rec2020"
  Material "diffuse" "rgb reflectance" [ 0.7 0.2 0.1 ]
  Shape "sphere"
  Attribute "material" "rgb reflectance" [ 0.7 0.2 0.1 ]
  ColorSpace "aces2065-1"
  Material "diffuse"
  Shape "sphere"
  Material "diffuse" "rgb reflectance" [ 0.7 0.2 0.1 ]
  Shape "sphere"
AttributeEnd
Material "diffuse" "rgb reflectance" [ 0.7 0.2 0.1 ]
Shape "sphere"
'''
    out = load(lib, text)
    d = out.contents.desc
    c = [tuple(d.materials[i].a.rgb_c) for i in range(1, 6)]
    lib.shm_pbrt_free(out)
    srgb, rec, attr_rec, aces, srgb_again = c
    assert srgb == srgb_again and rec == attr_rec           # restored by AttributeEnd; the attribute kept rec2020 although read under aces
    assert len({srgb, rec, aces}) == 3                      # the same numbers are three different spectra in the three spaces
    bad = C.POINTER(abi.ShmPbrtScene)()
    assert lib.shm_scene_parse_pbrt(b'ColorSpace "prophoto"\nWorldBegin\nShape "sphere"', None, C.byref(bad)) == -1
    assert "<string>:1: Unknown color space: prophoto" in lib.shm_last_error().decode()


def test_default_light_spectrum_is_the_colour_spaces_illuminant(lib):
    """light.rs:141-145, 433, 592-596: a light without "L" / "I" emits its parameter dictionary's colour-space illuminant — D65, or ACES D60."""
    def sky(cs):
        out = load(lib, f'WorldBegin\nColorSpace "{cs}"\nLightSource "infinite"\nShape "sphere"')
        d = out.contents.desc
        dense = np.array([d.spectrum_data[d.lights[0].spectrum.offset + k] for k in range(471)], np.float64)
        lib.shm_pbrt_free(out)
        return dense
    assert np.allclose(sky("srgb"), illuminant("CIE_ILLUM_D6500"), rtol=2e-5) and np.allclose(sky("rec2020"), sky("srgb"))
    assert np.allclose(sky("aces2065-1"), illuminant("ACES_ILLUM_D60"), rtol=2e-5) and not np.allclose(sky("aces2065-1"), sky("srgb"), rtol=1e-2)


def test_from_xy_zero_and_table_files(lib):
    """color.rs:1007-1012 `from_xy_zero`: XYZ::from_xy_y((1, 0), 0.5) is (0, 0, 0), not a division by zero. The product reaches from_xy_y only
    through the film's white balance (color.rs:404-416), where y is a chromaticity of a real illuminant and never 0: no counterpart needed.
    What is checked instead is that each colour space found its own generated table (rgb_to_spectra.rs:27-45: three files)."""
    from pathlib import Path
    data = Path(abi.__file__).resolve().parent / "data"
    for name in ("srgb", "rec2020", "aces2065_1"):
        f = data / f"rgb2spec_{name}_res64.spec"
        assert f.exists() and f.stat().st_size == 4 + 4 + 64 * 4 + 3 * 64 ** 3 * 3 * 4 and f.read_bytes()[:4] == b"SPEC"
