"""The reference's own in-source vectors for its PBRT-v4 front end, replayed against the product's front end (VERDICT r02 missing #3):
loading/tokenizer.rs (13 tests), loading/token.rs (6), loading/param.rs (5), loading/parser.rs (12). The vectors are data in
tests/golden/reference_loader_vectors.json (each with file:line); the code under test is shimmer_amd/csrc/host/pbrt_loader.cpp through
the C ABI: shm_pbrt_tokenize / shm_pbrt_parse_params (its tokenizer and parameter-list parser) and shm_scene_parse_pbrt (directives)."""
import ctypes as C
import json
from pathlib import Path

import numpy as np
import pytest

from shimmer_amd import abi

V = json.loads((Path(__file__).resolve().parent / "golden" / "reference_loader_vectors.json").read_text())


def tokenize(lib, text):
    buf = C.create_string_buffer(1 << 16)
    n = C.c_uint32()
    abi.check(lib, lib.shm_pbrt_tokenize(text.encode(), buf, len(buf), C.byref(n)), "shm_pbrt_tokenize")
    raw = buf.raw
    out, pos = [], 0
    for _ in range(n.value):
        end = raw.index(b"\0", pos)
        out.append((chr(raw[pos]), raw[pos + 1:end].decode()))
        pos = end + 1
    return out


def parse_params(lib, text):
    buf = C.create_string_buffer(1 << 16)
    rc = lib.shm_pbrt_parse_params(text.encode(), buf, len(buf))
    if rc != 0:
        raise abi.ShimmerHipError(f"{rc}: {lib.shm_last_error().decode()}")
    return json.loads(buf.value.decode())


def load(lib, text):
    out = C.POINTER(abi.ShmPbrtScene)()
    rc = lib.shm_scene_parse_pbrt(text.encode(), None, C.byref(out))
    return rc, out, lib.shm_last_error().decode()


@pytest.mark.parametrize("case", V["tokenizer"], ids=lambda c: c["source"].split()[-1])
def test_tokenizer_vectors(lib, case):
    assert [t for _, t in tokenize(lib, case["input"])] == case["tokens"]


def test_token_classification_vectors(lib):
    kinds = lambda text: tokenize(lib, text)
    q = V["token"]["is_quote"]
    for s in q["true"]:            # Token::is_quote / unquote: Some(inner)
        (k, raw), = kinds(s)
        assert k == "S" and raw == s and raw[1:-1] == q["unquoted"][s]
    for s in q["false"]:           # not a quoted string: nothing, an unterminated quote, or a word followed by one
        toks = kinds(s)
        assert not (len(toks) == 1 and toks[0][0] == "S")
    v = V["token"]["is_valid"]
    for s in v["true"]:            # one well-formed token
        toks = kinds(s)
        assert len(toks) == 1 and toks[0][0] in "SWD" and toks[0][1] == s
    for s in v["false"]:           # empty, an open quote, or more than one token
        toks = kinds(s)
        assert len(toks) != 1 or toks[0][0] == "Q"
    d = V["token"]["directive"]
    for s in d["true"]:
        assert kinds(s) == [("D", s)]
    for s in d["false"]:
        assert all(k != "D" for k, _ in kinds(s))
    # Token::parse::<u32>: "32" is the number, "" and "-" are not numbers (here: integer parameter values)
    assert parse_params(lib, '"integer n" 32')[0]["ints"] == [32]
    for bad in V["token"]["parse"]["err"]:
        with pytest.raises(abi.ShimmerHipError):
            parse_params(lib, f'"integer n" [ {bad} ]' if bad else '"integer n"')
    # the loader proper rejects what the reference's parser rejects at Token::is_valid: an opening quote without its partner
    rc, out, err = load(lib, 'WorldBegin\nShape "sphere')
    assert rc == -1 and not out and "<string>:2: unterminated string" in err


def test_param_vectors(lib):
    p = V["param"]
    for ty in p[0]["types_ok"]:
        vals = {"bool": "true", "rgb": "0.1 0.2 0.3", "point2": "0 1", "point3": "0 1 2"}.get(ty, "1")
        assert parse_params(lib, f'"{ty} x" [ {vals} ]')[0]["type"] == ty
    with pytest.raises(abi.ShimmerHipError, match="unknown parameter type"):
        parse_params(lib, '"colour x" 1')
    with pytest.raises(abi.ShimmerHipError, match="duplicated parameter name"):   # ParamList::add -> Err(DuplicatedParamName)
        parse_params(lib, p[1]["duplicate"])
    for case in p[2:]:
        got, = parse_params(lib, f'"{case["decl"]}" [ {case["value"]} ]')
        assert got["type"], got["name"] == case["decl"].split()
        assert got["ints"] == case.get("ints", []) and got["floats"] == case.get("floats", [])
    # ... and through the scene: "blackbody I" 5500 on a point light, "rgb L" [7 0 7] on an area light (Spectrum::Blackbody / Spectrum::Rgb)
    rc, out, err = load(lib, 'WorldBegin\nLightSource "point" "blackbody I" 5500\nAttributeBegin\nAreaLightSource "diffuse" "rgb L" [ 7 0 7 ]\nShape "sphere"\nAttributeEnd')
    assert rc == 0, err
    d = out.contents.desc
    # both arrive as DenselySampledSpectrum (light.rs: PointLight::create / DiffuseAreaLight::new sample their spectrum at 1 nm)
    lights = {d.lights[i].kind: d.lights[i] for i in range(d.n_lights)}
    dense = lambda l: np.array([d.spectrum_data[l.spectrum.offset + k] for k in range(471)])
    bb, rgb = dense(lights[abi.SHM_LIGHT_POINT]), dense(lights[abi.SHM_LIGHT_DIFFUSE_AREA])
    assert all(l.spectrum.kind == abi.SHM_SPECTRUM_DENSE for l in lights.values())
    assert np.argmax(bb) == round(2.8977721e-3 / 5500 * 1e9) - 360               # Wien's peak of a 5500 K blackbody, 527 nm
    assert rgb[550 - 360] < 0.2 * rgb[450 - 360] and rgb[550 - 360] < 0.2 * rgb[640 - 360]  # (7, 0, 7): magenta, no green
    lib.shm_pbrt_free(out)


def test_parser_param_to_parsed_param_vectors(lib):
    for case in V["parser"]["param_to_parsed_param"]:
        got, = parse_params(lib, f'"{case["decl"]}" [ {case["value"]} ]')
        assert got["name"] == case["name"] and got["type"] == case["type"]
        for field in ("ints", "bools", "strings"):
            assert got[field] == case.get(field, []), (case["source"], field)
        # Float is f32 in the reference (float.rs:1-4): vec![0.0, 10.0, 2.0, 0.1, 2.5, 3.4] are the f32 nearest those decimals
        assert np.array_equal(np.array(got["floats"], np.float32), np.array(case.get("floats", []), np.float32)), case["source"]


def test_parser_option_and_film_vectors(lib):
    o = V["parser"]["parse_option"]
    a, b = parse_params(lib, o["bracketed"]), parse_params(lib, o["bare"])
    assert a == b == [{"type": o["type"], "name": o["name"], "floats": [], "ints": [], "bools": [], "strings": o["strings"]}]
    # the same equivalence through a directive the loader acts on
    for text in ('Option "string rendercoordsys" [ "world" ]', 'Option "string rendercoordsys" "world"'):
        rc, out, err = load(lib, text + '\nWorldBegin\nShape "sphere"')
        assert rc == 0, err
        lib.shm_pbrt_free(out)
    f = V["parser"]["parse_film"]
    directive = f["directive"]
    params = parse_params(lib, directive[len('Film "rgb"'):])
    assert len(params) == f["n_params"]
    by = {p["name"]: p for p in params}
    assert by["filename"]["type"] == "string" and by["filename"]["strings"] == [f["filename"]]
    assert by["iso"]["type"] == "float" and by["iso"]["floats"] == [f["iso"]]
    assert by["yresolution"]["ints"] == [f["yresolution"]] and by["xresolution"]["ints"] == [f["xresolution"]]
    assert by["sensor"]["strings"] == [f["sensor"]]
    # the directive itself: the named sensor needs <name>_r/_g/_b spectra the reference's NamedSpectrum does not hold (its panic, here an
    # error that names it); without that parameter the film is what the vector says
    rc, out, err = load(lib, directive + '\nWorldBegin\nShape "sphere"')
    assert rc != 0 and f["sensor"] in err
    rc, out, err = load(lib, directive.replace('"string sensor" "canon_eos_5d_mkiv"', "") + '\nWorldBegin\nShape "sphere"')
    assert rc == 0, err
    assert list(out.contents.desc.film.full_resolution) == [f["xresolution"], f["yresolution"]] and out.contents.output_filename == f["filename"].encode()
    lib.shm_pbrt_free(out)
    # Film without parameters followed directly by another directive (parse_film_no_params)
    rc, out, err = load(lib, V["parser"]["parse_film_no_params"]["input"] + '\nWorldBegin\nShape "sphere"')
    assert rc == 0, err
    assert list(out.contents.desc.film.full_resolution) == [1280, 720]
    lib.shm_pbrt_free(out)


def _sphere_matrix(lib, body):
    rc, out, err = load(lib, "WorldBegin\n" + body + '\nShape "sphere"')
    assert rc == 0, err
    m = np.array(list(out.contents.desc.spheres[0].render_from_object), np.float64).reshape(4, 4)
    lib.shm_pbrt_free(out)
    return m


def test_parser_transform_directive_vectors(lib):
    p = V["parser"]
    # Scale -1 1 1 then Rotate 1 0 0 1: both consume exactly their numbers; ctm = S * R(1 degree about z)
    m = _sphere_matrix(lib, p["parse_scale_and_rotate"]["input"])
    t = np.radians(1.0)
    want = np.diag([-1.0, 1.0, 1.0, 1.0]) @ np.array([[np.cos(t), -np.sin(t), 0, 0], [np.sin(t), np.cos(t), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    assert np.allclose(m, want, atol=1e-6)
    # Transform / ConcatTransform [16 numbers, column-major as PBRT files give them]: the translation (3, 1, -4)
    for key in ("parse_transform", "parse_concat_transform"):
        m = _sphere_matrix(lib, p[key]["input"])
        assert np.allclose(m[:3, 3], p[key]["translation"]) and np.allclose(m[:3, :3], np.eye(3))
    # LookAt with numbers spread over three lines: the camera looks from `eye` towards `look`
    la = p["parse_look_at"]
    rc, out, err = load(lib, la["input"] + '\nCamera "perspective"\nWorldBegin\nShape "sphere"')  # (the Camera directive captures the CTM)
    assert rc == 0, err
    want = np.zeros(16, np.float32)
    f3 = lambda v: np.array(v, np.float32).ctypes.data_as(abi.c_float_p)
    abi.check(lib, lib.shm_look_at(f3(la["eye"]), f3(la["look"]), f3(la["up"]), want.ctypes.data_as(abi.c_float_p)), "shm_look_at")
    got = np.array(list(out.contents.desc.camera.render_from_camera), np.float32).reshape(4, 4)
    assert np.allclose(got[:3, :3], want.reshape(4, 4)[:3, :3], atol=1e-6)   # (camera-world render space: the translation lives in render_from_world)
    lib.shm_pbrt_free(out)


def test_parser_include_and_import_vectors(lib, tmp_path):
    inc = V["parser"]["parse_includes"]
    rc, out, err = load(lib, "WorldBegin\n" + inc["input"])
    assert rc == -1 and f'unable to read included file' in err and inc["include"] in err and "<string>:3" in err
    rc, out, err = load(lib, "WorldBegin\n" + inc["input"].replace('Include "geometry/car.pbrt"', ""))
    assert rc == -2 and inc["import"] in err   # Import is todo!() in the reference: unsupported, named
    # a bracket left open runs into the next directive: UnexpectedToken in the reference (parser.rs:603-606)
    rc, out, err = load(lib, 'WorldBegin\nShape "trianglemesh" "integer indices" [ 0 1 2\nShape "sphere"')
    assert rc == -1 and "unexpected directive Shape" in err
