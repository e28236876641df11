"""The C++ PBRT-v4 front end (shm_scene_parse_pbrt / shm_scene_load_pbrt; reference: loading/parser.rs:216-351, loading/scene.rs:1221-2033)
against the repository's Python scene generators: the same scene written as .pbrt text must come back as the SAME ShmSceneDesc — BVH
nodes, leaf-ordered primitives, meshes, spheres, materials, lights, spectra, camera and film byte for byte — and therefore render to the
same film (sha256 of the oracle's f64 sums). Plus the parser's own behaviour: defaults of the reference, named materials, textures,
object instances, transforms, Include, and errors with file:line instead of panics."""
import ctypes as C
import hashlib

import numpy as np
import pytest

import oracle_py
from shimmer_amd import abi, render, scene as scn, scenes


def load(lib, text, base_dir=None):
    out = C.POINTER(abi.ShmPbrtScene)()
    rc = lib.shm_scene_parse_pbrt(text.encode(), base_dir.encode() if base_dir else None, C.byref(out))
    if rc != 0:
        raise abi.ShimmerHipError(f"{rc}: {lib.shm_last_error().decode()}")
    return out


def arr(ptr, n, ctype=None):
    if n == 0 or not ptr:
        return b""
    size = C.sizeof(ptr._type_) * n
    return C.string_at(ptr, size)


def mesh_bytes(desc):
    out = []
    for i in range(desc.n_meshes):
        m = desc.meshes[i]
        out.append((m.n_triangles, m.n_vertices, arr(m.vertex_indices, 3 * m.n_triangles), arr(m.p, 3 * m.n_vertices), arr(m.n, 3 * m.n_vertices) if m.n else b"",
                    arr(m.s, 3 * m.n_vertices) if m.s else b"", arr(m.uv, 2 * m.n_vertices) if m.uv else b"", m.reverse_orientation, m.transform_swaps_handedness))
    return out


def spectrum_content(desc, sp):
    """A ShmSpectrum resolved to what it evaluates: kind + parameters + the table floats it points at (pool offsets are layout, not content)."""
    data = np.ctypeslib.as_array(desc.spectrum_data, shape=(max(1, desc.n_spectrum_floats),))
    n = {abi.SHM_SPECTRUM_DENSE: sp.n, abi.SHM_SPECTRUM_PIECEWISE_LINEAR: 2 * sp.n}.get(sp.kind, 0)
    return (sp.kind, np.float32(sp.c).tobytes(), sp.n if n else 0, sp.lambda_min if sp.kind == abi.SHM_SPECTRUM_DENSE else 0, data[sp.offset:sp.offset + n].tobytes())


def assert_same_scene(a, b, exact_pool=True):
    assert a.n_nodes == b.n_nodes and arr(a.nodes, a.n_nodes) == arr(b.nodes, b.n_nodes)
    assert a.n_primitives == b.n_primitives and arr(a.primitives, a.n_primitives) == arr(b.primitives, b.n_primitives)
    assert mesh_bytes(a) == mesh_bytes(b)
    assert a.n_spheres == b.n_spheres and arr(a.spheres, a.n_spheres) == arr(b.spheres, b.n_spheres)
    assert bytes(a.camera) == bytes(b.camera)
    fa, fb = a.film, b.film
    assert list(fa.pixel_bounds) == list(fb.pixel_bounds) and list(fa.full_resolution) == list(fb.full_resolution) and list(fa.filter_radius) == list(fb.filter_radius)
    assert fa.imaging_ratio == fb.imaging_ratio and fa.max_component_value == fb.max_component_value
    for t in ("sensor_r_bar", "sensor_g_bar", "sensor_b_bar"):
        assert arr(getattr(fa, t), 471) == arr(getattr(fb, t), 471)
    assert a.n_materials == b.n_materials
    for i in range(a.n_materials):
        ma, mb = a.materials[i], b.materials[i]
        for f, _ in abi.ShmMaterial._fields_:
            va, vb = getattr(ma, f), getattr(mb, f)
            if isinstance(va, abi.ShmSpectrum):
                assert spectrum_content(a, va) == spectrum_content(b, vb), (i, f)
            elif hasattr(va, "__len__"):
                assert list(va) == list(vb), (i, f)
            else:
                assert va == vb or (va != va and vb != vb), (i, f)
    assert a.n_lights == b.n_lights
    for i in range(a.n_lights):
        la, lb = a.lights[i], b.lights[i]
        assert (la.kind, la.primitive, la.two_sided, list(la.position)) == (lb.kind, lb.primitive, lb.two_sided, list(lb.position))
        assert np.float32(la.scale).tobytes() == np.float32(lb.scale).tobytes() and abs(la.area - lb.area) <= 1e-6 * abs(lb.area)
        assert spectrum_content(a, la.spectrum) == spectrum_content(b, lb.spectrum)
    if exact_pool:
        assert arr(a.spectrum_data, a.n_spectrum_floats) == arr(b.spectrum_data, b.n_spectrum_floats)


def film_sha(desc, params):
    o = oracle_py.Oracle(desc)
    try:
        film, _ = o.render(params, n_threads=8)
    finally:
        o.close()
    return hashlib.sha256(np.ascontiguousarray(film).tobytes()).hexdigest()


S1_TEXT = """
# S1 (BASELINE configs[0]): unit sphere, one-sided quad emitter, floor — shimmer_amd/scenes.py sphere_light
LookAt 0 1 5   0 0 0   0 1 0
Camera "perspective" "float fov" [ 40 ]
Film "rgb" "integer xresolution" [ 48 ] "integer yresolution" 40 "string filename" "s1.pfm"
Sampler "independent" "integer pixelsamples" 4
Integrator "path" "integer maxdepth" [ 5 ]
WorldBegin
MakeNamedMaterial "grey" "string type" "diffuse" "float reflectance" 0.5
MakeNamedMaterial "black" "string type" "diffuse" "float reflectance" 0
NamedMaterial "grey"
Shape "sphere" "float radius" 1
AttributeBegin
  NamedMaterial "black"
  AreaLightSource "diffuse" "blackbody L" [ 6500 ] "float scale" 10
  Shape "trianglemesh" "point3 P" [ -1 3 -1   1 3 -1   1 3 1   -1 3 1 ] "integer indices" [ 0 1 2  0 2 3 ]
AttributeEnd
Shape "trianglemesh" "point3 P" [ -4 -1 -4   -4 -1 4   4 -1 4   4 -1 -4 ] "integer indices" [ 0 1 2 0 2 3 ]
"""


def test_s1_text_equals_generator(lib):
    """The default material slot 0 of the loader (scene.rs:1296-1300 installs "diffuse") precedes the named ones; the generator has none."""
    got = load(lib, S1_TEXT)
    try:
        want = scenes.sphere_light(lib, 48, 40)
        d = got.contents.desc
        # materials: loader = [default diffuse 0.5, grey, black]; generator = [grey, black]: compare through the primitives' materials
        assert d.n_materials == want.desc.n_materials + 1
        prim_g = np.frombuffer(arr(d.primitives, d.n_primitives), np.uint32).reshape(-1, 4).copy()
        prim_w = np.frombuffer(arr(want.desc.primitives, want.desc.n_primitives), np.uint32).reshape(-1, 4)
        prim_g[:, 2] -= 1
        assert np.array_equal(prim_g, prim_w)
        assert arr(d.nodes, d.n_nodes) == arr(want.desc.nodes, want.desc.n_nodes)
        assert mesh_bytes(d) == mesh_bytes(want.desc)
        assert arr(d.spheres, 1) == arr(want.desc.spheres, 1)
        assert bytes(d.camera) == bytes(want.desc.camera)
        assert arr(d.spectrum_data, d.n_spectrum_floats) == arr(want.desc.spectrum_data, want.desc.n_spectrum_floats)  # one pooled blackbody table
        for i in range(d.n_lights):
            assert np.float32(d.lights[i].scale).tobytes() == np.float32(want.desc.lights[i].scale).tobytes()
            assert d.lights[i].primitive == want.desc.lights[i].primitive
        s = got.contents
        assert (s.params.samples_per_pixel, s.params.max_depth, s.integrator, s.output_filename) == (4, 5, b"path", b"s1.pfm")
        p = render.make_params(seed=3, spp=4, max_depth=5)
        assert film_sha(d, p) == film_sha(want.desc, p)  # the same film, bit for bit
    finally:
        lib.shm_pbrt_free(got)


def _pbrt_floats(a):
    return " ".join(repr(float(np.float32(x))) for x in np.asarray(a).ravel())


def cornell_text(record):
    """S2 as .pbrt text from the generator's own recorded calls (world-space vertices, spectra by content)."""
    lines = ["LookAt %s  %s  %s" % tuple(_pbrt_floats(v) for v in record["look_at"]), 'Camera "perspective" "float fov" [ %r ]' % record["fov"],
             'Film "rgb" "integer xresolution" %d "integer yresolution" %d' % record["res"], "WorldBegin"]
    for i, m in enumerate(record["materials"]):
        lines.append('MakeNamedMaterial "m%d" "string type" "diffuse" %s' % (i, m))
    for mesh in record["meshes"]:
        lines.append("AttributeBegin")
        lines.append('  NamedMaterial "m%d"' % mesh["material"])
        if mesh["emission"]:
            lines.append('  AreaLightSource "diffuse" "blackbody L" [ %r ] "float scale" %r' % mesh["emission"])
        lines.append('  Shape "trianglemesh" "point3 P" [ %s ] "integer indices" [ %s ]' % (_pbrt_floats(mesh["p"]), " ".join(str(int(x)) for x in mesh["vi"].ravel())))
        lines.append("AttributeEnd")
    return "\n".join(lines) + "\n"


def test_cornell_text_equals_generator(lib, monkeypatch):
    """S2 (BASELINE configs[1]): record what the generator feeds its builder (world-space vertices before the render-space translation),
    print that as .pbrt text, load it: same BVH, meshes, lights, camera, spectra -> same film."""
    record = {"meshes": [], "materials": []}
    world = []
    orig_to_render = scenes._to_render
    monkeypatch.setattr(scenes, "_to_render", lambda p, rfw: (world.append(np.asarray(p, np.float32)), orig_to_render(p, rfw))[1])
    orig_look = scn.SceneBuilder.set_camera_look_at
    orig_mesh = scn.SceneBuilder.add_mesh
    orig_diffuse = scn.SceneBuilder.material_diffuse

    def look(self, lib_, pos, look_at, up, fov, **kw):
        record["look_at"], record["fov"] = (pos, look_at, up), float(fov)
        return orig_look(self, lib_, pos, look_at, up, fov, **kw)

    def mesh(self, p, vi, material, emission=None, emission_scale=1.0, **kw):
        record["meshes"].append(dict(p=world[-1], vi=np.asarray(vi), material=material, emission=(6500.0, float(emission_scale)) if emission is not None else None))
        return orig_mesh(self, p, vi, material, emission=emission, emission_scale=emission_scale, **kw)

    def diffuse(self, reflectance):
        if isinstance(reflectance, abi.ShmSpectrum):  # the 2-knot piecewise-linear wall colours: written as lambda / value pairs
            k = reflectance
            pool = np.concatenate(self.spec)
            lam, val = pool[k.offset:k.offset + k.n], pool[k.offset + k.n:k.offset + 2 * k.n]
            record["materials"].append('"spectrum reflectance" [ %s ]' % " ".join("%r %r" % (float(a), float(b)) for a, b in zip(lam, val)))
        else:
            record["materials"].append('"float reflectance" %r' % float(reflectance))
        return orig_diffuse(self, reflectance)

    monkeypatch.setattr(scn.SceneBuilder, "set_camera_look_at", look)
    monkeypatch.setattr(scn.SceneBuilder, "add_mesh", mesh)
    monkeypatch.setattr(scn.SceneBuilder, "material_diffuse", diffuse)
    want = scenes.cornell_box(lib, 40, 40)
    record["res"] = (40, 40)
    text = cornell_text(record)
    got = load(lib, text)
    try:
        d = got.contents.desc
        assert d.n_primitives == 32 == want.desc.n_primitives
        prim_g = np.frombuffer(arr(d.primitives, 32), np.uint32).reshape(-1, 4).copy()
        prim_w = np.frombuffer(arr(want.desc.primitives, 32), np.uint32).reshape(-1, 4)
        prim_g[:, 2] -= 1  # the loader's default material occupies slot 0
        assert np.array_equal(prim_g, prim_w)
        assert arr(d.nodes, d.n_nodes) == arr(want.desc.nodes, want.desc.n_nodes)
        assert mesh_bytes(d) == mesh_bytes(want.desc)
        assert bytes(d.camera) == bytes(want.desc.camera)
        for i in range(want.desc.n_materials):
            assert spectrum_content(d, d.materials[i + 1].a) == spectrum_content(want.desc, want.desc.materials[i].a)
        p = render.make_params(seed=1, spp=2, max_depth=5)
        assert film_sha(d, p) == film_sha(want.desc, p)
    finally:
        lib.shm_pbrt_free(got)


def test_defaults_materials_textures_instances(lib, tmp_path):
    """Reference defaults and the rest of the directive set: default camera / film / sampler / integrator (scene.rs:1225-1262: perspective
    fov 90, 1280x720, independent 4 spp, path maxdepth 5), Cu conductor defaults, coated materials, named spectra, float / spectrum texture
    graphs, mix of named materials, transforms, ReverseOrientation, an object instanced twice, a point and an infinite light, Include."""
    (tmp_path / "inc.pbrt").write_text('Shape "sphere" "float radius" 0.25\n')
    text = """
    Film "rgb" "integer xresolution" 32 "integer yresolution" 16 "integer pixelbounds" [ 4 28 2 14 ] "float iso" 200
    Camera "perspective"
    WorldBegin
    LightSource "point" "point3 from" [ 0 4 0 ] "blackbody I" 5000 "float scale" 3
    LightSource "infinite" "float scale" 0.5
    Texture "rough" "float" "mix" "float tex1" 0.1 "float tex2" 0.4 "float amount" 0.25
    Texture "tint" "spectrum" "scale" "spectrum tex" [ 400 0.2 700 0.9 ] "float scale" 0.5
    MakeNamedMaterial "gold" "string type" "conductor" "spectrum eta" "metal-Au-eta" "spectrum k" "metal-Au-k" "texture roughness" "rough"
    MakeNamedMaterial "copper" "string type" "conductor"
    MakeNamedMaterial "glass" "string type" "dielectric" "spectrum eta" "glass-BK7"
    MakeNamedMaterial "paint" "string type" "coateddiffuse" "texture reflectance" "tint" "float thickness" 0.02 "integer maxdepth" 6
    MakeNamedMaterial "both" "string type" "mix" "string materials" [ "gold" "paint" ] "float amount" 0.3
    ObjectBegin "ball"
      NamedMaterial "glass"
      Include "inc.pbrt"
    ObjectEnd
    AttributeBegin
      Translate 0 0 5
      ObjectInstance "ball"
      Translate 1 0 0
      Scale 2 2 2
      ObjectInstance "ball"
    AttributeEnd
    AttributeBegin
      NamedMaterial "both"
      Translate 0 -1 6
      Rotate 90 1 0 0
      ReverseOrientation
      Shape "bilinearmesh" "point3 P" [ -3 -3 0  3 -3 0  -3 3 0  3 3 0 ] "normal N" [ 0 0 1  0 0 1  0 0 1  0 0 1 ]
    AttributeEnd
    NamedMaterial "copper"
    Shape "trianglemesh" "point3 P" [ -1 0 4  1 0 4  0 1 4 ]
    """
    got = load(lib, text, str(tmp_path))
    try:
        s = got.contents
        d = s.desc
        assert (s.params.samples_per_pixel, s.params.max_depth, s.integrator, s.output_filename) == (4, 5, b"path", b"shimmer.pfm")
        assert list(d.film.pixel_bounds) == [4, 2, 28, 14] and list(d.film.full_resolution) == [32, 16] and d.film.imaging_ratio == 2.0
        assert d.camera.kind == abi.SHM_CAMERA_PERSPECTIVE and d.camera.lens_radius == 0.0 and d.camera.focal_distance == 1e6
        kinds = [d.materials[i].kind for i in range(d.n_materials)]
        assert kinds == [abi.SHM_MATERIAL_DIFFUSE, abi.SHM_MATERIAL_CONDUCTOR, abi.SHM_MATERIAL_CONDUCTOR, abi.SHM_MATERIAL_DIELECTRIC, abi.SHM_MATERIAL_COATED_DIFFUSE,
                         abi.SHM_MATERIAL_MIX]
        gold, copper, glass, paint, both = (d.materials[i] for i in range(1, 6))
        assert gold.float_tex[abi.SHM_FLOATSLOT_U_ROUGHNESS] != 0 and gold.float_tex[abi.SHM_FLOATSLOT_V_ROUGHNESS] == gold.float_tex[abi.SHM_FLOATSLOT_U_ROUGHNESS]
        assert gold.remap_roughness == 1 and copper.a.kind == abi.SHM_SPECTRUM_PIECEWISE_LINEAR and copper.u_roughness == 0.0  # defaults: metal-Cu-eta / k
        assert glass.a.kind == abi.SHM_SPECTRUM_PIECEWISE_LINEAR  # dispersive: terminate_secondary on the device
        assert paint.a.kind == abi.SHM_SPECTRUM_TEXTURE_NODE and abs(paint.thickness - 0.02) < 1e-9 and paint.max_depth == 6 and paint.n_samples == 1 and paint.d.c == 1.5
        assert list(both.mix_material) == [1, 4] and abs(both.mix_amount - 0.3) < 1e-7
        assert d.n_float_textures >= 4 and d.n_spectrum_textures == 2 and d.n_instances == 2 and d.n_spheres == 1 and d.n_patch_meshes == 1 and d.n_meshes == 1
        assert d.patch_meshes[0].reverse_orientation == 1
        i0, i1 = d.instances[0], d.instances[1]
        m0, m1 = np.array(list(i0.render_from_primitive)).reshape(4, 4), np.array(list(i1.render_from_primitive)).reshape(4, 4)
        assert np.allclose(m0, np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 5], [0, 0, 0, 1]])) and np.allclose(m1[:3, :3], 2 * np.eye(3)) and np.allclose(m1[:3, 3], [1, 0, 5])
        assert i0.root_node == i1.root_node > 0
        lk = [d.lights[i].kind for i in range(d.n_lights)]
        assert lk == [abi.SHM_LIGHT_POINT, abi.SHM_LIGHT_UNIFORM_INFINITE] and list(d.lights[0].position) == [0.0, 4.0, 0.0]
        # the description is a valid scene: the oracle accepts and renders it
        o = oracle_py.Oracle(d)
        film, st = o.render(render.make_params(seed=0, spp=2, max_depth=4), n_threads=4)
        o.close()
        assert np.isfinite(film["rgb_sum"]).all() and film["rgb_sum"].sum() > 0 and (film["weight_sum"][2:14, 4:28] == 2).all()
    finally:
        lib.shm_pbrt_free(got)


@pytest.mark.parametrize("text,code,needle", [
    ('WorldBegin\nShape "sphere"\nFoo', -1, "<string>:3: unknown directive"),
    ('Shape "sphere"', -1, "only allowed after WorldBegin"),
    ('WorldBegin\nShape "trianglemesh" "point3 P" [ 0 0 0 1 0 0 0 1 0 0 0 1 ]', -1, "indices"),
    ('WorldBegin\nMaterial "diffuse" "rgb reflectance" [ 0.5 1.5 0.5 ]\nShape "sphere"', -1, "<string>:2: parameter \"reflectance\": RGB parameter has component value > 1.0"),
    ('WorldBegin\nMaterial "conductor" "rgb k" [ 0.5 -1 0.5 ]\nShape "sphere"', -1, "negative component"),
    ('WorldBegin\nTexture "t" "spectrum" "imagemap" "string filename" "nope.png"\nShape "sphere"', -1, "<string>:2: unable to read"),
    ('WorldBegin\nTexture "t" "spectrum" "imagemap" "string filename" "a.exr"\nShape "sphere"', -2, "Unsupported file extension"),
    ('WorldBegin\nTexture "t" "float" "imagemap" "string filename" "a.png" "string filter" "cubic"\nShape "sphere"', -1, "Unknown filter function"),
    ('WorldBegin\nTexture "t" "float" "wrinkled"\nShape "sphere"', -1, "Texture wrinkled unknown"),
    ('WorldBegin\nLightSource "infinite" "string filename" "e.png" "point3 portal" [ 0 0 0 1 0 0 1 1 0 0 1 0 ]\nShape "sphere"', -2, "portal"),
    ('WorldBegin\nImport "other.pbrt"', -2, "Import"),
    ('WorldBegin\nNamedMaterial "nope"', -1, "named material not found"),
    ('WorldBegin\nAttributeEnd', -1, "Unmatched"),
    # the graphics-state stack remembers what pushed each entry (scene.rs:1190-1192, 1693-1712, 1929-1962)
    ('WorldBegin\nObjectBegin "a"\nAttributeEnd\nObjectEnd', -1, "<string>:3: Mismatched nesting: open ObjectBegin from <string>:2 at attribute_end"),
    ('WorldBegin\nObjectBegin "a"\nAttributeBegin\nShape "sphere"\nObjectEnd', -1, "<string>:5: Mismatched nesting: open AttributeBegin from <string>:3 at ObjectEnd"),
    ('WorldBegin\nAttributeBegin\nShape "sphere"', -1, "Missing end to AttributeBegin from <string>:2"),
    ('WorldBegin\nObjectBegin "a"\nShape "sphere"', -1, "unmatched ObjectBegin from <string>:2"),
    ('WorldBegin\nShape "sphere"\nObjectEnd', -1, "ObjectEnd called outside"),
    ('WorldBegin\nMakeNamedMedium "fog"', -2, "media"),
    ('WorldBegin\nShape "curve"', -2, "not supported"),
    ('Integrator "bdpt"\nWorldBegin', -1, "Unknown integrator"),
    ('WorldBegin\nShape "sphere" "float radius" [ abc ]', -1, "expected a number"),
    ('WorldBegin', -1, "no shapes"),
])
def test_errors_are_codes_with_line_numbers(lib, text, code, needle):
    out = C.POINTER(abi.ShmPbrtScene)()
    rc = lib.shm_scene_parse_pbrt(text.encode(), None, C.byref(out))
    assert rc == code and not out
    assert needle in lib.shm_last_error().decode(), lib.shm_last_error()


def test_include_cycles_and_depth_are_errors_not_stack_overflows(lib, tmp_path):
    """parser.rs:191-197 recurses into included files without a limit; a cycle must come back as an error code through the ABI."""
    (tmp_path / "a.pbrt").write_text('WorldBegin\nInclude "b.pbrt"\n')
    (tmp_path / "b.pbrt").write_text('Shape "sphere"\nInclude "a.pbrt"\n')
    out = C.POINTER(abi.ShmPbrtScene)()
    assert lib.shm_scene_load_pbrt(str(tmp_path / "a.pbrt").encode(), C.byref(out)) == -1 and not out
    assert "Include cycle" in lib.shm_last_error().decode() and "b.pbrt:2" in lib.shm_last_error().decode()
    (tmp_path / "self.pbrt").write_text('WorldBegin\nShape "sphere"\nInclude "self.pbrt"\n')
    assert lib.shm_scene_load_pbrt(str(tmp_path / "self.pbrt").encode(), C.byref(out)) == -1 and "Include cycle" in lib.shm_last_error().decode()
    for i in range(70):  # a chain without a cycle, deeper than any scene nests files
        (tmp_path / f"c{i}.pbrt").write_text(('WorldBegin\nShape "sphere"\n' if i == 0 else '') + f'Include "c{i + 1}.pbrt"\n')
    (tmp_path / "c70.pbrt").write_text('Shape "sphere"\n')
    assert lib.shm_scene_load_pbrt(str(tmp_path / "c0.pbrt").encode(), C.byref(out)) == -1 and "deeper than 64" in lib.shm_last_error().decode()
    # an ordinary nested include still loads, and the same file may be included twice in sequence (not a cycle)
    (tmp_path / "top.pbrt").write_text('WorldBegin\nInclude "leaf.pbrt"\nTranslate 3 0 0\nInclude "leaf.pbrt"\n')
    (tmp_path / "leaf.pbrt").write_text('Shape "sphere"\n')
    abi.check(lib, lib.shm_scene_load_pbrt(str(tmp_path / "top.pbrt").encode(), C.byref(out)), "shm_scene_load_pbrt")
    assert out.contents.desc.n_primitives == 2
    lib.shm_pbrt_free(out)


def test_load_from_file_and_look_at_blackbody_helpers(lib, tmp_path):
    f = tmp_path / "s1.pbrt"
    f.write_text(S1_TEXT)
    out = C.POINTER(abi.ShmPbrtScene)()
    abi.check(lib, lib.shm_scene_load_pbrt(str(f).encode(), C.byref(out)), "shm_scene_load_pbrt")
    assert out.contents.desc.n_primitives == 5
    lib.shm_pbrt_free(out)
    assert lib.shm_scene_load_pbrt(str(tmp_path / "missing.pbrt").encode(), C.byref(out)) == -1
    # shm_look_at = Transform::look_at's world_from_camera (transform.rs:270-303): an orthonormal frame at `eye` looking at `look`
    m = np.zeros(16, np.float32)
    abi.check(lib, lib.shm_look_at(scn._fptr(np.array([1, 2, 3], np.float32)), scn._fptr(np.array([1, 2, 7], np.float32)), scn._fptr(np.array([0, 1, 0], np.float32)),
                                   scn._fptr(m)), "shm_look_at")
    assert np.array_equal(m.reshape(4, 4), np.array([[1, 0, 0, 1], [0, 1, 0, 2], [0, 0, 1, 3], [0, 0, 0, 1]], np.float32))
    assert lib.shm_look_at(scn._fptr(np.zeros(3, np.float32)), scn._fptr(np.zeros(3, np.float32)), scn._fptr(np.array([0, 1, 0], np.float32)), scn._fptr(m)) == -1
    # blackbody: normalised to 1 at the Wien peak (spectrum.rs:452-461), float64 Planck within float32 rounding
    bb = scn.blackbody_dense(6500.0)
    lam = np.arange(360, 831) * 1e-9
    planck = (2 * 6.62606957e-34 * 299792458.0 ** 2) / (lam ** 5 * (np.exp(6.62606957e-34 * 299792458.0 / (lam * 1.3806488e-23 * 6500.0)) - 1))
    lmax = 2.8977721e-3 / 6500.0
    peak = (2 * 6.62606957e-34 * 299792458.0 ** 2) / (lmax ** 5 * (np.exp(6.62606957e-34 * 299792458.0 / (lmax * 1.3806488e-23 * 6500.0)) - 1))
    assert np.allclose(bb, planck / peak, rtol=2e-5) and bb.max() <= 1.0 + 1e-6


def test_rgb_spectra_png_textures_normal_map_and_environment_light(lib, tmp_path):
    """The rest of the reference's front end: "rgb" values become RgbAlbedo / RgbUnbounded / RgbIlluminantSpectrum by the slot that reads them
    (paramdict.rs:605-656; spectrum.rs:502-509, 536-547, 574-588) through the sRGB rgb2spec table; "imagemap" textures, "normalmap" and the
    "infinite" light's "filename" read PNG files (image.rs:1140-1311) into the ABI's level / texel tables; a named spectrum texture exists
    per spectrum type (scene.rs:268-294, 380-520), so the same image bound to a reflectance and to a conductor's eta is two ShmImageTextures
    over ONE pyramid."""
    from test_image_io import load_png, write_png
    from test_textures import fetch64
    rng = np.random.default_rng(11)
    write_png(tmp_path / "albedo.png", rng.integers(0, 256, size=(8, 8, 3)), 8, 2, filters=(1, 4))
    write_png(tmp_path / "bump.png", rng.integers(0, 256, size=(4, 4, 1)), 8, 0)
    write_png(tmp_path / "nmap.png", rng.integers(0, 256, size=(4, 4, 3)), 8, 2)
    write_png(tmp_path / "env.png", rng.integers(0, 256, size=(16, 16, 4)), 8, 6)
    (tmp_path / "tint.spd").write_text("400 0.1\n550 0.6 700 0.9\n")
    text = """
    Film "rgb" "integer xresolution" 24 "integer yresolution" 16
    Camera "perspective" "float fov" 50
    WorldBegin
    AttributeBegin
      Rotate 30 0 1 0
      LightSource "infinite" "string filename" "env.png" "float scale" 2
    AttributeEnd
    LightSource "point" "rgb I" [ 1 0.5 0.25 ] "point3 from" [ 0 3 2 ]
    Texture "wood" "spectrum" "imagemap" "string filename" "albedo.png" "string filter" "trilinear" "float uscale" 2 "float vdelta" 0.25
    Texture "bumps" "float" "imagemap" "string filename" "bump.png" "string wrap" "clamp" "string encoding" "linear" "float scale" 0.05
    Texture "blend" "spectrum" "mix" "texture tex1" "wood" "rgb tex2" [ 0.9 0.1 0.1 ] "float amount" 0.25
    Texture "plain" "spectrum" "constant" "spectrum value" "tint.spd"
    Material "diffuse" "texture reflectance" "blend" "texture displacement" "bumps"
    Shape "trianglemesh" "point3 P" [ -2 -1 3  2 -1 3  2 -1 7  -2 -1 7 ] "integer indices" [ 0 1 2 0 2 3 ] "point2 uv" [ 0 0 1 0 1 1 0 1 ]
    Material "conductor" "texture eta" "wood" "rgb k" [ 2 3 4 ] "float roughness" 0.2
    Translate 0 0 5
    Shape "sphere" "float radius" 0.7
    Material "coateddiffuse" "rgb reflectance" [ 0.2 0.5 0.7 ] "string normalmap" "nmap.png" "texture albedo" "plain"
    Translate 1.5 0 0
    Shape "sphere" "float radius" 0.5
    AttributeBegin
      AreaLightSource "diffuse" "rgb L" [ 1 1 1 ] "float scale" 4
      Translate -3 2 0
      Shape "sphere" "float radius" 0.25
    AttributeEnd
    """
    got = load(lib, text, str(tmp_path))
    try:
        d = got.contents.desc
        b = scn.SceneBuilder()
        cs = b.use_srgb_color_space()
        # ShmColorSpace: the table and the normalised D65 the Python generators use too, bit for bit
        assert d.color_space.rgb2spec_res == 64 and arr(d.color_space.rgb2spec_scale, 64) == cs["scale"].tobytes()
        assert arr(d.color_space.rgb2spec_data, 3 * 64 ** 3 * 3) == cs["data"].tobytes() and arr(d.color_space.illuminant, 471) == cs["illuminant"].tobytes()
        # image textures: wood (albedo, at the Texture directive), bumps, wood again (unbounded, when the conductor's eta names it), the normal map
        it = [d.image_textures[i] for i in range(d.n_image_textures)]
        assert [(t.spectrum_type, t.filter, t.wrap, t.n_channels, t.has_color_space) for t in it] == [
            (abi.SHM_SPECTRUM_TYPE_ALBEDO, abi.SHM_TEXFILTER_TRILINEAR, abi.SHM_WRAP_REPEAT, 3, 1), (0, abi.SHM_TEXFILTER_BILINEAR, abi.SHM_WRAP_CLAMP, 1, 0),
            (abi.SHM_SPECTRUM_TYPE_UNBOUNDED, abi.SHM_TEXFILTER_TRILINEAR, abi.SHM_WRAP_REPEAT, 3, 1), (0, abi.SHM_TEXFILTER_BILINEAR, abi.SHM_WRAP_REPEAT, 3, 0)]
        assert (it[0].su, it[0].sv, it[0].du, it[0].dv, it[1].scale) == (2.0, 1.0, 0.0, 0.25, np.float32(0.05))
        assert (it[0].first_level, it[0].n_levels) == (it[2].first_level, it[2].n_levels) == (1, 4)  # level 0 of the table is the environment map
        assert (it[3].n_levels, it[1].n_levels) == (1, 3)  # the normal map hands over its finest level only (material.rs:1453-1475)
        texels = np.ctypeslib.as_array(d.texel_data, shape=(d.n_texel_floats,))

        def level(i, nc):
            lv = d.image_levels[i]
            return texels[lv.texel_offset:lv.texel_offset + lv.width * lv.height * nc].reshape(lv.height, lv.width, nc)

        wood, _ = load_png(lib, tmp_path / "albedo.png", "sRGB", pyramid=True)
        for k, lv in enumerate(wood):
            assert np.array_equal(level(1 + k, 3), lv)
        bump, _ = load_png(lib, tmp_path / "bump.png", "linear", wrap=abi.SHM_WRAP_CLAMP, pyramid=True)
        assert np.array_equal(level(it[1].first_level, 1), bump[0])
        nmap, _ = load_png(lib, tmp_path / "nmap.png", "linear")
        assert np.array_equal(level(it[3].first_level, 3), nmap[0])
        env, _ = load_png(lib, tmp_path / "env.png", "sRGB")
        assert d.n_image_lights == 1 and d.image_lights[0].image_level == 0 and np.array_equal(level(0, 3), env[0])
        m = np.array(list(d.image_lights[0].render_from_light)).reshape(4, 4)
        assert np.allclose(m[:3, :3], [[np.cos(np.pi / 6), 0, np.sin(np.pi / 6)], [0, 1, 0], [-np.sin(np.pi / 6), 0, np.cos(np.pi / 6)]], atol=1e-6)
        lights = [d.lights[i] for i in range(d.n_lights)]
        assert [l.kind for l in lights] == [abi.SHM_LIGHT_IMAGE_INFINITE, abi.SHM_LIGHT_POINT, abi.SHM_LIGHT_DIFFUSE_AREA]
        photometric = float(scn.spectrum_to_photometric(cs["illuminant"]))
        assert lights[0].scale == pytest.approx(2.0 / photometric, rel=1e-6)  # light.rs:181: scale / spectrum_to_photometric(cs.illuminant)
        # materials: [default, diffuse(blend), conductor, coateddiffuse]
        diffuse, conductor, coated = d.materials[1], d.materials[2], d.materials[3]
        assert diffuse.a.kind == abi.SHM_SPECTRUM_TEXTURE_NODE and diffuse.float_tex[abi.SHM_FLOATSLOT_DISPLACEMENT] != 0 and diffuse.has_displacement == 1
        node = d.spectrum_textures[diffuse.a.offset]
        leaf1, leaf2 = d.spectrum_textures[node.a].leaf, d.spectrum_textures[node.b].leaf
        assert node.kind == abi.SHM_SPECTEX_MIX and (leaf1.kind, leaf1.offset) == (abi.SHM_SPECTRUM_IMAGE_TEXTURE, 0) and leaf2.kind == abi.SHM_SPECTRUM_RGB_ALBEDO
        assert np.allclose(list(leaf2.rgb_c), fetch64(cs, [0.9, 0.1, 0.1]), rtol=2e-4, atol=1e-6)
        assert (conductor.a.kind, conductor.a.offset) == (abi.SHM_SPECTRUM_IMAGE_TEXTURE, 2) and conductor.b.kind == abi.SHM_SPECTRUM_RGB_UNBOUNDED and conductor.b.c == 8.0
        assert np.allclose(list(conductor.b.rgb_c), fetch64(cs, [0.25, 0.375, 0.5]), rtol=2e-4, atol=1e-6)  # rgb / (2 max), spectrum.rs:538-541
        assert coated.a.kind == abi.SHM_SPECTRUM_RGB_ALBEDO and np.allclose(list(coated.a.rgb_c), fetch64(cs, [0.2, 0.5, 0.7]), rtol=2e-4, atol=1e-6)
        assert coated.normal_map == 4 and coated.c.kind == abi.SHM_SPECTRUM_PIECEWISE_LINEAR and coated.c.n == 3  # the spectrum file
        # RgbIlluminantSpectrum(1, 1, 1) = 2 s(0.5, 0.5, 0.5) D65 = D65 up to the table's fit (the light stores it densely sampled)
        spec = np.ctypeslib.as_array(d.spectrum_data, shape=(d.n_spectrum_floats,))
        area = spec[lights[2].spectrum.offset:lights[2].spectrum.offset + 471]
        assert np.allclose(area, cs["illuminant"], rtol=0.02) and lights[2].scale == pytest.approx(4.0 / float(scn.spectrum_to_photometric(area)), rel=1e-6)
        point = spec[lights[1].spectrum.offset:lights[1].spectrum.offset + 471]
        c = fetch64(cs, [0.5, 0.25, 0.125])
        lam = np.arange(360.0, 831.0)
        x = (c[0] * lam + c[1]) * lam + c[2]
        assert np.allclose(point, 2.0 * (0.5 + 0.5 * x / np.sqrt(1 + x * x)) * cs["illuminant"], rtol=2e-3)
        # and it is a valid scene: the oracle renders it
        o = oracle_py.Oracle(d)
        film, _ = o.render(render.make_params(seed=0, spp=2, max_depth=4), n_threads=4)
        o.close()
        assert np.isfinite(film["rgb_sum"]).all() and film["rgb_sum"].sum() > 0
    finally:
        lib.shm_pbrt_free(got)


def test_attribute_directive_sets_defaults_in_scope(lib):
    """Attribute "target" params (scene.rs:1714-1730 + ParameterDictionary::new_with_unowned, paramdict.rs:440-455): defaults for the shapes /
    lights / materials / textures that follow inside the attribute scope; a directive's own parameter wins; a name repeated by a later Attribute
    resolves to the later value; a name repeated within ONE directive is the reference's DuplicatedParamName error (param.rs:133-139)."""
    text = """
    Camera "perspective"
    WorldBegin
    AttributeBegin
      Attribute "material" "float reflectance" 0.25
      Attribute "shape" "float radius" 2
      Attribute "light" "float scale" 3
      Material "diffuse"
      Shape "sphere"
      Attribute "material" "float reflectance" 0.75
      Material "diffuse"
      Shape "sphere" "float radius" 0.5
      LightSource "point" "float scale" 1
      LightSource "point"
    AttributeEnd
    Material "diffuse"
    Shape "sphere"
    """
    got = load(lib, text)
    try:
        d = got.contents.desc
        assert [d.materials[i].a.c for i in range(d.n_materials)] == [0.5, 0.25, 0.75, 0.5]  # default slot, attribute, the later attribute, out of scope
        assert sorted(d.spheres[i].radius for i in range(d.n_spheres)) == [0.5, 1.0, 2.0]
        assert d.lights[1].scale == pytest.approx(3.0 * d.lights[0].scale, rel=1e-6)
    finally:
        lib.shm_pbrt_free(got)
    out = C.POINTER(abi.ShmPbrtScene)()
    assert lib.shm_scene_parse_pbrt(b'WorldBegin\nAttribute "camera" "float fov" 3\nShape "sphere"', None, C.byref(out)) == -1
    assert "Unknown attribute target camera" in lib.shm_last_error().decode()
    assert lib.shm_scene_parse_pbrt(b'WorldBegin\nMaterial "diffuse" "float reflectance" 0.75 "float reflectance" 0.5\nShape "sphere"', None, C.byref(out)) == -1
    assert '<string>:2: duplicated parameter name "reflectance"' in lib.shm_last_error().decode()


def test_film_output_matrix_and_white_balance(lib):
    """RgbFilm::new (film.rs:524): output_rgb_from_sensor_rgb = rgb_from_xyz(sRGB, built from the primaries and the D65 white, colorspace.rs:38-72)
    x the sensor's matrix — the identity for the cie1931 sensor, or the von Kries transform of Film "whitebalance" (color.rs:404-416) from the
    white of DenselySampledSpectrum::d(T) (spectrum.rs:215-262) to the colour space's."""
    def matrix(film_params=""):
        got = load(lib, f'Film "rgb" {film_params}\nCamera "perspective"\nWorldBegin\nShape "sphere"')
        m = np.array(list(got.contents.output_rgb_from_sensor_rgb), np.float64).reshape(3, 3)
        lib.shm_pbrt_free(got)
        return m
    base = matrix()
    assert np.allclose(base, render.SRGB_FROM_XYZ, atol=2e-3)  # IEC 61966-2-1, up to the D65 table's white point
    assert np.allclose(base @ np.array([0.95047, 1.0, 1.08883]), 1.0, atol=3e-3)  # D65 white -> (1, 1, 1)
    warm = matrix('"float whitebalance" 3200')  # below 4000 K the reference takes a blackbody: a warm illuminant is balanced towards blue
    lam = np.arange(360.0, 831.0) * 1e-9
    t = 3200.0 * 1.4388 / 1.4380
    planck = 1.0 / (lam ** 5 * (np.exp(6.62606957e-34 * 299792458.0 / (lam * 1.3806488e-23 * t)) - 1))
    tb = scn.tables()
    white = np.array([(tb[k].astype(np.float64) * planck).sum() for k in ("CIE_X", "CIE_Y", "CIE_Z")])
    rgb = warm @ (white / white[1])
    assert np.allclose(rgb / rgb[1], 1.0, atol=0.02), rgb  # the illuminant's own white comes out neutral
    assert (base @ (white / white[1]))[2] < 0.4  # ... which it does not without the balance
    # At and above 4000 K the reference's CCT -> xy polynomial reads `2.9678e6 / cct * cct` (spectrum.rs:233, 238: the square was meant to
    # divide), so x is about 3e6 and the "balanced" matrix is far from any colour transform. Restated as written.
    wb = matrix('"float whitebalance" 6500')
    assert np.abs(wb - base).max() > 1.0
    out = C.POINTER(abi.ShmPbrtScene)()
    assert lib.shm_scene_parse_pbrt(b'Film "rgb" "string sensor" "canon_eos_5d"\nWorldBegin\nShape "sphere"', None, C.byref(out)) == -1
    assert "Unknown sensor type" in lib.shm_last_error().decode()


def test_camera_screen_window_aspect_and_render_space(lib):
    """The rest of Camera::create (camera.rs:676-705, 848-890: "frameaspectratio", "screenwindow" as x0 x1 y0 y1) and Option "rendercoordsys"
    (scene.rs:1411-1431 -> CameraTransform::new, camera.rs:507-523): the three rendering spaces place the same scene differently and render
    the same picture."""
    body = """
    LookAt 1 2 6  0 0.5 0  0 1 0
    Camera "perspective" "float fov" 40 {cam}
    Film "rgb" "integer xresolution" 32 "integer yresolution" 24
    Sampler "independent" "integer pixelsamples" 16
    WorldBegin
    AttributeBegin
      AreaLightSource "diffuse" "blackbody L" 6000 "float scale" 8
      Translate 0 4 0
      Shape "sphere" "float radius" 0.7
    AttributeEnd
    Material "diffuse" "float reflectance" 0.6
    Shape "sphere" "float radius" 1
    Shape "trianglemesh" "point3 P" [ -5 -1 -5  5 -1 -5  5 -1 5  -5 -1 5 ] "integer indices" [ 0 2 1 0 3 2 ]
    """
    def scene(option="", cam=""):
        return load(lib, (f'Option "string rendercoordsys" "{option}"\n' if option else "") + body.format(cam=cam))
    imgs, cams = {}, {}
    for space in ("cameraworld", "camera", "world"):
        got = scene(space)
        d = got.contents.desc
        cams[space] = np.array(list(d.camera.render_from_camera), np.float64).reshape(4, 4)
        o = oracle_py.Oracle(d)
        film, _ = o.render(got.contents.params, n_threads=8)
        o.close()
        imgs[space] = film["rgb_sum"] / film["weight_sum"][..., None]
        centre = np.array(list(d.spheres[1].render_from_object), np.float64).reshape(4, 4)[:3, 3]
        if space == "world":
            assert np.allclose(centre, 0.0, atol=1e-6) and np.allclose(cams[space][:3, 3], [1, 2, 6], atol=1e-5)
        elif space == "camera":
            assert np.allclose(cams[space], np.eye(4), atol=1e-6) and np.linalg.norm(centre) == pytest.approx(np.sqrt(1 + 4 + 36), rel=1e-5) and centre[2] > 6
        else:
            assert np.allclose(cams[space][:3, 3], 0.0, atol=1e-6) and np.allclose(centre, [-1, -2, -6], atol=1e-5)
        lib.shm_pbrt_free(got)
    ref = imgs["cameraworld"]
    for space in ("camera", "world"):  # the same picture up to Monte-Carlo noise and float rounding of the different coordinates
        assert abs(imgs[space].mean() - ref.mean()) < 0.03 * ref.mean() and np.abs(imgs[space] - ref).mean() < 0.25 * ref.mean()
    # screen window: half the default extent in x (the film is 4:3 -> default x in [-4/3, 4/3]) doubles the magnification in x only
    base = scene()
    zoom = scene(cam='"float screenwindow" [ -0.6666667 0.6666667 -1 1 ]')
    wide = scene(cam='"float frameaspectratio" 2')
    bx = np.array(list(base.contents.desc.camera.dx_camera))
    zx, zy = np.array(list(zoom.contents.desc.camera.dx_camera)), np.array(list(zoom.contents.desc.camera.dy_camera))
    assert np.allclose(zx, bx / 2, rtol=1e-5) and np.allclose(zy, np.array(list(base.contents.desc.camera.dy_camera)), rtol=1e-5)
    wx = np.array(list(wide.contents.desc.camera.dx_camera))
    assert np.allclose(wx, bx * (2.0 / (32.0 / 24.0)), rtol=1e-5)  # frame 2: screen x in [-2, 2]
    for g in (base, zoom, wide):
        lib.shm_pbrt_free(g)
    out = C.POINTER(abi.ShmPbrtScene)()
    assert lib.shm_scene_parse_pbrt(b'Option "string rendercoordsys" "object"\nWorldBegin\nShape "sphere"', None, C.byref(out)) == -1


def test_example_scene_files_load(lib):
    """examples/scenes/*.pbrt (what examples/render_pbrt.c is pointed at) load and are valid scenes."""
    from pathlib import Path
    for f in sorted((Path(__file__).resolve().parents[1] / "examples" / "scenes").glob("*.pbrt")):
        out = C.POINTER(abi.ShmPbrtScene)()
        abi.check(lib, lib.shm_scene_load_pbrt(str(f).encode(), C.byref(out)), f.name)
        s = out.contents
        s.params.samples_per_pixel = 1
        o = oracle_py.Oracle(s.desc)
        film, _ = o.render(s.params, n_threads=8)
        o.close()
        assert np.isfinite(film["rgb_sum"]).all() and film["rgb_sum"].sum() > 0, f.name
        lib.shm_pbrt_free(out)
