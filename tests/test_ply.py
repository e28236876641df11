"""TriQuadMesh::read_ply (shape/mesh.rs:179-358) through the host mirror `shm_ply_read`, and the "plymesh" shape built from it.
The reference's own test (mesh.rs:361-377, basic_ply_read) reads test_files/cube.ply — a file that is not in its repository —
and asserts 8 vertices, no triangles, 6 quads: the cube written here is held to the same assertions."""
import ctypes as C
import struct

import numpy as np
import pytest

from shimmer_amd import abi, render, scenes
from shimmer_amd.scene import SceneBuilder

CUBE_V = [(-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1), (-1, -1, 1), (1, -1, 1), (1, 1, 1), (-1, 1, 1)]
CUBE_Q = [(0, 3, 2, 1), (4, 5, 6, 7), (0, 1, 5, 4), (2, 3, 7, 6), (1, 2, 6, 5), (0, 4, 7, 3)]


def write_ply(path, fmt, vertices, faces, normals=None, uvs=None, uv_names=("u", "v"), index_type="int", extra_header="", vertex_type="float"):
    hdr = ["ply", f"format {fmt} 1.0", "comment made by tests/test_ply.py", f"element vertex {len(vertices)}"]
    hdr += [f"property {vertex_type} {k}" for k in "xyz"]
    if normals is not None:
        hdr += [f"property float {k}" for k in ("nx", "ny", "nz")]
    if uvs is not None:
        hdr += [f"property float {k}" for k in uv_names]
    hdr += [f"element face {len(faces)}", f"property list uchar {index_type} vertex_indices"]
    if extra_header:
        hdr.append(extra_header)
    hdr.append("end_header")
    with open(path, "wb") as f:
        f.write(("\n".join(hdr) + "\n").encode())
        e = "<" if fmt == "binary_little_endian" else ">"
        for i, v in enumerate(vertices):
            row = list(v) + (list(normals[i]) if normals is not None else []) + (list(uvs[i]) if uvs is not None else [])
            if fmt == "ascii":
                f.write((" ".join(repr(float(x)) for x in row) + "\n").encode())
            else:
                f.write(struct.pack(e + ("d" if vertex_type == "double" else "f") * 3, *row[:3]) + struct.pack(e + "f" * (len(row) - 3), *row[3:]))
        for face in faces:
            if fmt == "ascii":
                f.write((" ".join(str(x) for x in [len(face), *face]) + "\n").encode())
            else:
                f.write(struct.pack("B", len(face)) + struct.pack(e + ("i" if index_type == "int" else "I") * len(face), *face))


def read(lib, path):
    m = abi.ShmPlyMesh()
    rc = lib.shm_ply_read(str(path).encode(), C.byref(m))
    if rc != 0:
        return rc, lib.shm_last_error().decode()
    out = dict(p=np.ctypeslib.as_array(m.p, shape=(m.n_vertices, 3)).copy(), n=np.ctypeslib.as_array(m.n, shape=(m.n_vertices, 3)).copy(),
               uv=np.ctypeslib.as_array(m.uv, shape=(m.n_vertices, 2)).copy(),
               tri=np.ctypeslib.as_array(m.tri_indices, shape=(m.n_tri_indices,)).copy() if m.n_tri_indices else np.zeros(0, np.int32),
               quad=np.ctypeslib.as_array(m.quad_indices, shape=(m.n_quad_indices,)).copy() if m.n_quad_indices else np.zeros(0, np.int32))
    lib.shm_ply_free(C.byref(m))
    assert not m.p
    return 0, out


@pytest.mark.parametrize("fmt", ["ascii", "binary_little_endian", "binary_big_endian"])
def test_basic_ply_read(lib, tmp_path, fmt):
    """mesh.rs:361-377: 8 vertices, no triangles, 6 quads — and the quad corner order v0 v1 v3 v2 of a bilinear patch (:262-266)."""
    write_ply(tmp_path / "cube.ply", fmt, CUBE_V, CUBE_Q)
    rc, m = read(lib, tmp_path / "cube.ply")
    assert rc == 0
    assert len(m["p"]) == 8 and m["tri"].size == 0 and m["quad"].size == 6 * 4
    assert np.array_equal(m["p"], np.array(CUBE_V, np.float32))
    assert np.array_equal(m["quad"].reshape(6, 4), np.array([(a, b, d, c) for a, b, c, d in CUBE_Q]))
    assert not m["n"].any() and not m["uv"].any() and m["n"].shape == (8, 3)  # absent properties stay PlyVertex::new's zeros


def test_mixed_faces_normals_and_uv_aliases(lib, tmp_path):
    rng = np.random.default_rng(0)
    v = rng.normal(size=(6, 3)).astype(np.float32)
    n = rng.normal(size=(6, 3)).astype(np.float32)
    uv = rng.uniform(size=(6, 2)).astype(np.float32)
    faces = [(0, 1, 2), (2, 3, 4, 5), (1, 3, 5)]
    for names in (("u", "v"), ("s", "t"), ("texture_u", "texture_v"), ("texture_s", "texture_t")):
        write_ply(tmp_path / "m.ply", "binary_little_endian", v, faces, normals=n, uvs=uv, uv_names=names)
        rc, m = read(lib, tmp_path / "m.ply")
        assert rc == 0 and np.array_equal(m["p"], v) and np.array_equal(m["n"], n) and np.array_equal(m["uv"], uv)
        assert m["tri"].tolist() == [0, 1, 2, 1, 3, 5] and m["quad"].tolist() == [2, 3, 5, 4]
    write_ply(tmp_path / "a.ply", "ascii", v, faces, normals=n, uvs=uv)
    rc, m = read(lib, tmp_path / "a.ply")
    assert rc == 0 and np.array_equal(m["p"], v) and np.array_equal(m["uv"], uv)  # repr(float32 as float) round-trips


def test_what_the_reference_panics_on_is_an_error(lib, tmp_path):
    cases = {
        "pentagon": dict(faces=[(0, 1, 2, 3, 4)], msg="Only tris and quads are supported"),              # mesh.rs:270
        "uint_indices": dict(faces=[(0, 1, 2)], index_type="uint", msg="Face: Unexpected key/value"),    # ListUInt is not ListInt (:349-356)
        "double_positions": dict(faces=[(0, 1, 2)], vertex_type="double", msg="Vertex: Unexpected key/value"),
        "index_out_of_range": dict(faces=[(0, 1, 9)], msg="out of range"),                               # :278-287
        "other_element": dict(faces=[(0, 1, 2)], extra_header="element edge 0", msg="Unexpected element: edge"),
    }
    for name, c in cases.items():
        write_ply(tmp_path / f"{name}.ply", "binary_little_endian", CUBE_V, c["faces"], index_type=c.get("index_type", "int"),
                  extra_header=c.get("extra_header", ""), vertex_type=c.get("vertex_type", "float"))
        rc, msg = read(lib, tmp_path / f"{name}.ply")
        assert rc == -1 and c["msg"] in msg, (name, rc, msg)
    rc, msg = read(lib, tmp_path / "missing.ply")
    assert rc == -1 and "Unable to read PLY file" in msg
    (tmp_path / "short.ply").write_bytes(b"ply\nformat binary_little_endian 1.0\nelement vertex 2\nproperty float x\nend_header\n\x00\x00")
    rc, msg = read(lib, tmp_path / "short.ply")
    assert rc == -1 and "end of file" in msg


def test_plymesh_renders_like_the_same_mesh_given_directly(lib, tmp_path):
    """"plymesh" = TriangleMesh + BilinearPatchMesh of the file (shape/shape.rs:97-135): a scene built from the file and the same
    scene built from the arrays give the same film in the oracle, bit for bit."""
    import oracle_py
    tri_faces = [(0, 3, 2), (0, 2, 1)]  # one cube face as two triangles, the rest as quads
    write_ply(tmp_path / "cube.ply", "binary_big_endian", CUBE_V, tri_faces + CUBE_Q[1:])

    def scene(from_file):
        sc = scenes.cornell_box(lib, 24, 24)
        b = sc.builder
        rfo = np.eye(4, dtype=np.float32)
        rfo[:3, :3] *= np.float32(0.25)
        rfo[:3, 3] = np.array([0.1, 0.2, -3.0], np.float32)  # render space: in front of the camera, above the floor
        m = b.material_diffuse(0.6)
        if from_file:
            b.add_ply(lib, tmp_path / "cube.ply", m, render_from_object=rfo)
        else:
            p = (np.array(CUBE_V, np.float32) * np.float32(0.25) + rfo[:3, 3]).astype(np.float32)
            z3, z2 = np.zeros((8, 3), np.float32), np.zeros((8, 2), np.float32)
            b.add_mesh(p, np.array(tri_faces, np.uint32), m, n=z3, uv=z2)
            b.add_patch_mesh(p, np.array([(a, bb, d, c) for a, bb, c, d in CUBE_Q[1:]], np.uint32), m, n=z3, uv=z2)
        desc, _ = b.build(lib)
        o = oracle_py.Oracle(desc)
        try:
            film, stats = o.render(render.make_params(spp=4, max_depth=4, seed=2), n_threads=4)
        finally:
            o.close()
        return film, stats

    f1, s1 = scene(True)
    f2, s2 = scene(False)
    assert s1["rays_closest"] == s2["rays_closest"] and f1.tobytes() == f2.tobytes()
    assert np.isfinite(f1["rgb_sum"]).all()
