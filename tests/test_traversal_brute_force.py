"""BvhAggregate::intersect / intersect_predicate (aggregate.rs:71-203) with Triangle::intersect_triangle (triangle.rs:173-302) against NO acceleration structure at all:
every ray of a random batch is tested in float64 (Möller-Trumbore, written here) against every triangle of the scene, and the nearest of those hits must be what the
oracle's traversal returns — same hit / miss decision, same distance; the any-hit entry must say "occluded" exactly when that nearest hit lies inside the ray's extent.
A traversal that skipped a subtree, ordered children wrongly with a shrinking t_max, or mis-built a node's bounds fails here whatever the Rust text says
(tests/test_bvh_independent.py rebuilds the tree itself from the text; the GPU suite holds the device's traversal to the oracle's, hit records and visit counters)."""
import numpy as np
import pytest

import oracle_py
from shimmer_amd import scenes


def scene_triangles(desc):
    tris = []
    for mi in range(desc.n_meshes):
        m = desc.meshes[mi]
        p = np.array([m.p[i] for i in range(3 * m.n_vertices)], np.float64).reshape(-1, 3)
        vi = np.array([m.vertex_indices[i] for i in range(3 * m.n_triangles)], np.int64).reshape(-1, 3)
        tris.append(p[vi])
    return np.concatenate(tris, 0)  # (n, 3, 3)


def brute_force(tris, o, d, t_max):
    """nearest Möller-Trumbore hit of one ray over all triangles: (t, smallest barycentric of that hit) or (inf, 0)."""
    p0, e1, e2 = tris[:, 0], tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0]
    h = np.cross(d, e2)
    det = np.einsum("ij,ij->i", e1, h)
    ok = np.abs(det) > 1e-300
    inv = np.where(ok, 1.0 / np.where(ok, det, 1.0), 0.0)
    s = o - p0
    u = np.einsum("ij,ij->i", s, h) * inv
    q = np.cross(s, e1)
    v = np.einsum("ij,ij->i", q, np.broadcast_to(d, q.shape)) * inv
    t = np.einsum("ij,ij->i", e2, q) * inv
    inside = ok & (u >= 0) & (v >= 0) & (u + v <= 1) & (t > 0) & (t < t_max)
    if not inside.any():
        return np.inf, 0.0, 0
    k = np.argmin(np.where(inside, t, np.inf))
    # how many triangles are hit within a hair of the nearest distance (shared edges, coincident faces)
    near = int(np.sum(inside & (np.abs(t - t[k]) <= 1e-6 * max(1.0, abs(t[k])))))
    return float(t[k]), float(min(u[k], v[k], 1.0 - u[k] - v[k])), near


SCENES = {
    "S3_n12": lambda lib: scenes.ganesha_proxy(lib, 16, 16, n=12),
    "S2_cornell": lambda lib: scenes.cornell_box(lib, 16, 16),
    "S4_small": lambda lib: scenes.crown_proxy(lib, 15, 21, level=1, n_glass=6, n_gold=2),
}


@pytest.mark.parametrize("name", list(SCENES))
def test_traversal_finds_the_nearest_of_all_triangles(lib, name):
    sc = SCENES[name](lib)
    desc = sc.desc
    assert desc.n_spheres == 0 and desc.n_patch_meshes == 0 and desc.n_instances == 0
    tris = scene_triangles(desc)
    assert len(tris) == desc.n_primitives  # nothing but triangles in these scenes: the brute force sees everything the traversal sees
    o = oracle_py.Oracle(desc)
    lo, hi = tris.reshape(-1, 3).min(0), tris.reshape(-1, 3).max(0)
    rng = np.random.default_rng(31)
    n = 1500
    org = lo + (hi - lo) * (0.05 + 0.9 * rng.random((n, 3)))
    z, phi = 2.0 * rng.random(n) - 1.0, 2.0 * np.pi * rng.random(n)
    r = np.sqrt(1.0 - z * z)
    d = np.stack([r * np.cos(phi), r * np.sin(phi), z], 1)
    t_max = np.where(rng.random(n) < 0.3, rng.uniform(0.05, 1.0, n) * np.linalg.norm(hi - lo), np.inf)
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3], rays[:, 3:6], rays[:, 6] = org, d, t_max
    hits, st = o.trace(rays)
    occ, _ = o.trace(rays, any_hit=True)
    rays64 = rays.astype(np.float64)  # (the brute force sees the float32 rays the traversal saw)
    n_hit = n_edge = 0
    for i in range(n):
        t, bmin, near = brute_force(tris, rays64[i, 0:3], rays64[i, 3:6], float(rays64[i, 6]))
        got_hit = hits["prim"][i] >= 0
        # a hit within 1e-5 (barycentric) of an edge, or within a hair of the ray's end, may fall either way between float32 edge functions and float64: not judged
        if np.isfinite(t) and (bmin < 1e-5 or abs(t - rays64[i, 6]) <= 1e-5 * max(1.0, t)):
            n_edge += 1
            continue
        assert got_hit == np.isfinite(t), (name, i, t, hits[i])
        assert bool(occ[i]) == np.isfinite(t), (name, i, t, occ[i])
        if got_hit:
            n_hit += 1
            assert abs(float(hits["t"][i]) - t) <= 3e-5 * max(1.0, t), (name, i, t, hits[i], near)
    o.close()
    assert n_hit > n // 5 and n_edge < n // 50, (n_hit, n_edge)  # (the crown proxy stands in the open: most rays leave)
    assert st["rays_closest"] == n
