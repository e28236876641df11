"""Row a32 (BVH build + flatten) pinned two ways that do not share code with the product (missing #3):
  * the reference's own Bounds3 vectors (bounding_box.rs:699-733 union_point / union, :950-995 surface_area / volume / max_dimension) and
    math.rs:549-570 (lerp, difference_of_products, evaluate_polynomial) replayed against the helpers the builder calls;
  * an INDEPENDENT numpy restatement of BvhAggregate::new (aggregate.rs:207-467, split method "middle") written from the Rust text on
    index SETS — which primitives go left is decided by `centroid < pmid` alone, so itertools::partition's unstable order cannot matter —
    compared with shm_bvh_build node for node (bounds bits, child offsets, axes, leaf order) on the Cornell box and on the S3 proxy."""
import ctypes as C

import numpy as np
import pytest

from shimmer_amd import abi, scenes

f32 = np.float32


def probe(lib, a, b=None, p=None):
    a = np.asarray(a, f32)
    b = np.asarray(a if b is None else b, f32)
    p = np.asarray([0, 0, 0] if p is None else p, f32)
    out = np.zeros(16, f32)
    ptr = lambda x: x.ctypes.data_as(abi.c_float_p)
    abi.check(lib, lib.shm_bounds3_probe(ptr(a), ptr(b), ptr(p), ptr(out)), "shm_bounds3_probe")
    return out


def test_reference_bounds3_vectors(lib):
    # bounding_box.rs:698-708 bounds3_union_point: [(0,0,1), (1,1,1)] + point (-1,-1,-1) -> [(-1,-1,-1), (1,1,1)]
    o = probe(lib, [0, 0, 1, 1, 1, 1], p=[-1, -1, -1])
    assert list(o[6:12]) == [-1, -1, -1, 1, 1, 1]
    # bounding_box.rs:722-733 bounds3_union: [(0,0,0), (1,1,0)] U [(10,10,10), (11,11,11)] -> [(0,0,0), (11,11,11)]
    o = probe(lib, [0, 0, 0, 1, 1, 0], b=[10, 10, 10, 11, 11, 11])
    assert list(o[0:6]) == [0, 0, 0, 11, 11, 11]
    # bounding_box.rs:953-967 bounds3_surface_area / bounds3_volume: the cube [0, 4]^3 -> 96, 64
    o = probe(lib, [0, 0, 0, 4, 4, 4])
    assert o[12] == 96.0 and o[13] == 64.0
    # bounding_box.rs:982-998 bounds3_max_dimension
    assert probe(lib, [0, 0, 0, 5, 4, 4])[14] == 0 and probe(lib, [0, 0, 0, 5, 6, 4])[14] == 1 and probe(lib, [0, 0, 0, 5, 4, 10])[14] == 2
    # Bounds3::default() (bounding_box.rs:568-581) is the identity of union: an inverted box
    big = np.finfo(f32).max
    o = probe(lib, [big, big, big, -big, -big, -big], b=[1, 2, 3, 4, 5, 6])
    assert list(o[0:6]) == [1, 2, 3, 4, 5, 6]
    # (the 2-D vectors — bounds2_union, bounds2_area, bounds2_max_dimension, bounding_box.rs:686-721, 941-981 — are the same functions on two
    #  components; Bounds2i only reaches the path as pixel bounds and tiles, which tests/test_host_mirror.py checks against tile.rs)


def test_reference_math_vectors(orc):
    """math.rs:549-570: lerp(0.45, 0, 10) = 4.5; difference_of_products(10, 10, 5, 5) = 75; poly(2, [1, 2, 3]) = 17 (fast_polynomial:
    c0 + c1 x + c2 x^2). The last is the crate the RGB sigmoid (color.rs:359) and the equal-area mapping go through; here its FMA order is
    defined in shm/spectrum.h / shm/texture.h: the vector is exact in any order."""
    g = orc
    assert g.orc_fn_lerp(f32(0.45), f32(0.0), f32(10.0)) == 4.5
    assert g.orc_fn_difference_of_products(f32(10), f32(10), f32(5), f32(5)) == 75.0
    assert g.orc_fn_poly3(f32(2.0), f32(1.0), f32(2.0), f32(3.0)) == 17.0


# ---- the independent builder ------------------------------------------------------------------------------------------------------
def numpy_bvh(bounds):
    """aggregate.rs:207-467 on index sets, float32 throughout. Returns (nodes as rows [bmin 3, bmax 3, offset, n_prims, axis], leaf order).
    Raises if a tie would make the result depend on an unstable ordering (then the fixture is not usable, not the product wrong)."""
    bounds = np.asarray(bounds, f32)
    mn, mx = bounds[:, :3], bounds[:, 3:]
    cen = (f32(0.5) * mn + mx * f32(0.5)).astype(f32)   # BvhPrimitive::centroid, aggregate.rs:489-492: 0.5 * min + max * 0.5
    nodes, order = [], []

    def build(idx):
        me = len(nodes)
        nodes.append(None)
        bmin, bmax = mn[idx].min(axis=0), mx[idx].max(axis=0)
        d = (bmax - bmin).astype(f32)
        area = f32(2.0) * (d[0] * d[1] + d[0] * d[2] + d[1] * d[2])

        def leaf():
            nodes[me] = (bmin, bmax, len(order), len(idx), 0)
            order.extend(int(i) for i in idx)
        if area == 0 or len(idx) == 1:
            return leaf()
        cmin, cmax = cen[idx].min(axis=0), cen[idx].max(axis=0)
        e = (cmax - cmin).astype(f32)
        dim = 0 if (e[0] > e[1] and e[0] > e[2]) else (1 if e[1] > e[2] else 2)   # max_dimension, bounding_box.rs:404-415
        if cmax[dim] == cmin[dim]:
            return leaf()
        pmid = f32((cmin[dim] + cmax[dim]) / f32(2.0))
        c = cen[idx, dim]
        left = idx[c < pmid]
        right = idx[~(c < pmid)]
        if len(left) == 0 or len(right) == 0:   # aggregate.rs:372-385: fall back to the median (pdqselect: membership by rank)
            k = len(idx) // 2
            srt = np.sort(c)
            if srt[k - 1] == srt[k]:
                raise RuntimeError("tie at the median: order-dependent")
            left, right = idx[c < srt[k]], idx[c >= srt[k]]
        build(left)
        second = len(nodes)
        build(right)
        l, r = nodes[me + 1], nodes[second]
        nodes[me] = (np.minimum(l[0], r[0]), np.maximum(l[1], r[1]), second, 0, dim)

    import sys
    sys.setrecursionlimit(10000)
    build(np.arange(len(bounds)))
    return nodes, order


def product_bvh(lib, bounds):
    b = np.ascontiguousarray(bounds, f32)
    n = len(b)
    nodes = (abi.ShmBvhNode * (2 * n))()
    n_nodes = C.c_uint32()
    order = (C.c_uint32 * n)()
    abi.check(lib, lib.shm_bvh_build(b.ctypes.data_as(abi.c_float_p), n, 0, nodes, C.byref(n_nodes), order), "shm_bvh_build")
    return [nodes[i] for i in range(n_nodes.value)], list(order)


@pytest.mark.parametrize("scene", ["S2_cornell", "S3_n12", "S4_small"])
def test_bvh_build_equals_independent_numpy_build(lib, scene):
    sc = {"S2_cornell": lambda: scenes.cornell_box(lib, 32, 32), "S3_n12": lambda: scenes.ganesha_proxy(lib, 32, 32, n=12),
          "S4_small": lambda: scenes.crown_proxy(lib, 30, 42, level=1, n_glass=6, n_gold=2)}[scene]()
    bounds = sc.info["bounds"]
    want_nodes, want_order = numpy_bvh(bounds)
    got_nodes, got_order = product_bvh(lib, bounds)
    assert len(got_nodes) == len(want_nodes)
    n_leaf_prims = [g.n_prims for g in got_nodes if g.n_prims]
    assert sum(n_leaf_prims) == len(bounds)
    # (one primitive per leaf, aggregate.rs:326 — except the two triangles of an axis-aligned quad, which share one AABB, hence one centroid,
    #  and end in a two-primitive leaf through the degenerate-centroid-bounds exit, aggregate.rs:345)
    assert max(n_leaf_prims) <= 2 and len(got_nodes) >= 2 * (len(bounds) - n_leaf_prims.count(2)) - 1
    # DFS leaf order = the device's primitive order. Inside a leaf of several primitives (identical centroids) the order is whatever
    # itertools::partition's swaps left (unstable, crate not vendored: unpinned) — compared as a set there, exactly everywhere else
    for g in got_nodes:
        if g.n_prims == 1:
            assert got_order[g.offset] == want_order[g.offset]
        elif g.n_prims > 1:
            assert sorted(got_order[g.offset:g.offset + g.n_prims]) == sorted(want_order[g.offset:g.offset + g.n_prims])
    for i, (g, w) in enumerate(zip(got_nodes, want_nodes)):
        assert np.array_equal(np.array(list(g.bmin), f32).view(np.uint32), np.asarray(w[0], f32).view(np.uint32)), i
        assert np.array_equal(np.array(list(g.bmax), f32).view(np.uint32), np.asarray(w[1], f32).view(np.uint32)), i
        assert (g.offset, g.n_prims, g.axis) == (w[2], w[3], w[4]), i
    # ... and what the scene generator uploaded is that tree (it called the same entry): the description's node array, bit for bit
    d = sc.desc
    assert d.n_nodes == len(got_nodes)
    for i in (0, 1, len(got_nodes) // 2, len(got_nodes) - 1):
        assert bytes(d.nodes[i]) == bytes(got_nodes[i])
