// shm/scene.h — flat, pointer-based view of a scene as the kernels (and the CPU oracle) read it.
//
// Data layout in HBM (DESIGN.md §"Data layout"):
//   nodes[]      32-B LinearBvhNode records, DFS order (aggregate.rs:425-481 narrowed from 64 B; the device copy: sibling pairs + link words, wavefront.h)
//   prim_recs[]  64-B ALIGNED records in BVH LEAF ORDER: the three triangle vertices pre-gathered (the reference
//                chases Vec<Arc<Primitive>> -> Arc<Shape> -> Arc<TriangleMesh> -> Vec<usize> -> Vec<Point3f>,
//                shape/triangle.rs:148-160) + kind/shape index + mesh id + global triangle id + the hit's material and emitter index.
//                A leaf's primitive_offset indexes this array directly: three 16-B loads of one 64-byte line per candidate (a traversal reads 48 of the 64 bytes;
//                bench.py's roofline_hbm_algorithmic keeps SURVEY 8d's 48 B per primitive tested: the survey's convention, not the record's size).
//   primitives[] shape kind / index + material / area-light ids per leaf-order slot (primitive.rs:66-130): the ABI's array; the device reads material and emitter from the
//                record above (flatten copies them once), `primitives[]` only where a kernel wants the shape index of a slot
//   vi/vn/vs/vuv global per-vertex shading arrays (only read when a mesh has N/S/uv)
//   texel_data   every MIP level of every image texture as f32, [level][row from the top][x][channel]; image_levels[] holds
//                {width, height, offset}; the rgb2spec coefficient table (3 x res^3 x 3 floats) sits beside it
#pragma once
#include <assert.h>
#include "shapes.h"
#include "patch.h"
#include "spectrum.h"

namespace shm {

struct PrimRec {
    Float p0[3], p1[3], p2[3];
    uint32_t kind_index;  // bit 31: sphere, bit 30: bilinear patch; low bits: sphere / patch index (0 for a triangle)
    uint32_t mesh;        // triangle: mesh id                 | bilinear patch: p11.x (bits)
    uint32_t tri;         // triangle: global triangle index   | bilinear patch: p11.y
    // ... and, filling the record to 64 bytes, the ShmPrimitive fields the shading kernels read of a hit: one aligned 64-byte fetch gives a triangle test its three vertices
    // (48-byte records straddled a 64-byte boundary every other time) and a vertex its material and emitter as well (a second gather from `primitives` before)
    uint32_t material;
    int32_t area_light;
    uint32_t pad[2];      // bilinear patch: pad[0] = p11.z — the fourth corner travels with the record (PatchExtra keeps its own copy for the shading kernels)
};
static_assert(sizeof(PrimRec) == 64, "PrimRec is one aligned 64-byte record");
// A bilinear patch keeps p00, p10, p01 in {p0, p1, p2}; its fourth corner and the per-patch constants live here.
struct PatchExtra {
    Float p11[3];
    Float area;      // BilinearPatch::new (bilinear_patch.rs:40-69)
    uint32_t flags;  // bit 0: is_rectangle (:108-142), bit 1: reverse_orientation ^ transform_swaps_handedness,
                     // bit 2: the mesh has per-vertex normals, bit 3: per-vertex uv (attributes through SceneView::patch_vi)
    uint32_t pad[3];
};
static_assert(sizeof(PatchExtra) == 32, "PatchExtra must be 32 bytes");
constexpr uint32_t PRIM_PATCH_BIT = 0x40000000u;
constexpr uint32_t PRIM_INSTANCE_BIT = 0x20000000u;  // a TransformedPrimitive: the index is into SceneView::instances, p0..p2 unused
constexpr uint32_t PRIM_DEGENERATE_BIT = 0x10000000u;  // a triangle whose edge cross product is exactly zero (triangle.rs:182-185 rejects it before anything
                                                       // else): evaluated once at scene creation with the shared arithmetic, so the traversal kernel
                                                       // skips 20 instructions of every leaf test (the oracle evaluates the check itself)
constexpr uint32_t PRIM_INDEX_MASK = 0x0fffffffu;
constexpr uint32_t PRIM_SPHERE_BIT = 0x80000000u;

enum : uint32_t { MESH_HAS_N = 1, MESH_HAS_S = 2, MESH_HAS_UV = 4, MESH_FLIP = 8 };

struct SceneView {
    const ShmBvhNode* nodes;
    uint32_t n_nodes;
    const PrimRec* prim_recs;
    const ShmPrimitive* primitives;
    uint32_t n_primitives;
    const uint32_t* mesh_flags;
    const uint32_t* vi;   // 3 per triangle, global vertex ids
    const Float* vn;      // 3 per vertex
    const Float* vs;      // 3 per vertex
    const Float* vuv;     // 2 per vertex
    const ShmSphere* spheres;
    const PatchExtra* patches;
    const uint32_t* patch_vi;  // 4 per patch: global ids into patch_vn / patch_vuv (only read for patches with attributes)
    const Float* patch_vn;     // 3 per patch-mesh vertex
    const Float* patch_vuv;    // 2 per patch-mesh vertex
    const ShmMaterial* materials;
    const ShmLight* lights;
    uint32_t n_lights;
    const struct PrimRec* light_prim_recs;  // per light: a copy of prim_recs[light.primitive] (area lights; zeros otherwise) — small enough to sit beside the light table in LDS,
                                           // so that a light sample does not start with a dependent fetch of the emitter's record from the 200 MB array (null: not built)
    const uint32_t* infinite_lights;  // indices into lights (integrator.rs:86-92)
    uint32_t n_infinite_lights;
    const Float* spectrum_data;
    Float scene_radius;  // Light::preprocess(scene_bounds): Bounds3::bounding_sphere of the aggregate's bounds (bounding_box.rs:460-468)
    ShmCamera camera;
    int32_t pixel_bounds[4];
    Float filter_radius[2];
    Float imaging_ratio;
    Float max_component_value;
    const Float* sensor_r_bar;
    const Float* sensor_g_bar;
    const Float* sensor_b_bar;
    // image textures (ABI v6; all null / 0 in a scene without them)
    const ShmImageTexture* image_textures;
    const ShmImageLevel* image_levels;
    const Float* texel_data;
    uint32_t rgb2spec_res;
    const Float* rgb2spec_scale;
    const Float* rgb2spec_data;
    const Float* cs_illuminant;  // 471 floats, 360..=830
    const Float* ewa_lut;        // MIP_FILTER_LUT, 128 floats
    const ShmInstance* instances;
    const ShmFloatTexture* float_textures;
    const struct FloatTexRange* ftex_ranges;  // per node: its evaluation program (children first) in ftex_ops
    const struct FloatTexOp* ftex_ops;
    const ShmSpectrumTexture* spectrum_textures;
    const struct FloatTexRange* stex_ranges;  // the same post-order programs for SpectrumTexture trees
    const struct FloatTexOp* stex_ops;
    // image infinite lights: per light its transform + image + the two PiecewiseConstant2D distributions, flattened into dist_data
    const struct ImageLightRec* image_lights;
    const Float* dist_data;
    uint32_t quirks_off;  // ShmRenderParams::disable_reference_quirks of the render in flight (0 = reference-exact; set per render call)
    // device only: the workgroup's copy of this object in LDS, which the texture evaluators that are real calls read the scene through (shm/texture.h,
    // SHM_SV_FOR_CALL; set by stage_scene_tables_tex in the kernels that evaluate textures, never read on the host)
    const SceneView* call_copy;
    // device only (null on the host): per instance, a copy of the device record of its tree's root node — what the traversal kernel tests where a ray enters the instance,
    // fetched with the instance's matrices instead of behind them (render.hip, upload)
    const ShmBvhNode* inst_roots;
};

// A FloatTexture tree flattened at scene creation into a post-order program: evaluating the ops in order (each into slot k of a
// small value array) and combining children by their slots reproduces FloatTexture::evaluate's recursion without device recursion.
// The reference's lazy branches ("if amt != 1 { t1 = tex1.evaluate() }", texture.rs:244-305) only skip work: evaluation has no side
// effects, so selecting 0 for a skipped child gives the same value.
constexpr int FTEX_MAX_OPS = 32;
constexpr int STEX_MAX_OPS = 8;
struct FloatTexOp {
    uint32_t node;       // index into SceneView::float_textures
    uint8_t a, b, c, pad;  // slots of the children within this program
};
struct FloatTexRange {
    uint32_t first, count;  // into SceneView::ftex_ops; the root is the last op
};

// PiecewiseConstant2D (sampling.rs:113-179) of an n x n image, flattened: conditional func [n*n], conditional cdf [n*(n+1)],
// conditional integrals = marginal func [n], marginal cdf [n+1]; marginal integral beside the offsets.
struct Dist2DRec {
    uint32_t func, cdf, marginal_func, marginal_cdf;  // offsets into SceneView::dist_data
    Float marginal_int;
    uint32_t pad[3];
};
struct ImageLightRec {
    Float render_from_light[16];
    Float light_from_render[16];
    uint32_t image_level;  // ShmImageLevel index: res x res, 3 channels
    uint32_t n;            // res
    uint32_t pad[2];
    Dist2DRec distribution, compensated;
};

SHM_HD V3 ld3(const Float* p) { return v3(p[0], p[1], p[2]); }

// Triangle::get_points + mesh attribute fetch (shape/triangle.rs:148-160, 311-320)
SHM_HD TriangleData load_triangle_rec(const SceneView& sv, const PrimRec& pr) {
    TriangleData t;
    t.p0 = ld3(pr.p0);
    t.p1 = ld3(pr.p1);
    t.p2 = ld3(pr.p2);
    uint32_t f = sv.mesh_flags[pr.mesh];
    t.has_n = (f & MESH_HAS_N) != 0;
    t.has_s = (f & MESH_HAS_S) != 0;
    t.has_uv = (f & MESH_HAS_UV) != 0;
    t.flip = (f & MESH_FLIP) != 0;
    t.uv0 = t.uv1 = t.uv2 = v2(0.0f, 0.0f);
    t.n0 = t.n1 = t.n2 = v3s(0.0f);
    t.s0 = t.s1 = t.s2 = v3s(0.0f);
    if (f & (MESH_HAS_N | MESH_HAS_S | MESH_HAS_UV)) {
        uint32_t i0 = sv.vi[3 * pr.tri], i1 = sv.vi[3 * pr.tri + 1], i2 = sv.vi[3 * pr.tri + 2];
        if (t.has_n) { t.n0 = ld3(sv.vn + 3 * i0); t.n1 = ld3(sv.vn + 3 * i1); t.n2 = ld3(sv.vn + 3 * i2); }
        if (t.has_s) { t.s0 = ld3(sv.vs + 3 * i0); t.s1 = ld3(sv.vs + 3 * i1); t.s2 = ld3(sv.vs + 3 * i2); }
        if (t.has_uv) {
            t.uv0 = v2(sv.vuv[2 * i0], sv.vuv[2 * i0 + 1]);
            t.uv1 = v2(sv.vuv[2 * i1], sv.vuv[2 * i1 + 1]);
            t.uv2 = v2(sv.vuv[2 * i2], sv.vuv[2 * i2 + 1]);
        }
    }
    return t;
}
SHM_HD TriangleData load_triangle(const SceneView& sv, uint32_t slot) { return load_triangle_rec(sv, sv.prim_recs[slot]); }
// the emitter's record of an area light: the per-light copy when the scene has one (same bytes). `light` must be an ELEMENT of the table sv.lights names — the global
// one or its LDS-staged copy —: its position there is the index (checked on the host, where the oracle and the tests run this code; a light held by value would index garbage)
SHM_HD const PrimRec& light_prim_rec(const SceneView& sv, const ShmLight& light) {
#if !defined(__HIP_DEVICE_COMPILE__)
    assert(&light >= sv.lights && &light < sv.lights + sv.n_lights);
#endif
    return sv.light_prim_recs ? sv.light_prim_recs[&light - sv.lights] : sv.prim_recs[light.primitive];
}

// BilinearPatch::get_points (bilinear_patch.rs:87-98) + the constants fixed at scene creation
SHM_HD PatchData load_patch_rec(const SceneView& sv, const PrimRec& pr) {
    const PatchExtra& px = sv.patches[pr.kind_index & PRIM_INDEX_MASK];
    PatchData pd;
    pd.p00 = ld3(pr.p0); pd.p10 = ld3(pr.p1); pd.p01 = ld3(pr.p2); pd.p11 = ld3(px.p11);
    pd.is_rect = (px.flags & 1u) != 0;
    pd.flip = (px.flags & 2u) != 0;
    pd.area = px.area;
    pd.has_n = (px.flags & 4u) != 0;
    pd.has_uv = (px.flags & 8u) != 0;
    pd.n00 = pd.n10 = pd.n01 = pd.n11 = v3s(0.0f);
    pd.uv00 = pd.uv10 = pd.uv01 = pd.uv11 = v2(0.0f, 0.0f);
    if (px.flags & 12u) {
        const uint32_t* vi = sv.patch_vi + 4u * (pr.kind_index & PRIM_INDEX_MASK);
        if (pd.has_n) { pd.n00 = ld3(sv.patch_vn + 3 * vi[0]); pd.n10 = ld3(sv.patch_vn + 3 * vi[1]); pd.n01 = ld3(sv.patch_vn + 3 * vi[2]); pd.n11 = ld3(sv.patch_vn + 3 * vi[3]); }
        if (pd.has_uv) {
            pd.uv00 = v2(sv.patch_vuv[2 * vi[0]], sv.patch_vuv[2 * vi[0] + 1]);
            pd.uv10 = v2(sv.patch_vuv[2 * vi[1]], sv.patch_vuv[2 * vi[1] + 1]);
            pd.uv01 = v2(sv.patch_vuv[2 * vi[2]], sv.patch_vuv[2 * vi[2] + 1]);
            pd.uv11 = v2(sv.patch_vuv[2 * vi[3]], sv.patch_vuv[2 * vi[3] + 1]);
        }
    }
    return pd;
}

SHM_HD PatchData load_patch(const SceneView& sv, uint32_t slot) { return load_patch_rec(sv, sv.prim_recs[slot]); }

// Result of BvhAggregate::intersect reduced to identifying data (see ShmHit).
struct Hit {
    int32_t prim;  // leaf-order slot, -1 = miss
    Float t;
    Float b0, b1, b2;  // triangle barycentrics | sphere p_obj | bilinear patch (u, v, -)
    Float phi;         // sphere
    int32_t inst = -1; // leaf-order slot of the TransformedPrimitive the hit was found through (prim, t, b* in its space), -1: none
};

// Transform::apply_ray_inverse (transform.rs:700-722) into an instance's space: the origin picks up the transform's rounding error
// (apply_inverse(Point3fi::from(o)), transform.rs:631-698 = xf_point_i with m_inv), is pushed along d to the edge of that error
// box, and t_max shrinks by the same dt.
SHM_HD Ray xf_ray_inverse(const Float* minv, V3 ro, V3 rd, Float& t_max) {
    P3i o = xf_point_i(minv, p3i_exact(ro));
    V3 d = xf_vector(minv, rd);
    Float ls = length_squared(d);
    if (ls > 0.0f) {
        V3 o_error = v3(o.x.width() / 2.0f, o.y.width() / 2.0f, o.z.width() / 2.0f);
        Float dt = dot(abs3(d), o_error) / ls;
        t_max = t_max - dt;
        o = o + p3i_exact(d * dt);
    }
    Ray r;
    r.o = o.mid();
    r.d = d;
    return r;
}

// Shape::intersect's interaction for the CLOSEST hit only (the reference builds it per accepted
// candidate, triangle.rs:529-535; the result for the surviving candidate is identical).
// TRI_ONLY: the caller knows the scene holds triangles only (the GPU shade kernel is instantiated per scene class so that the
// quadric / patch code does not cost registers where it cannot run); the general form is the default.
template <bool TRI_ONLY = false>
SHM_HD SurfaceInteraction hit_interaction_local(const SceneView& sv, const Hit& h, V3 wo);
// ... and for a hit found through a TransformedPrimitive (primitive.rs:158-171): the interaction is built in the instance's space
// with the instance-space -ray.d, then mapped by render_from_primitive.apply(SurfaceInteraction) (transform.rs:573-608, quirk 6)
template <bool TRI_ONLY = false>
SHM_HD SurfaceInteraction hit_interaction(const SceneView& sv, const Hit& h, V3 wo) {
    if (!TRI_ONLY && h.inst >= 0) {
        const ShmInstance& in = sv.instances[sv.prim_recs[h.inst].kind_index & PRIM_INDEX_MASK];
        V3 wo_local = xf_vector(in.primitive_from_render, wo);  // -(m_inv d) == m_inv (-d) exactly
        return xf_surface_interaction(in.render_from_primitive, in.primitive_from_render, hit_interaction_local<false>(sv, h, wo_local), sv.quirks_off != 0);
    }
    return hit_interaction_local<TRI_ONLY>(sv, h, wo);
}
template <bool TRI_ONLY>
SHM_HD SurfaceInteraction hit_interaction_local(const SceneView& sv, const Hit& h, V3 wo) {
    const PrimRec& pr = sv.prim_recs[h.prim];
    if (!TRI_ONLY && (pr.kind_index & PRIM_SPHERE_BIT)) {
        QuadricIntersection qi;
        qi.t_hit = h.t;
        qi.p_obj = v3(h.b0, h.b1, h.b2);
        qi.phi = h.phi;
        return sphere_interaction(sv.spheres[pr.kind_index & PRIM_INDEX_MASK], qi, wo, sv.quirks_off != 0);
    }
    if (!TRI_ONLY && (pr.kind_index & PRIM_PATCH_BIT)) return blp_interaction(load_patch(sv, (uint32_t)h.prim), h.b0, h.b1, wo);
    TriangleData tr = load_triangle(sv, (uint32_t)h.prim);
    TriangleIntersection ti;
    ti.b0 = h.b0; ti.b1 = h.b1; ti.b2 = h.b2; ti.t = h.t;
    return triangle_interaction(tr, ti, wo);
}

// Primitive::intersect for one leaf-order slot (primitive.rs:95-103 -> shape/shape.rs:164-170).
// Updates hit/t_max on acceptance exactly as aggregate.rs:101-110 does.
SHM_HD bool prim_intersect(const SceneView& sv, uint32_t slot, V3 ro, V3 rd, Float t_max, Hit& h) {
    const PrimRec& pr = sv.prim_recs[slot];
    if (pr.kind_index & PRIM_SPHERE_BIT) {
        QuadricIntersection qi;
        if (!sphere_basic_intersect(sv.spheres[pr.kind_index & PRIM_INDEX_MASK], ro, rd, t_max, qi)) return false;
        h.prim = (int32_t)slot; h.t = qi.t_hit; h.b0 = qi.p_obj.x; h.b1 = qi.p_obj.y; h.b2 = qi.p_obj.z; h.phi = qi.phi;
        return true;
    }
    if (pr.kind_index & PRIM_PATCH_BIT) {
        BilinearIntersection bi;
        if (!blp_intersect(ro, rd, t_max, ld3(pr.p0), ld3(pr.p1), ld3(pr.p2), ld3(sv.patches[pr.kind_index & PRIM_INDEX_MASK].p11), bi)) return false;
        h.prim = (int32_t)slot; h.t = bi.t; h.b0 = bi.u; h.b1 = bi.v; h.b2 = 0.0f; h.phi = 0.0f;
        return true;
    }
    TriangleIntersection ti;
    if (!intersect_triangle(ro, rd, t_max, ld3(pr.p0), ld3(pr.p1), ld3(pr.p2), ti)) return false;
    h.prim = (int32_t)slot; h.t = ti.t; h.b0 = ti.b0; h.b1 = ti.b1; h.b2 = ti.b2; h.phi = 0.0f;
    return true;
}

}  // namespace shm
