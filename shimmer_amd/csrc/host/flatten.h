// host/flatten.h — validate a ShmSceneDesc and marshal it into the flat arrays of shm::SceneView
// (host memory). libshimmer_hip uploads these arrays to HBM; the CPU oracle reads them in place.
// This is data marshalling only: no arithmetic of the hot path lives here.
#pragma once
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <string>
#include <utility>
#include <vector>

#include "../shm/scene.h"

static_assert(sizeof(ShmMaterial) == 240 && sizeof(ShmFloatTexture) == 48 && sizeof(ShmSpectrumTexture) == 64 && sizeof(ShmSpectrum) == 32 && sizeof(ShmBvhNode) == 32 && sizeof(ShmPrimitive) == 16,
              "POD layouts of include/shimmer_hip.h (mirrored by shimmer_amd/abi.py)");
namespace shm_host {

struct FlatScene {
    std::vector<ShmBvhNode> nodes;
    std::vector<shm::PrimRec> prim_recs;
    std::vector<ShmPrimitive> primitives;
    std::vector<uint32_t> mesh_flags;
    std::vector<uint32_t> vi;
    std::vector<float> vn, vs, vuv;
    std::vector<ShmSphere> spheres;
    std::vector<shm::PatchExtra> patches;
    std::vector<uint32_t> patch_vi;
    std::vector<float> patch_vn, patch_vuv;
    std::vector<ShmMaterial> materials;
    std::vector<ShmLight> lights;
    std::vector<shm::PrimRec> light_prim_recs;  // per light: the emitter's record (area lights; zeros otherwise): SceneView::light_prim_recs
    std::vector<uint32_t> infinite_lights;
    std::vector<float> spectrum_data;
    std::vector<float> sensor_r, sensor_g, sensor_b;
    ShmCamera camera;
    ShmFilm film;
    uint32_t max_leaf_depth = 0;  // deepest leaf (root = 0); bounds the traversal stack
    float scene_radius = 0.0f;
    uint64_t n_quadric_patch_prims = 0;  // primitive records that are spheres or bilinear patches (instances not counted): a scene where they are MANY runs the traversal
                                         // kernels' five-wave instantiations (k_trace.hip, K5_GEN_HEAVY_WAVES: the parked round's registers instead of its spill code)
    bool has_spheres = false;  // any non-triangle shape (sphere or bilinear patch): selects k_trace5<.., GEN = true> and the general-geometry shading instantiations
    bool has_layered = false;  // any Coated* material: selects the k_shade instantiation that carries LayeredBxDF
    bool has_rough_dielectric = false;  // a DielectricMaterial whose roughness is not the constant 0 (the specular / rough split of its scatter kernels)
    bool has_class[4] = {false, false, false, false};  // BxDF classes present in the material table (staged shading launches one scatter
                                                       // kernel per class): 0 diffuse, 1 conductor, 2 dielectric / thin dielectric, 3 coated
    bool diffuse_only = true;  // every material is a DiffuseMaterial: selects the k_shade instantiation with the other BxDFs compiled out
    // image textures (ABI v6)
    std::vector<ShmImageTexture> image_textures;
    std::vector<ShmImageLevel> image_levels;
    std::vector<float> texel_data;
    std::vector<float> rgb2spec_scale, rgb2spec_data, cs_illuminant, ewa_lut;
    uint32_t rgb2spec_res = 0;
    bool has_textures = false;  // a material slot binds an image texture (the path carries ray differentials) or an image infinite
                                // light exists (both read the colour-space tables): selects k_shade<.., HAS_TEX>
    bool has_material_textures = false, has_image_light = false;
    double area_plain_diffuse = 0.0, area_total = 0.0;  // surface area of those primitives / of all (instances' placements not counted): the split pass's second criterion
    uint64_t n_plain_diffuse_prims = 0;  // primitives whose material is a DiffuseMaterial that binds no texture (the split pass of textured scenes, render.hip)  // ... which of the two (round 5: an image light alone needs no differentials — render.hip, env_lean)
    std::vector<ShmInstance> instances;
    bool has_instances = false;
    std::vector<ShmFloatTexture> float_textures;
    std::vector<shm::FloatTexRange> ftex_ranges;
    std::vector<shm::FloatTexOp> ftex_ops;
    std::vector<ShmSpectrumTexture> spectrum_textures;
    std::vector<shm::FloatTexRange> stex_ranges;
    std::vector<shm::FloatTexOp> stex_ops;
    std::vector<shm::ImageLightRec> image_lights;
    std::vector<float> dist_data;

    shm::SceneView view() const {
        shm::SceneView v;
        v.quirks_off = 0;
        v.call_copy = nullptr;  // (device only: set per workgroup by the kernels that evaluate textures)
        v.inst_roots = nullptr;  // (device only)
        v.nodes = nodes.data();
        v.n_nodes = (uint32_t)nodes.size();
        v.prim_recs = prim_recs.data();
        v.primitives = primitives.data();
        v.n_primitives = (uint32_t)primitives.size();
        v.mesh_flags = mesh_flags.data();
        v.vi = vi.data();
        v.vn = vn.data();
        v.vs = vs.data();
        v.vuv = vuv.data();
        v.spheres = spheres.data();
        v.patches = patches.data();
        v.patch_vi = patch_vi.data();
        v.patch_vn = patch_vn.data();
        v.patch_vuv = patch_vuv.data();
        v.materials = materials.data();
        v.lights = lights.data();
        v.n_lights = (uint32_t)lights.size();
        v.light_prim_recs = light_prim_recs.empty() ? nullptr : light_prim_recs.data();
        v.infinite_lights = infinite_lights.data();
        v.n_infinite_lights = (uint32_t)infinite_lights.size();
        v.spectrum_data = spectrum_data.data();
        v.scene_radius = scene_radius;
        v.camera = camera;
        for (int i = 0; i < 4; ++i) v.pixel_bounds[i] = film.pixel_bounds[i];
        v.filter_radius[0] = film.filter_radius[0];
        v.filter_radius[1] = film.filter_radius[1];
        v.imaging_ratio = film.imaging_ratio;
        v.max_component_value = film.max_component_value;
        v.sensor_r_bar = sensor_r.data();
        v.sensor_g_bar = sensor_g.data();
        v.sensor_b_bar = sensor_b.data();
        v.image_textures = image_textures.data();
        v.image_levels = image_levels.data();
        v.texel_data = texel_data.data();
        v.rgb2spec_res = rgb2spec_res;
        v.rgb2spec_scale = rgb2spec_scale.data();
        v.rgb2spec_data = rgb2spec_data.data();
        v.cs_illuminant = cs_illuminant.data();
        v.ewa_lut = ewa_lut.data();
        v.instances = instances.data();
        v.float_textures = float_textures.data();
        v.ftex_ranges = ftex_ranges.data();
        v.ftex_ops = ftex_ops.data();
        v.spectrum_textures = spectrum_textures.data();
        v.stex_ranges = stex_ranges.data();
        v.stex_ops = stex_ops.data();
        v.image_lights = image_lights.data();
        v.dist_data = dist_data.data();
        return v;
    }
};

// n_textures > 0 only for the SpectrumTexture slots of a material (a, b, c): eta and light spectra are plain Spectrum values
inline bool check_spectrum(const ShmSpectrum& s, uint32_t n_floats, std::string& err, uint32_t n_textures = 0, bool* is_texture = nullptr) {
    if (s.kind == SHM_SPECTRUM_TEXTURE_NODE) {
        // n_textures > 0 marks a SpectrumTexture slot here too; the node index is checked by the caller (it knows the table)
        if (!is_texture) { err = "a spectrum texture is not valid in this slot"; return false; }
        *is_texture = true;
        return true;
    }
    if (s.kind == SHM_SPECTRUM_IMAGE_TEXTURE) {
        if (s.offset >= n_textures) { err = n_textures ? "image texture index out of range" : "an image texture is not valid in this slot"; return false; }
        if (is_texture) *is_texture = true;
        return true;
    }
    if (s.kind == SHM_SPECTRUM_CONSTANT || s.kind == SHM_SPECTRUM_RGB_ALBEDO || s.kind == SHM_SPECTRUM_RGB_UNBOUNDED) return true;
    if (s.kind == SHM_SPECTRUM_DENSE || s.kind == SHM_SPECTRUM_RGB_ILLUMINANT) {
        if ((uint64_t)s.offset + s.n > n_floats) { err = "dense spectrum out of range"; return false; }
        return true;
    }
    if (s.kind == SHM_SPECTRUM_PIECEWISE_LINEAR) {
        if (s.n < 2 || (uint64_t)s.offset + 2ull * s.n > n_floats) { err = "piecewise spectrum out of range"; return false; }
        return true;
    }
    err = "unknown spectrum kind";
    return false;
}

// PiecewiseConstant1D::new_bounded(f, 0, 1) (sampling.rs:27-61) appended to `pool`: func (|f|) at func_off, cdf (n + 1) at cdf_off;
// returns func_int. f32 arithmetic in the reference's order.
inline float build_pc1d(const float* f, size_t n, std::vector<float>& pool, size_t func_off, size_t cdf_off) {
    for (size_t i = 0; i < n; ++i) pool[func_off + i] = f[i] < 0.0f ? -f[i] : f[i];
    pool[cdf_off] = 0.0f;
    for (size_t i = 1; i <= n; ++i) pool[cdf_off + i] = pool[cdf_off + i - 1] + pool[func_off + i - 1] * (1.0f - 0.0f) / (float)n;
    const float func_int = pool[cdf_off + n];
    if (func_int == 0.0f) for (size_t i = 1; i <= n; ++i) pool[cdf_off + i] = (float)i / (float)n;
    else for (size_t i = 1; i <= n; ++i) pool[cdf_off + i] /= func_int;
    return func_int;
}
// PiecewiseConstant2D::new(func, n, n, [0,1]^2) (sampling.rs:124-153)
inline shm::Dist2DRec build_pc2d(const std::vector<float>& d, size_t n, std::vector<float>& pool) {
    shm::Dist2DRec r{};
    size_t base = pool.size();
    pool.resize(base + n * n + n * (n + 1) + n + (n + 1));
    r.func = (uint32_t)base;
    r.cdf = (uint32_t)(base + n * n);
    r.marginal_func = (uint32_t)(base + n * n + n * (n + 1));
    r.marginal_cdf = r.marginal_func + (uint32_t)n;
    std::vector<float> integrals(n);
    for (size_t v = 0; v < n; ++v) integrals[v] = build_pc1d(d.data() + v * n, n, pool, r.func + v * n, r.cdf + v * (n + 1));
    r.marginal_int = build_pc1d(integrals.data(), n, pool, r.marginal_func, r.marginal_cdf);
    return r;
}

// Returns 0 or a negative ShmError; err receives a message.
inline int flatten_scene(const ShmSceneDesc* d, FlatScene& out, std::string& err) {
    if (!d) { err = "null scene description"; return SHM_ERR_INVALID_ARGUMENT; }
    if (d->abi_version != SHM_ABI_VERSION) { err = "abi_version mismatch"; return SHM_ERR_INVALID_ARGUMENT; }
    if (d->n_nodes == 0 || !d->nodes) { err = "scene has no BVH nodes"; return SHM_ERR_INVALID_ARGUMENT; }
    if (d->n_primitives == 0 || !d->primitives) { err = "scene has no primitives"; return SHM_ERR_INVALID_ARGUMENT; }
    if (d->n_materials == 0 || !d->materials) { err = "scene has no materials"; return SHM_ERR_INVALID_ARGUMENT; }
    if (!d->film.sensor_r_bar || !d->film.sensor_g_bar || !d->film.sensor_b_bar) { err = "film sensor tables missing"; return SHM_ERR_INVALID_ARGUMENT; }
    if (d->film.pixel_bounds[2] <= d->film.pixel_bounds[0] || d->film.pixel_bounds[3] <= d->film.pixel_bounds[1]) { err = "empty pixel bounds"; return SHM_ERR_INVALID_ARGUMENT; }

    out.nodes.assign(d->nodes, d->nodes + d->n_nodes);
    out.primitives.assign(d->primitives, d->primitives + d->n_primitives);
    out.spheres.assign(d->spheres, d->spheres + (d->spheres ? d->n_spheres : 0));
    out.materials.assign(d->materials, d->materials + d->n_materials);
    out.lights.assign(d->lights, d->lights + (d->lights ? d->n_lights : 0));
    out.spectrum_data.assign(d->spectrum_data, d->spectrum_data + (d->spectrum_data ? d->n_spectrum_floats : 0));
    if (out.spectrum_data.empty()) out.spectrum_data.push_back(0.0f);
    out.sensor_r.assign(d->film.sensor_r_bar, d->film.sensor_r_bar + 471);
    out.sensor_g.assign(d->film.sensor_g_bar, d->film.sensor_g_bar + 471);
    out.sensor_b.assign(d->film.sensor_b_bar, d->film.sensor_b_bar + 471);
    out.camera = d->camera;
    out.film = d->film;
    out.film.sensor_r_bar = out.film.sensor_g_bar = out.film.sensor_b_bar = nullptr;

    // meshes -> global arrays
    std::vector<uint32_t> tri_base(d->n_meshes + 1, 0), vert_base(d->n_meshes + 1, 0);
    for (uint32_t m = 0; m < d->n_meshes; ++m) {
        const ShmTriangleMesh& mesh = d->meshes[m];
        if (!mesh.vertex_indices || !mesh.p) { err = "mesh without indices/positions"; return SHM_ERR_INVALID_ARGUMENT; }
        tri_base[m + 1] = tri_base[m] + mesh.n_triangles;
        vert_base[m + 1] = vert_base[m] + mesh.n_vertices;
    }
    uint32_t n_tris = tri_base[d->n_meshes], n_verts = vert_base[d->n_meshes];
    bool any_attr = false;
    out.mesh_flags.resize(d->n_meshes ? d->n_meshes : 1, 0);
    for (uint32_t m = 0; m < d->n_meshes; ++m) {
        const ShmTriangleMesh& mesh = d->meshes[m];
        uint32_t f = 0;
        if (mesh.n) f |= shm::MESH_HAS_N;
        if (mesh.s) f |= shm::MESH_HAS_S;
        if (mesh.uv) f |= shm::MESH_HAS_UV;
        if ((mesh.reverse_orientation != 0) ^ (mesh.transform_swaps_handedness != 0)) f |= shm::MESH_FLIP;
        out.mesh_flags[m] = f;
        if (f & (shm::MESH_HAS_N | shm::MESH_HAS_S | shm::MESH_HAS_UV)) any_attr = true;
    }
    if (any_attr) {
        out.vi.resize(3ull * n_tris);
        out.vn.assign(3ull * n_verts, 0.0f);
        out.vs.assign(3ull * n_verts, 0.0f);
        out.vuv.assign(2ull * n_verts, 0.0f);
        for (uint32_t m = 0; m < d->n_meshes; ++m) {
            const ShmTriangleMesh& mesh = d->meshes[m];
            for (uint64_t i = 0; i < 3ull * mesh.n_triangles; ++i)
                out.vi[3ull * tri_base[m] + i] = mesh.vertex_indices[i] + vert_base[m];
            if (mesh.n) std::copy(mesh.n, mesh.n + 3ull * mesh.n_vertices, out.vn.begin() + 3ull * vert_base[m]);
            if (mesh.s) std::copy(mesh.s, mesh.s + 3ull * mesh.n_vertices, out.vs.begin() + 3ull * vert_base[m]);
            if (mesh.uv) std::copy(mesh.uv, mesh.uv + 2ull * mesh.n_vertices, out.vuv.begin() + 2ull * vert_base[m]);
        }
    } else {
        out.vi.assign(3, 0);
        out.vn.assign(3, 0.0f);
        out.vs.assign(3, 0.0f);
        out.vuv.assign(2, 0.0f);
    }

    // bilinear patch meshes (per-vertex n / uv into global arrays, like the triangle meshes' vi / vn / vuv)
    const uint32_t n_pm = d->patch_meshes ? d->n_patch_meshes : 0;
    std::vector<uint32_t> patch_base(n_pm + 1, 0), patch_vert_base(n_pm + 1, 0);
    bool any_patch_attr = false;
    for (uint32_t m = 0; m < n_pm; ++m) {
        const ShmBilinearPatchMesh& mesh = d->patch_meshes[m];
        if (!mesh.vertex_indices || !mesh.p) { err = "patch mesh without indices/positions"; return SHM_ERR_INVALID_ARGUMENT; }
        patch_base[m + 1] = patch_base[m] + mesh.n_patches;
        patch_vert_base[m + 1] = patch_vert_base[m] + mesh.n_vertices;
        if (mesh.n || mesh.uv) any_patch_attr = true;
    }
    const uint32_t n_patches = patch_base[n_pm];
    if (n_patches > shm::PRIM_INDEX_MASK) { err = "too many bilinear patches"; return SHM_ERR_INVALID_ARGUMENT; }
    out.patches.assign(std::max<uint32_t>(n_patches, 1), shm::PatchExtra{});
    if (any_patch_attr) {
        out.patch_vi.assign(4ull * n_patches, 0);
        out.patch_vn.assign(3ull * patch_vert_base[n_pm], 0.0f);
        out.patch_vuv.assign(2ull * patch_vert_base[n_pm], 0.0f);
        for (uint32_t m = 0; m < n_pm; ++m) {
            const ShmBilinearPatchMesh& mesh = d->patch_meshes[m];
            for (uint64_t i = 0; i < 4ull * mesh.n_patches; ++i) {
                if (mesh.vertex_indices[i] >= mesh.n_vertices) { err = "vertex index out of range"; return SHM_ERR_INVALID_ARGUMENT; }
                out.patch_vi[4ull * patch_base[m] + i] = mesh.vertex_indices[i] + patch_vert_base[m];
            }
            if (mesh.n) std::copy(mesh.n, mesh.n + 3ull * mesh.n_vertices, out.patch_vn.begin() + 3ull * patch_vert_base[m]);
            if (mesh.uv) std::copy(mesh.uv, mesh.uv + 2ull * mesh.n_vertices, out.patch_vuv.begin() + 2ull * patch_vert_base[m]);
        }
    } else {
        out.patch_vi.assign(4, 0);
        out.patch_vn.assign(3, 0.0f);
        out.patch_vuv.assign(2, 0.0f);
    }

    // 64-B leaf-order records
    out.prim_recs.resize(d->n_primitives);
    for (uint32_t s = 0; s < d->n_primitives; ++s) {
        const ShmPrimitive& pr = d->primitives[s];
        shm::PrimRec rec;
        memset(&rec, 0, sizeof(rec));
        // GeometricPrimitive::material == None marks an interface between media: get_bsdf returns None and li() continues the ray with
        // skip_intersection (interaction.rs:199-202, 410-427; integrator.rs:816-822). Media are todo!() in the reference, so no scene
        // reaches that branch there; here the index is plain and the all-ones value is named and rejected rather than misread.
        if (pr.material == 0xffffffffu) { err = "primitive without a material (medium interface, skip_intersection) is not supported"; return SHM_ERR_UNSUPPORTED; }
        if (pr.material >= d->n_materials) { err = "primitive material out of range"; return SHM_ERR_INVALID_ARGUMENT; }
        if (pr.area_light >= (int32_t)d->n_lights) { err = "primitive area light out of range"; return SHM_ERR_INVALID_ARGUMENT; }
        rec.material = pr.material;
        rec.area_light = pr.area_light;
        if (pr.shape_kind == SHM_SHAPE_SPHERE) {
            if (pr.shape_index >= d->n_spheres) { err = "sphere index out of range"; return SHM_ERR_INVALID_ARGUMENT; }
            out.n_quadric_patch_prims += 1u;
            rec.kind_index = shm::PRIM_SPHERE_BIT | pr.shape_index;
            out.has_spheres = true;
        } else if (pr.shape_kind == SHM_SHAPE_TRIANGLE) {
            if (pr.shape_index >= n_tris) { err = "triangle index out of range"; return SHM_ERR_INVALID_ARGUMENT; }
            uint32_t m = (uint32_t)(std::upper_bound(tri_base.begin(), tri_base.end(), pr.shape_index) - tri_base.begin()) - 1;
            const ShmTriangleMesh& mesh = d->meshes[m];
            uint32_t local = pr.shape_index - tri_base[m];
            for (int k = 0; k < 3; ++k) {
                uint32_t vidx = mesh.vertex_indices[3ull * local + k];
                if (vidx >= mesh.n_vertices) { err = "vertex index out of range"; return SHM_ERR_INVALID_ARGUMENT; }
                float* dst = (k == 0) ? rec.p0 : (k == 1 ? rec.p1 : rec.p2);
                dst[0] = mesh.p[3ull * vidx];
                dst[1] = mesh.p[3ull * vidx + 1];
                dst[2] = mesh.p[3ull * vidx + 2];
            }
            rec.kind_index = shm::triangle_is_degenerate(shm::ld3(rec.p0), shm::ld3(rec.p1), shm::ld3(rec.p2)) ? shm::PRIM_DEGENERATE_BIT : 0u;
            rec.mesh = m;
            rec.tri = pr.shape_index;
        } else if (pr.shape_kind == SHM_SHAPE_BILINEAR_PATCH) {
            if (pr.shape_index >= n_patches) { err = "bilinear patch index out of range"; return SHM_ERR_INVALID_ARGUMENT; }
            uint32_t m = (uint32_t)(std::upper_bound(patch_base.begin(), patch_base.end(), pr.shape_index) - patch_base.begin()) - 1;
            const ShmBilinearPatchMesh& mesh = d->patch_meshes[m];
            uint32_t local = pr.shape_index - patch_base[m];
            shm::V3 c[4];
            for (int k = 0; k < 4; ++k) {
                uint32_t vidx = mesh.vertex_indices[4ull * local + k];
                if (vidx >= mesh.n_vertices) { err = "vertex index out of range"; return SHM_ERR_INVALID_ARGUMENT; }
                c[k] = shm::v3(mesh.p[3ull * vidx], mesh.p[3ull * vidx + 1], mesh.p[3ull * vidx + 2]);
            }
            rec.p0[0] = c[0].x; rec.p0[1] = c[0].y; rec.p0[2] = c[0].z;  // p00
            rec.p1[0] = c[1].x; rec.p1[1] = c[1].y; rec.p1[2] = c[1].z;  // p10
            rec.p2[0] = c[2].x; rec.p2[1] = c[2].y; rec.p2[2] = c[2].z;  // p01
            shm::PatchExtra& px = out.patches[pr.shape_index];
            px.p11[0] = c[3].x; px.p11[1] = c[3].y; px.p11[2] = c[3].z;
            // ... and in the record's own 64 bytes, in the words a patch does not use (mesh, tri, pad[0]): the traversal's patch test reads the fourth corner from the line it
            // already fetched instead of through a dependent gather from `patches` (k_trace.hip, the parked round; round 6)
            memcpy(&rec.mesh, &c[3].x, 4); memcpy(&rec.tri, &c[3].y, 4); memcpy(&rec.pad[0], &c[3].z, 4);
            // BilinearPatch::new / is_rectangle (bilinear_patch.rs:40-69, 108-142), evaluated once, with the shared arithmetic
            bool is_rect = shm::blp_is_rectangle(c[0], c[1], c[2], c[3]);
            px.area = shm::blp_area(c[0], c[1], c[2], c[3], is_rect);
            px.flags = (is_rect ? 1u : 0u) | ((((mesh.reverse_orientation != 0) ^ (mesh.transform_swaps_handedness != 0)) ? 2u : 0u)) |
                       (mesh.n ? 4u : 0u) | (mesh.uv ? 8u : 0u);
            rec.kind_index = shm::PRIM_PATCH_BIT | pr.shape_index;
            out.n_quadric_patch_prims += 1u;
            out.has_spheres = true;
        } else if (pr.shape_kind == SHM_SHAPE_INSTANCE) {
            if (!d->instances || pr.shape_index >= d->n_instances) { err = "instance index out of range"; return SHM_ERR_INVALID_ARGUMENT; }
            rec.kind_index = shm::PRIM_INSTANCE_BIT | pr.shape_index;
            out.has_spheres = true;  // non-triangle code paths
            out.has_instances = true;
        } else {
            err = "unsupported shape kind";
            return SHM_ERR_UNSUPPORTED;
        }
        out.prim_recs[s] = rec;
    }

    // UniformInfiniteLight::preprocess (light.rs:799-803): bounding sphere of the scene bounds = the root node's box
    if (!out.nodes.empty()) {
        const ShmBvhNode& root = out.nodes[0];
        shm::V3 lo = shm::v3(root.bmin[0], root.bmin[1], root.bmin[2]), hi = shm::v3(root.bmax[0], root.bmax[1], root.bmax[2]);
        shm::V3 center = (lo + hi) / 2.0f;
        bool inside = center.x >= lo.x && center.x <= hi.x && center.y >= lo.y && center.y <= hi.y && center.z >= lo.z && center.z <= hi.z;
        out.scene_radius = inside ? shm::distance(center, hi) : 0.0f;
    }

    // image textures
    if (d->n_image_textures) {
        if (!d->image_textures || !d->image_levels || !d->texel_data) { err = "image texture arrays missing"; return SHM_ERR_INVALID_ARGUMENT; }
        if (d->n_texel_floats >= (1ull << 32)) { err = "more than 2^32 texel floats"; return SHM_ERR_UNSUPPORTED; }
        out.image_textures.assign(d->image_textures, d->image_textures + d->n_image_textures);
        out.image_levels.assign(d->image_levels, d->image_levels + d->n_image_levels);
        out.texel_data.assign(d->texel_data, d->texel_data + d->n_texel_floats);
        bool need_cs = false, need_illum = false, need_lut = false;
        for (const ShmImageTexture& t : out.image_textures) {
            if (t.mapping > SHM_TEXMAP_PLANAR || t.filter > SHM_TEXFILTER_EWA || t.wrap > SHM_WRAP_OCTAHEDRAL_SPHERE || t.spectrum_type > SHM_SPECTRUM_TYPE_ILLUMINANT) { err = "image texture: unknown mapping / filter / wrap / spectrum type"; return SHM_ERR_INVALID_ARGUMENT; }
            if (t.n_channels != 1 && t.n_channels != 3) { err = "image texture: n_channels must be 1 or 3"; return SHM_ERR_INVALID_ARGUMENT; }
            // the EWA footprint at its level spans about 2 * max_anisotropy texels per axis: bound the device loop
            if (!(t.max_anisotropy >= 0.0f && t.max_anisotropy <= 256.0f)) { err = "image texture: max_anisotropy must lie in [0, 256]"; return SHM_ERR_UNSUPPORTED; }
            if (t.n_levels == 0 || (uint64_t)t.first_level + t.n_levels > d->n_image_levels) { err = "image texture: level range out of bounds"; return SHM_ERR_INVALID_ARGUMENT; }
            for (uint32_t l = 0; l < t.n_levels; ++l) {
                const ShmImageLevel& lv = out.image_levels[t.first_level + l];
                if (lv.width <= 0 || lv.height <= 0 || (uint64_t)lv.texel_offset + (uint64_t)lv.width * (uint64_t)lv.height * t.n_channels > d->n_texel_floats) { err = "image texture: level texels out of bounds"; return SHM_ERR_INVALID_ARGUMENT; }
            }
            // MIPMap::filter returns texel(n_levels - 1, (0, 0)) for wide filters (mipmap.rs:165-168): Image::generate_pyramid ends in a 1x1 level
            need_cs |= t.has_color_space != 0;
            need_illum |= t.has_color_space && t.spectrum_type == SHM_SPECTRUM_TYPE_ILLUMINANT;
            need_lut |= t.filter == SHM_TEXFILTER_EWA;
        }
        const ShmColorSpace& cs = d->color_space;
        if (need_cs) {
            if (cs.rgb2spec_res < 2 || cs.rgb2spec_res > 256 || !cs.rgb2spec_scale || !cs.rgb2spec_data) { err = "image textures with a colour space need ShmSceneDesc::color_space (rgb2spec table)"; return SHM_ERR_INVALID_ARGUMENT; }
            out.rgb2spec_res = cs.rgb2spec_res;
            out.rgb2spec_scale.assign(cs.rgb2spec_scale, cs.rgb2spec_scale + cs.rgb2spec_res);
            size_t r = cs.rgb2spec_res;
            out.rgb2spec_data.assign(cs.rgb2spec_data, cs.rgb2spec_data + 9 * r * r * r);
        }
        if (need_illum) {
            if (!cs.illuminant) { err = "illuminant image textures need ShmColorSpace::illuminant"; return SHM_ERR_INVALID_ARGUMENT; }
            out.cs_illuminant.assign(cs.illuminant, cs.illuminant + 471);
        }
        if (need_lut) {
            if (!d->ewa_filter_lut) { err = "EWA-filtered image textures need ShmSceneDesc::ewa_filter_lut"; return SHM_ERR_INVALID_ARGUMENT; }
            out.ewa_lut.assign(d->ewa_filter_lut, d->ewa_filter_lut + 128);
        }
    }

    // ImageInfinitelight::new (light.rs:915-966): the two sampling distributions of each environment map
    if (d->n_image_lights) {
        if (!d->image_lights || !d->image_levels || !d->texel_data) { err = "image infinite light arrays missing"; return SHM_ERR_INVALID_ARGUMENT; }
        const ShmColorSpace& cs = d->color_space;
        if (cs.rgb2spec_res < 2 || cs.rgb2spec_res > 256 || !cs.rgb2spec_scale || !cs.rgb2spec_data || !cs.illuminant) { err = "an image infinite light needs ShmSceneDesc::color_space (rgb2spec table and illuminant)"; return SHM_ERR_INVALID_ARGUMENT; }
        if (out.image_levels.empty()) {
            if (d->n_texel_floats >= (1ull << 32)) { err = "more than 2^32 texel floats"; return SHM_ERR_UNSUPPORTED; }
            out.image_levels.assign(d->image_levels, d->image_levels + d->n_image_levels);
            out.texel_data.assign(d->texel_data, d->texel_data + d->n_texel_floats);
        }
        if (out.rgb2spec_res == 0) {
            out.rgb2spec_res = cs.rgb2spec_res;
            out.rgb2spec_scale.assign(cs.rgb2spec_scale, cs.rgb2spec_scale + cs.rgb2spec_res);
            size_t r = cs.rgb2spec_res;
            out.rgb2spec_data.assign(cs.rgb2spec_data, cs.rgb2spec_data + 9 * r * r * r);
        }
        if (out.cs_illuminant.empty()) out.cs_illuminant.assign(cs.illuminant, cs.illuminant + 471);
        for (uint32_t i = 0; i < d->n_image_lights; ++i) {
            const ShmImageInfiniteLight& il = d->image_lights[i];
            if (il.image_level >= d->n_image_levels) { err = "image infinite light: image level out of range"; return SHM_ERR_INVALID_ARGUMENT; }
            const ShmImageLevel& lv = out.image_levels[il.image_level];
            // light.rs:934-937: "Image resolution is non-square; it is unlikely that it is an environment map"
            if (lv.width <= 0 || lv.width != lv.height) { err = "image infinite light: the environment map must be square"; return SHM_ERR_INVALID_ARGUMENT; }
            const size_t n = (size_t)lv.width;
            if ((uint64_t)lv.texel_offset + 3ull * n * n > d->n_texel_floats) { err = "image infinite light: texels out of bounds"; return SHM_ERR_INVALID_ARGUMENT; }
            if (n > 8192) { err = "image infinite light: environment map larger than 8192^2"; return SHM_ERR_UNSUPPORTED; }
            shm::ImageLightRec rec{};
            for (int k = 0; k < 16; ++k) { rec.render_from_light[k] = il.render_from_light[k]; rec.light_from_render[k] = il.light_from_render[k]; }
            rec.image_level = il.image_level;
            rec.n = (uint32_t)n;
            // Image::get_default_sampling_distribution (image.rs:1379-1405): ImageChannelValues::average per pixel, dx_da = 1
            std::vector<float> dd(n * n);
            const float* t = out.texel_data.data() + lv.texel_offset;
            for (size_t k = 0; k < n * n; ++k) dd[k] = (((0.0f + t[3 * k]) + t[3 * k + 1]) + t[3 * k + 2]) / 3.0f;
            rec.distribution = build_pc2d(dd, n, out.dist_data);
            // light.rs:948-955: subtract the average, clamp at zero; all zero -> uniform
            float sum = 0.0f;
            for (float v : dd) sum += v;
            const float average = sum / (float)dd.size();
            bool all_zero = true;
            for (float& v : dd) { v = shm::max(v - average, 0.0f); all_zero &= (v == 0.0f); }
            if (all_zero) for (float& v : dd) v = 1.0f;
            rec.compensated = build_pc2d(dd, n, out.dist_data);
            out.image_lights.push_back(rec);
        }
        out.has_image_light = true;  // (has_textures: set where flatten_scene returns)
    }

    // FloatTexture node table (texture.rs:88-305): children precede parents, nesting <= 4 (float_texture_evaluate's bound)
    if (d->n_float_textures) {
        if (!d->float_textures) { err = "float texture array missing"; return SHM_ERR_INVALID_ARGUMENT; }
        out.float_textures.assign(d->float_textures, d->float_textures + d->n_float_textures);
        // per node its post-order evaluation program (children first; a child shared by two parents is evaluated once per use, as
        // the reference's recursion does)
        out.ftex_ranges.resize(d->n_float_textures);
        for (uint32_t i = 0; i < d->n_float_textures; ++i) {
            const ShmFloatTexture& t = out.float_textures[i];
            if (t.kind > SHM_FLOATTEX_IMAGE) { err = "unknown float texture kind"; return SHM_ERR_INVALID_ARGUMENT; }
            std::vector<shm::FloatTexOp> prog;
            auto append_child = [&](uint32_t k) -> int {  // copies the child's program, returns the slot of its root, or -1
                if (k >= i) return -1;
                const shm::FloatTexRange cr = out.ftex_ranges[k];
                const uint32_t base = (uint32_t)prog.size();
                for (uint32_t q = 0; q < cr.count; ++q) {
                    shm::FloatTexOp op = out.ftex_ops[cr.first + q];
                    op.a = (uint8_t)(op.a + base); op.b = (uint8_t)(op.b + base); op.c = (uint8_t)(op.c + base);
                    prog.push_back(op);
                }
                return (int)prog.size() - 1;
            };
            shm::FloatTexOp self{};
            self.node = i;
            if (t.kind == SHM_FLOATTEX_IMAGE) {
                if (t.image >= d->n_image_textures) { err = "float image texture: image index out of range"; return SHM_ERR_INVALID_ARGUMENT; }
            } else if (t.kind != SHM_FLOATTEX_CONSTANT) {
                int a = append_child(t.a), b = append_child(t.b), c = (t.kind == SHM_FLOATTEX_MIX) ? append_child(t.c) : 0;
                if (a < 0 || b < 0 || c < 0) { err = "float texture children must precede their parent"; return SHM_ERR_INVALID_ARGUMENT; }
                self.a = (uint8_t)a; self.b = (uint8_t)b; self.c = (uint8_t)c;
            }
            prog.push_back(self);
            if (prog.size() > (size_t)shm::FTEX_MAX_OPS) { err = "float texture tree larger than 32 nodes"; return SHM_ERR_UNSUPPORTED; }
            out.ftex_ranges[i].first = (uint32_t)out.ftex_ops.size();
            out.ftex_ranges[i].count = (uint32_t)prog.size();
            out.ftex_ops.insert(out.ftex_ops.end(), prog.begin(), prog.end());
        }
    }

    // SpectrumTexture node table: the same post-order programs (<= 8 ops)
    if (d->n_spectrum_textures) {
        if (!d->spectrum_textures) { err = "spectrum texture array missing"; return SHM_ERR_INVALID_ARGUMENT; }
        out.spectrum_textures.assign(d->spectrum_textures, d->spectrum_textures + d->n_spectrum_textures);
        out.stex_ranges.resize(d->n_spectrum_textures);
        const uint32_t nsf0 = (uint32_t)out.spectrum_data.size();
        for (uint32_t i = 0; i < d->n_spectrum_textures; ++i) {
            const ShmSpectrumTexture& t = out.spectrum_textures[i];
            if (t.kind > SHM_SPECTEX_DIRECTION_MIX) { err = "unknown spectrum texture kind"; return SHM_ERR_INVALID_ARGUMENT; }
            std::vector<shm::FloatTexOp> prog;
            auto append_child = [&](uint32_t k) -> int {
                if (k >= i) return -1;
                const shm::FloatTexRange cr = out.stex_ranges[k];
                const uint32_t base = (uint32_t)prog.size();
                for (uint32_t q = 0; q < cr.count; ++q) {
                    shm::FloatTexOp op = out.stex_ops[cr.first + q];
                    op.a = (uint8_t)(op.a + base); op.b = (uint8_t)(op.b + base);
                    prog.push_back(op);
                }
                return (int)prog.size() - 1;
            };
            shm::FloatTexOp self{};
            self.node = i;
            if (t.kind == SHM_SPECTEX_LEAF) {
                if (t.leaf.kind == SHM_SPECTRUM_TEXTURE_NODE) { err = "a spectrum texture leaf cannot be a texture node"; return SHM_ERR_INVALID_ARGUMENT; }
                bool is_tex = false;
                if (!check_spectrum(t.leaf, nsf0, err, d->n_image_textures, &is_tex)) return SHM_ERR_INVALID_ARGUMENT;
            } else {
                int a = append_child(t.a), b = (t.kind == SHM_SPECTEX_SCALED) ? 0 : append_child(t.b);
                if (a < 0 || b < 0) { err = "spectrum texture children must precede their parent"; return SHM_ERR_INVALID_ARGUMENT; }
                self.a = (uint8_t)a; self.b = (uint8_t)b;
                if (t.kind != SHM_SPECTEX_DIRECTION_MIX && t.f >= d->n_float_textures) { err = "spectrum texture: float texture index out of range"; return SHM_ERR_INVALID_ARGUMENT; }
            }
            prog.push_back(self);
            if (prog.size() > (size_t)shm::STEX_MAX_OPS) { err = "spectrum texture tree larger than 8 nodes"; return SHM_ERR_UNSUPPORTED; }
            out.stex_ranges[i].first = (uint32_t)out.stex_ops.size();
            out.stex_ranges[i].count = (uint32_t)prog.size();
            out.stex_ops.insert(out.stex_ops.end(), prog.begin(), prog.end());
        }
    }

    // materials / lights
    const uint32_t ntex = d->n_image_textures;
    uint32_t nsf = (uint32_t)out.spectrum_data.size();
    for (const ShmMaterial& m : out.materials) {
        if (m.kind > SHM_MATERIAL_MIX) { err = "unsupported material kind"; return SHM_ERR_UNSUPPORTED; }
        if (m.kind != SHM_MATERIAL_DIFFUSE) out.diffuse_only = false;
        if (m.kind == SHM_MATERIAL_DIFFUSE) out.has_class[0] = true;
        else if (m.kind == SHM_MATERIAL_CONDUCTOR) out.has_class[1] = true;
        else if (m.kind == SHM_MATERIAL_DIELECTRIC || m.kind == SHM_MATERIAL_THIN_DIELECTRIC) out.has_class[2] = true;
        else if (m.kind == SHM_MATERIAL_COATED_DIFFUSE || m.kind == SHM_MATERIAL_COATED_CONDUCTOR) out.has_class[3] = true;
        if (m.kind == SHM_MATERIAL_DIELECTRIC && (m.u_roughness != 0.0f || m.v_roughness != 0.0f || m.float_tex[SHM_FLOATSLOT_U_ROUGHNESS] != 0u ||
                                                  m.float_tex[SHM_FLOATSLOT_V_ROUGHNESS] != 0u))
            out.has_rough_dielectric = true;
        for (int k = 0; k < 8; ++k)
            if (m.float_tex[k] != 0u) {
                if (m.float_tex[k] > d->n_float_textures) { err = "material float texture index out of range"; return SHM_ERR_INVALID_ARGUMENT; }
                out.has_textures = true;
            }
        for (const ShmSpectrum* sp : {&m.a, &m.b, &m.c})
            if (sp->kind == SHM_SPECTRUM_TEXTURE_NODE && sp->offset >= d->n_spectrum_textures) { err = "material spectrum texture index out of range"; return SHM_ERR_INVALID_ARGUMENT; }
        if (m.normal_map != 0u) {
            if (m.normal_map > d->n_image_textures || out.image_textures[m.normal_map - 1].n_channels != 3) { err = "normal map: image texture index out of range or not RGB"; return SHM_ERR_INVALID_ARGUMENT; }
            out.has_textures = true;
        }
        if (m.kind == SHM_MATERIAL_MIX) {
            // both branches must reach a single material: follow every path with a step bound (a cycle never terminates)
            if (m.mix_material[0] >= d->n_materials || m.mix_material[1] >= d->n_materials) { err = "mix material index out of range"; return SHM_ERR_INVALID_ARGUMENT; }
            std::vector<std::pair<uint32_t, int>> todo{{m.mix_material[0], 1}, {m.mix_material[1], 1}};
            size_t steps = 0;
            while (!todo.empty()) {
                auto [j, depth] = todo.back();
                todo.pop_back();
                const ShmMaterial& c = out.materials[j];
                if (c.kind != SHM_MATERIAL_MIX) continue;
                if (depth >= 16 || ++steps > 65536) { err = "mix materials form a cycle or nest deeper than 16 levels"; return SHM_ERR_INVALID_ARGUMENT; }
                if (c.mix_material[0] >= d->n_materials || c.mix_material[1] >= d->n_materials) { err = "mix material index out of range"; return SHM_ERR_INVALID_ARGUMENT; }
                todo.push_back({c.mix_material[0], depth + 1});
                todo.push_back({c.mix_material[1], depth + 1});
            }
            continue;
        }
        // Dielectric / ThinDielectric keep eta in `a`: a Spectrum, not a texture (material.rs:520-527, 655-660)
        const bool a_is_texture_slot = !(m.kind == SHM_MATERIAL_DIELECTRIC || m.kind == SHM_MATERIAL_THIN_DIELECTRIC);
        if (!check_spectrum(m.a, nsf, err, a_is_texture_slot ? ntex : 0, a_is_texture_slot ? &out.has_textures : nullptr)) return SHM_ERR_INVALID_ARGUMENT;
        const bool coated = m.kind == SHM_MATERIAL_COATED_DIFFUSE || m.kind == SHM_MATERIAL_COATED_CONDUCTOR;
        if ((m.kind == SHM_MATERIAL_CONDUCTOR || (m.kind == SHM_MATERIAL_COATED_CONDUCTOR && !m.conductor_from_reflectance)) &&
            !check_spectrum(m.b, nsf, err, ntex, &out.has_textures))
            return SHM_ERR_INVALID_ARGUMENT;
        if (coated) {
            if (!check_spectrum(m.c, nsf, err, ntex, &out.has_textures) || !check_spectrum(m.d, nsf, err)) return SHM_ERR_INVALID_ARGUMENT;
            if (m.max_depth < 0 || m.max_depth > 1024 || m.n_samples < 1 || m.n_samples > 1024) { err = "coated material: max_depth / n_samples out of range"; return SHM_ERR_INVALID_ARGUMENT; }
            out.has_layered = true;
        }
    }
    for (uint32_t i = 0; i < out.lights.size(); ++i) {
        const ShmLight& l = out.lights[i];
        if (l.kind > SHM_LIGHT_IMAGE_INFINITE) { err = "unsupported light kind"; return SHM_ERR_UNSUPPORTED; }
        if (l.kind == SHM_LIGHT_IMAGE_INFINITE) {
            if (l.primitive >= d->n_image_lights) { err = "image infinite light index out of range"; return SHM_ERR_INVALID_ARGUMENT; }
            out.infinite_lights.push_back(i);
            continue;
        }
        if (l.spectrum.kind != SHM_SPECTRUM_DENSE) { err = "light spectrum must be densely sampled (light.rs:404,551,717)"; return SHM_ERR_INVALID_ARGUMENT; }
        if (!check_spectrum(l.spectrum, nsf, err)) return SHM_ERR_INVALID_ARGUMENT;
        if (l.kind == SHM_LIGHT_DIFFUSE_AREA && l.primitive >= d->n_primitives) { err = "area light primitive out of range"; return SHM_ERR_INVALID_ARGUMENT; }
        if (l.kind == SHM_LIGHT_UNIFORM_INFINITE) out.infinite_lights.push_back(i);
    }
    // the emitters' records beside the light table (same bytes as prim_recs[light.primitive]: a light sample starts from this small, cacheable / LDS-staged copy)
    out.light_prim_recs.assign(out.lights.size(), shm::PrimRec{});
    for (uint32_t i = 0; i < out.lights.size(); ++i)
        if (out.lights[i].kind == SHM_LIGHT_DIFFUSE_AREA) out.light_prim_recs[i] = out.prim_recs[out.lights[i].primitive];

    // BVH validation + depth (explicit stack; DFS order means child0 = i+1). The top-level tree starts at node 0, the tree of every
    // instanced object at its ShmInstance::root_node; a top-level leaf holding an instance continues into that tree (one more stack
    // entry for the way back).
    if (d->n_instances) out.instances.assign(d->instances, d->instances + d->n_instances);
    {
        struct Item { uint32_t node, depth; bool inner; };
        std::vector<Item> st;
        st.push_back({0u, 0u, false});
        uint64_t visited = 0;
        const uint64_t visit_cap = (uint64_t)d->n_nodes * (1ull + d->n_instances);
        while (!st.empty()) {
            auto [i, depth, inner] = st.back();
            st.pop_back();
            if (i >= d->n_nodes) { err = "BVH child index out of range"; return SHM_ERR_INVALID_ARGUMENT; }
            if (++visited > visit_cap) { err = "BVH is not a tree"; return SHM_ERR_INVALID_ARGUMENT; }
            const ShmBvhNode& n = d->nodes[i];
            if (n.n_prims > 0) {
                if ((uint64_t)n.offset + n.n_prims > d->n_primitives) { err = "BVH leaf range out of bounds"; return SHM_ERR_INVALID_ARGUMENT; }
                if (depth > out.max_leaf_depth) out.max_leaf_depth = depth;
                for (uint32_t k = 0; k < n.n_prims; ++k) {
                    const ShmPrimitive& pr = out.primitives[n.offset + k];
                    if (pr.shape_kind != SHM_SHAPE_INSTANCE) {
                        if (inner && pr.area_light >= 0) { err = "an instanced primitive cannot carry an area light"; return SHM_ERR_UNSUPPORTED; }
                        continue;
                    }
                    if (inner) { err = "an instanced primitive cannot itself be an instance"; return SHM_ERR_UNSUPPORTED; }
                    if (n.n_prims != 1) { err = "an instance must be alone in its BVH leaf"; return SHM_ERR_UNSUPPORTED; }
                    const ShmInstance& in = out.instances[pr.shape_index];
                    if (in.root_node == 0 || in.root_node >= d->n_nodes) { err = "instance root node out of range"; return SHM_ERR_INVALID_ARGUMENT; }
                    st.push_back({in.root_node, depth + 2, true});
                }
            } else {
                if (n.axis > 2) { err = "BVH axis out of range"; return SHM_ERR_INVALID_ARGUMENT; }
                if (n.offset <= i) { err = "BVH second child must follow its parent"; return SHM_ERR_INVALID_ARGUMENT; }
                st.push_back({n.offset, depth + 1, inner});
                st.push_back({i + 1, depth + 1, inner});
            }
        }
    }
    // The reference's traversal stack is [usize; 64] (aggregate.rs:90); deeper trees would index out of bounds there.
    if (getenv("SHM_DEBUG")) fprintf(stderr, "[shm] flatten: %u nodes, %u prims, max leaf depth %u\n", d->n_nodes, d->n_primitives, out.max_leaf_depth);
    if (out.max_leaf_depth >= 64) { err = "BVH deeper than the reference's 64-entry traversal stack"; return SHM_ERR_UNSUPPORTED; }
    // ShmMaterial::pad[0] of the DEVICE copy, bit 0: a DiffuseMaterial that binds no texture — in a scene with textures what the split pass (k_split_plain, render.hip) sends to
    // the lean fused kernel: a diffuse bounce ends the ray differentials (interaction.rs:430-514: only specular bounces carry them on), so nothing a later texture look-up reads
    // depends on which kernel shaded such a vertex
    for (ShmMaterial& m : out.materials) {
        const bool tex_a = m.a.kind == SHM_SPECTRUM_IMAGE_TEXTURE || m.a.kind == SHM_SPECTRUM_TEXTURE_NODE;
        bool any_float_tex = false;
        for (int k = 0; k < 8; ++k) any_float_tex = any_float_tex || m.float_tex[k] != 0u;
        m.pad[0] = (m.kind == SHM_MATERIAL_DIFFUSE && !tex_a && !any_float_tex && m.normal_map == 0u) ? 1u : 0u;
    }
    for (const shm::PrimRec& pr : out.prim_recs) {
        const bool plain = pr.material < out.materials.size() && (out.materials[pr.material].pad[0] & 1u);
        if (plain) out.n_plain_diffuse_prims += 1u;
        // ... and by SURFACE AREA (round 6): where the hits fall. A textured 4.3 M-triangle object in a plain room of 14 triangles is 3 ppm plain by count and half plain by hits
        double area = 0.0;
        if (pr.kind_index & shm::PRIM_SPHERE_BIT) { const double r = d->spheres[pr.kind_index & shm::PRIM_INDEX_MASK].radius; area = 4.0 * 3.14159265358979 * r * r; }
        else if (pr.kind_index & shm::PRIM_PATCH_BIT) area = out.patches[pr.kind_index & shm::PRIM_INDEX_MASK].area;
        else if (!(pr.kind_index & shm::PRIM_INSTANCE_BIT)) {
            const double ax = (double)pr.p1[0] - pr.p0[0], ay = (double)pr.p1[1] - pr.p0[1], az = (double)pr.p1[2] - pr.p0[2];
            const double bx = (double)pr.p2[0] - pr.p0[0], by = (double)pr.p2[1] - pr.p0[1], bz = (double)pr.p2[2] - pr.p0[2];
            const double cx = ay * bz - az * by, cy = az * bx - ax * bz, cz = ax * by - ay * bx;
            area = 0.5 * std::sqrt(cx * cx + cy * cy + cz * cz);
        }
        if (area == area && area < 1e300) { out.area_total += area; if (plain) out.area_plain_diffuse += area; }
    }
    out.has_material_textures = out.has_textures;
    out.has_textures = out.has_material_textures || out.has_image_light;
    return SHM_OK;
}

}  // namespace shm_host
