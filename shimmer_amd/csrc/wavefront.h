// wavefront.h — shared declarations of the gfx950 wavefront path tracer behind include/shimmer_hip.h: path state (SoA in HBM), queue
// bookkeeping, the scene object, and the launchers each kernel translation unit exports. The kernels are split over several .hip files
// (k_trace.hip, k_shade_*.hip, render.hip) so that they compile in parallel; nothing here is exported from the library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "host/flatten.h"
#include "shm/path.h"

using namespace shm;

// thread-local message behind shm_last_error() (defined in render.hip)
__attribute__((visibility("hidden"))) std::string& shm_err();

namespace wf {

#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            shm_err() = std::string(#expr) + ": " + hipGetErrorString(_e);                           \
            (void)hipGetLastError(); /* reported here: must not resurface at the next launch's LAUNCH_TRY */ \
            return SHM_ERR_DEVICE;                                                               \
        }                                                                                        \
    } while (0)

// every kernel launch is followed by LAUNCH_TRY: a failed launch (bad configuration, missing code object) is reported by the call that
// made it, not by the stream synchronisation at the end
#define LAUNCH_TRY(what)                                                                         \
    do {                                                                                         \
        hipError_t _e = hipGetLastError();                                                       \
        if (_e != hipSuccess) {                                                                  \
            shm_err() = std::string("launch of ") + (what) + ": " + hipGetErrorString(_e);           \
            return SHM_ERR_DEVICE;                                                               \
        }                                                                                        \
    } while (0)

static inline bool dbg_on() { static int v = -1; if (v < 0) v = getenv("SHM_DEBUG") ? 1 : 0; return v == 1; }
#define DBG(...) do { if (dbg_on()) { fprintf(stderr, "[shm] " __VA_ARGS__); fprintf(stderr, "\n"); fflush(stderr); } } while (0)

constexpr int WAVE = 64;
constexpr int TRACE_BLOCK = 256;  // 4 waves per workgroup
constexpr int SHADE_BLOCK = 128;

// The DEVICE copy of a BVH node (render.hip lays the tree out by sibling pairs and rewrites `offset` into a link word; the ABI's array and the
// oracle's keep the reference's fields): everything a traversal step needs to go on from a node without reading it again —
//   interior  axis << 29 | index of the first child (the second is + 1; pairs start at even indices, so (index << 5) ^ 32 is the sibling's byte offset)
//   leaf      1 << 31 | min(n_prims, 7) << 27 | offset of its first primitive record (a count of 7 means: read ShmScene::d_big_leaf_n[offset], the primitives left from that slot on);
//             bit 30 is 0 in the array and set by a traversing lane (k_trace5<., GEN>): "the record at this slot is no triangle, its test is pending"
// `n_prims` and `axis` stay where they were (no kernel reads them since the one-node-step kernels were retired: the link word holds everything).
constexpr uint32_t HIT_HAS_SECOND = 0x40000000u;  // the compact hit records' primitive word, split form (scenes with spheres / patches, no instances): this hit has a second record
constexpr uint32_t LINK_LEAF = 0x80000000u, LINK_OTHER = 0x40000000u, LINK_INDEX_MASK = 0x07ffffffu, LINK_COUNT_SHIFT = 27u, LINK_COUNT_MAX = 7u, LINK_AXIS_SHIFT = 29u;

struct DeviceCounters {
    unsigned long long rays_closest, rays_any, nodes_closest, tris_closest, nodes_any, tris_any, paths;
};

// Queue bookkeeping that lives in HBM so that no launch needs a host round trip.
struct QueueState {
    uint32_t n_active[2];   // entries in q_active[0/1]
    uint32_t n_shadow[2];   // entries in q_shadow, double-buffered by bounce parity: K3 of bounce b may still be reading its count
                            // while the counters of bounce b+1 are recycled (K3(b) overlaps K2(b+1) on a second stream)
    uint32_t n_scatter[4];  // staged shading: entries in q_scatter[class], filled by k_vertex, drained by k_scatter<class>
    uint32_t n_lean;        // staged shading with the lean diversion: hits on plain DiffuseMaterial, which k_vertex hands to the fused kernel whole
    uint32_t n_emit;        // the fused kernel's deferred emitter hits (q_emit), worked off by k_emit_jobs after it
    uint32_t n_split;       // scenes with textures: the hits the split pass left to the textured kernels (q_split; the plain-diffuse ones went to q_lean)
};

// The part of a path's state that EVERY vertex reads (and rewrites), as one 64-byte record. As six separate arrays (rounds 1-4) a vertex touched six cache lines for
// these 56 bytes — harmless while every path of a line is alive, but paths die (Russian roulette, escapes) and the survivors stay where they are: at bounce 4 of the
// headline frame one path in ten is left, a 128-byte line serves one of them, and the fused kernel fetched 666 bytes per vertex against 108 at bounce 0
// (rocprofv3 FETCH_SIZE per dispatch). One record: one half line per vertex whatever the survival.
struct alignas(64) PathRec {
    // first 32-byte sector: what k_generate writes (all of it, so that a new path costs one full sector, not two partial ones) and bounce 0 reads
    float4 lambda;     // (the film reads its own copy, PathArrays::lambda)
    uint2 rng;         // PCG32 state (inc is re-derived from pixel + seed)
    uint32_t pixel;    // x | y << 16 (absolute pixel coordinates, < 65536)
    uint32_t flags;    // depth | specular_bounce << 8 | any_non_specular << 9 | ray has auxiliary rays << 10
    // second sector: written by the first vertex (a new path's constants — beta = 1, p_b = eta_scale = 1 — are not stored where the consumer knows them: k_generate<LEAN>)
    float4 beta;
    float2 pb_eta;     // p_b, eta_scale
    uint32_t pad[2];
};
static_assert(sizeof(PathRec) == 64, "PathRec is one half cache line");
// Staged shading's parameter block — what get_bsdf left at a vertex, written by k_vertex and read by the scatter kernel of the vertex's BxDF class (the layered one
// reads it in each of its stages) — as one 128-byte record, ordered so that a class reads whole 32-byte sectors: {bx0, bx2 | fr, bx1} for diffuse / conductor /
// dielectric, + {bx3, bx4} for the coated ones, + {siwo} in scenes with quadrics, patches or instances. Six separate arrays before (PathRec, above, for the why).
struct alignas(128) BxRec {
    float4 bx0;   // r[4]: reflectance (Diffuse, CoatedDiffuse) / conductor eta (Conductor, CoatedConductor)
    float4 bx2;   // eta, alpha_x, alpha_y, kind | max_depth << 8 | n_samples << 20 (as bits)
    float4 fr;    // shading frame x = normalize(dpdus) (y = z cross x is recomputed; z = ns lives in CtxRec::c2)
    float4 bx1;   // k[4]: conductor absorption
    float4 bx3;   // albedo[4]                         (coated materials)
    float4 bx4;   // alpha_x2, alpha_y2, thickness, g  (coated materials)
    float4 siwo;  // intr.wo (scenes with non-triangle shapes or instances only: elsewhere it is -ray.d bit for bit)
    float4 pad;
};
static_assert(sizeof(BxRec) == 128, "BxRec is one cache line");
// A vertex's LightSampleContext (light.rs:1001-1009): the scatter half's geometry and the next vertex's prev_intr_ctx, one record
struct alignas(64) CtxRec { float4 c0, c1, c2, pad; };
// Path state (DESIGN.md §"Data layout in HBM"). All arrays have `capacity` entries.
struct PathArrays {
    ShmRay* ray;            // 32 B: o, d, t_max — input of K2
    ShmHit* hit;            // 32 B: output of K2 — or, with hit16, 16 B per path in the same allocation: {primitive, b0, b1, b2} as one float4
    const float4* hit_prev; // hit16 renders of an all-diffuse triangle scene: the 16-byte records are double-buffered by bounce parity in the two halves of `hit`; this is the previous bounce's
    uint32_t hit16;         // set per render: a triangle scene without textures under the path integrator (every consumer is a TRI_ONLY kernel, none reads a triangle hit's t)
    ShmRay* shadow_ray;     // 32 B: input of K3
    float4* shadow_contrib; // beta * Ld, added to L by K3 when unoccluded
    float4* L;
    PathRec* rec;           // 64 B: beta, lambda, p_b / eta_scale, sampler state, pixel, flags — what every vertex reads of its path, in ONE half line (below)
    float4* lambda;         // the film's copy of the wavelengths (written once by k_generate; the shading kernels read PathRec::lambda)
    uint2* rng0;            // all-diffuse triangle scenes (bounce 0 on known constants): what k_generate<LEAN> leaves INSTEAD of a record — the sampler state ...
    uint32_t* pixel0;       // ... and the pixel; the fused kernel's bounce 0 reads them (and `lambda`) and writes the path's first record whole
    float4* lambda_pdf;
    CtxRec* ctx;            // 64 B: prev_intr_ctx — c0 = pi.low.xyz, pi.high.x; c1 = pi.high.yz, n.xy; c2 = n.z, ns.xyz (one record: three arrays before)
    // scenes with image textures only (null otherwise): the ray's AuxiliaryRays (ray.rs:104-135)
    float4* aux0;           // rx_origin.xyz, rx_direction.x
    float4* aux1;           // rx_direction.yz, ry_origin.xy
    float4* aux2;           // ry_origin.z, ry_direction.xyz
    // staged shading (k_vertex -> k_scatter<class>; null when the scene runs the fused kernel): the BxDF parameter block get_bsdf
    // left at this vertex and the x axis of its shading frame. The rest of the vertex geometry (pi, n, ns) is the CtxRec, which
    // k_vertex overwrites with THIS vertex's LightSampleContext once the previous one has served the emitter MIS weight.
    BxRec* bx;              // 128 B: the parameter block as ONE record per path (above)
    uint32_t has_layered;   // the scene holds coated materials: k_vertex writes the record's third sector (bx3, bx4)
    // scenes with image textures only: what Igehy's specular differentials need beside the auxiliary rays (interaction.rs:430-514)
    const float4* hit2;     // hit16 in a scene with spheres / patches: the second records (t, phi) of the hits that have one (HIT_HAS_SECOND), behind the first ones in the hit allocation
    float4* dd0;            // dpdx.xyz, dpdy.x
    float4* dd1;            // dpdy.yz, dndx.xy
    float4* dd2;            // dndx.z, dndy.xyz
    // scenes whose vertices (or some of them: the lean diversion) run the fused kernel: what emission at the hit needs of the state that kernel overwrites,
    // written for the rare vertex that hit an emitter and read back by k_emit_jobs (k_shade.inl)
    float4* e_ray;          // ray.d.xyz, p_b
    float4* e_beta;
    float4* e_ctx0;         // prev_intr_ctx as ctx0..2 (only written when the MIS weight needs it)
    float4* e_ctx1;
    float4* e_ctx2;
    uint32_t* e_flags;      // the path's flags word at the hit (depth, specular_bounce)
};
enum : int { CLASS_DIFFUSE = 0, CLASS_CONDUCTOR = 1, CLASS_DIELECTRIC = 2, CLASS_LAYERED = 3, N_BXDF_CLASSES = 4 };
__host__ __device__ inline int bxdf_class_of(uint32_t kind) {
    return kind == SHM_MATERIAL_DIFFUSE ? CLASS_DIFFUSE : (kind == SHM_MATERIAL_CONDUCTOR ? CLASS_CONDUCTOR : (kind <= SHM_MATERIAL_THIN_DIELECTRIC ? CLASS_DIELECTRIC : CLASS_LAYERED));
}

__device__ __forceinline__ AuxRays ld_aux(const PathArrays& pa, uint32_t path) {
    float4 a = pa.aux0[path], b = pa.aux1[path], c = pa.aux2[path];
    AuxRays x;
    x.has = true;
    x.rx_o = v3(a.x, a.y, a.z); x.rx_d = v3(a.w, b.x, b.y);
    x.ry_o = v3(b.z, b.w, c.x); x.ry_d = v3(c.y, c.z, c.w);
    return x;
}
__device__ __forceinline__ void st_aux(const PathArrays& pa, uint32_t path, const AuxRays& x) {
    pa.aux0[path] = make_float4(x.rx_o.x, x.rx_o.y, x.rx_o.z, x.rx_d.x);
    pa.aux1[path] = make_float4(x.rx_d.y, x.rx_d.z, x.ry_o.x, x.ry_o.y);
    pa.aux2[path] = make_float4(x.ry_o.z, x.ry_d.x, x.ry_d.y, x.ry_d.z);
}

// ---- small scene tables staged in LDS --------------------------------------------------------------------------------------------------
// The material table, the light table and the spectrum pool are tens of bytes to a few KB, and every vertex reaches them at the END of a chain of dependent
// gathers (hit -> primitive -> material -> spectrum samples; light -> spectrum samples): from L2 each link costs a vector-memory round trip. A workgroup copies them
// into LDS once and the kernel goes on with a SceneView whose pointers name the copies — generic pointers into the LDS aperture: the shared leaf code (shm/*.h,
// compiled for the oracle too) reads them through the same flat loads as before, at LDS latency. Same bytes: results cannot change. `LdsTables` says what fits
// (host side: wf_lds_tables); a scene whose tables exceed the budget runs with the global pointers (bytes = 0).
constexpr uint32_t LDS_TABLE_BUDGET = 12 * 1024, LDS_TABLE_BUDGET_SMALL = 1536;
// the tables a kernel may stage, hottest and smallest first (the host fills a kernel's budget greedily in this order: wf_lds_tables, render.hip)
enum : int { LT_MESH_FLAGS, LT_LIGHTS, LT_LIGHT_PRIMS, LT_MATERIALS, LT_SPECTRUM, LT_RGB2SPEC_SCALE, LT_IMAGE_TEXTURES, LT_IMAGE_LEVELS, LT_FLOAT_TEXTURES, LT_FTEX_RANGES, LT_FTEX_OPS, LT_SPECTRUM_TEXTURES,
              LT_STEX_RANGES, LT_STEX_OPS, LT_EWA_LUT, N_LDS_TABLES };  // (rgb2spec's scale table: 64 floats, walked by a binary search of six DEPENDENT loads per lookup)
struct LdsTables {
    uint32_t bytes[N_LDS_TABLES];  // each a multiple of 16 (dev_upload allocates whole 16-byte groups); 0: that table stays in global memory
};
__device__ __forceinline__ SceneView stage_scene_tables(const SceneView& sv, const LdsTables& t, uint4* lds) {
    SceneView out = sv;
    uint32_t total = 0;
#pragma unroll
    for (int k = 0; k < N_LDS_TABLES; ++k) total += t.bytes[k];
    if (total == 0u) return out;
    uint32_t at = 0;  // in uint4s
#define SHM_STAGE_TABLE(K, FIELD, TYPE)                                                                \
    if (t.bytes[K]) {                                                                                  \
        const uint32_t n = t.bytes[K] / 16u;                                                           \
        const uint4* g = reinterpret_cast<const uint4*>(sv.FIELD);                                     \
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) lds[at + i] = g[i];                     \
        out.FIELD = reinterpret_cast<const TYPE*>(lds + at);                                           \
        at += n;                                                                                       \
    }
    SHM_STAGE_TABLE(LT_MESH_FLAGS, mesh_flags, uint32_t)
    SHM_STAGE_TABLE(LT_LIGHTS, lights, ShmLight)
    SHM_STAGE_TABLE(LT_LIGHT_PRIMS, light_prim_recs, PrimRec)
    SHM_STAGE_TABLE(LT_MATERIALS, materials, ShmMaterial)
    SHM_STAGE_TABLE(LT_SPECTRUM, spectrum_data, Float)
    SHM_STAGE_TABLE(LT_RGB2SPEC_SCALE, rgb2spec_scale, Float)
    SHM_STAGE_TABLE(LT_IMAGE_TEXTURES, image_textures, ShmImageTexture)
    SHM_STAGE_TABLE(LT_IMAGE_LEVELS, image_levels, ShmImageLevel)
    SHM_STAGE_TABLE(LT_FLOAT_TEXTURES, float_textures, ShmFloatTexture)
    SHM_STAGE_TABLE(LT_FTEX_RANGES, ftex_ranges, FloatTexRange)
    SHM_STAGE_TABLE(LT_FTEX_OPS, ftex_ops, FloatTexOp)
    SHM_STAGE_TABLE(LT_SPECTRUM_TEXTURES, spectrum_textures, ShmSpectrumTexture)
    SHM_STAGE_TABLE(LT_STEX_RANGES, stex_ranges, FloatTexRange)
    SHM_STAGE_TABLE(LT_STEX_OPS, stex_ops, FloatTexOp)
    SHM_STAGE_TABLE(LT_EWA_LUT, ewa_lut, Float)
#undef SHM_STAGE_TABLE
    __syncthreads();
    return out;
}

// ... and for the kernels that evaluate textures: the staged view itself copied to LDS once per workgroup, for the evaluators that are real calls (shm/texture.h)
constexpr uint32_t SCENE_VIEW_UINT4S = (sizeof(SceneView) + 15) / 16;
__device__ __forceinline__ void attach_call_copy(SceneView& sv, uint4* slot) {
    SceneView* call_copy = reinterpret_cast<SceneView*>(slot);
    sv.call_copy = call_copy;
    if (threadIdx.x == 0) *call_copy = sv;
    __syncthreads();
}
// (HAS_TEX = false: the plain staging; `slot` is a one-element dummy then)
template <bool HAS_TEX>
__device__ __forceinline__ SceneView stage_scene_tables_tex(const SceneView& sv, const LdsTables& t, uint4* lds, uint4* slot) {
    SceneView out = stage_scene_tables(sv, t, lds);
    if (HAS_TEX) attach_call_copy(out, slot);
    return out;
}

// the primitive of a compact hit record's first word (a miss stays negative)
__device__ __forceinline__ int hit_prim_of(int w) { return w < 0 ? w : (int)((uint32_t)w & ~HIT_HAS_SECOND); }
// a path's hit record as the TRI_ONLY shading kernels read it: the 32-byte ShmHit, or the compact form the render's own traversal launches write (PathArrays::hit16)
__device__ __forceinline__ Hit load_hit_tri(const PathArrays& pa, uint32_t path) {
    Hit hit;
    if (pa.hit16) {
        const float4 h = reinterpret_cast<const float4*>(pa.hit)[path];
        const int w = __float_as_int(h.x);
        hit.prim = hit_prim_of(w); hit.t = 0.0f; hit.b0 = h.y; hit.b1 = h.z; hit.b2 = h.w; hit.phi = 0.0f; hit.inst = -1;
        if (w >= 0 && ((uint32_t)w & HIT_HAS_SECOND)) {  // (the split form: a sphere / patch hit's t and phi — p_obj / (u, v) ride in b0..b2)
            const float4 h2 = pa.hit2[path];
            hit.t = h2.x; hit.phi = h2.y;
        }
    } else {
        const float4* hp = reinterpret_cast<const float4*>(pa.hit + path);
        const float4 h0 = hp[0], h1 = hp[1];
        hit.prim = __float_as_int(h0.x); hit.t = h0.y; hit.b0 = h0.z; hit.b1 = h0.w; hit.b2 = h1.x; hit.phi = h1.y; hit.inst = __float_as_int(h1.z) - 1;
    }
    return hit;
}

__device__ __forceinline__ uint32_t wave_lane() { return __lane_id(); }

// Wave-aggregated append: one atomic per wave, lanes get consecutive slots.
__device__ __forceinline__ uint32_t queue_push_slot(uint32_t* counter, bool pred) {
    unsigned long long mask = __ballot(pred);
    if (mask == 0ull) return 0u;
    uint32_t lane = wave_lane();
    int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    return base + rank;
}
__device__ __forceinline__ Spec ld_spec(const float4& f) { Spec s; s.v[0] = f.x; s.v[1] = f.y; s.v[2] = f.z; s.v[3] = f.w; return s; }
__device__ __forceinline__ float4 st_spec(const Spec& s) { return make_float4(s.v[0], s.v[1], s.v[2], s.v[3]); }
constexpr int SHADE2_BLOCK = 256;
constexpr int SHADE_CHUNK = 2048;  // queue entries per workgroup chunk: ONE global atomic per queue per chunk (a single
                                   // counter word saturates near 88 atomics/us: MI355X_MICROARCH.md "dequeue")
// Two waves per SIMD for every instantiation: the lean one needs 236 VGPRs anyway; the ones with the quadric / patch or the
// LayeredBxDF code want 264 / 460 and are better off spilling a little than running one wave per SIMD (measured: patch scene
// shade 16.6 -> 10.9 ms, coated S3 187 -> 177 ms; 3 or 4 waves lose to spills).
#ifndef K_SHADE_WAVES
#define K_SHADE_WAVES 2
#endif
#define K_SHADE_ATTR __attribute__((amdgpu_waves_per_eu(K_SHADE_WAVES, K_SHADE_WAVES)))
}  // namespace wf
using namespace wf;

// ---------------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------------
struct ShmScene {
    int device = 0;
    hipStream_t stream = nullptr;
    shm_host::FlatScene flat;
    SceneView dsv;                 // device pointers
    std::vector<void*> allocs;
    std::vector<void*> ws_allocs;  // path workspace (regrown on demand)
    ShmFilmPixel* d_film = nullptr;
    size_t n_film_pixels = 0;
    // path workspace
    uint32_t capacity = 0;
    PathArrays pa;
    uint32_t* d_q_active[2] = {nullptr, nullptr};
    uint32_t* d_q_shadow = nullptr;
    uint32_t* d_q_scatter[4] = {nullptr, nullptr, nullptr, nullptr};  // staged shading: one queue per BxDF class present in the scene
    uint32_t* d_q_split = nullptr; // scenes with textures and plain diffuse materials: what the split pass leaves to the textured kernels
    int split_pass = 0;            // ... that pass is on (SHM_SPLIT_PASS; a quarter of the primitives or more are plain diffuse)
    uint32_t* d_q_lean = nullptr;  // the lean diversion's queue (triangle-only scenes without textures that hold plain diffuse materials beside others)
    bool lean_divert = false;      // (round 4 A/B: DESIGN.md section 6)
    bool staged = false;           // the scene class runs k_vertex -> k_scatter<class> (everything but all-diffuse triangle scenes without textures)
    bool ws_staged = false;        // the workspace holds the staging arrays
    QueueState* d_qs = nullptr;
    DeviceCounters* d_counters = nullptr;
    uint32_t* d_pixels = nullptr;
    size_t pixels_capacity = 0;
    ShmTile* d_tiles = nullptr;
    uint32_t* d_tile_offset = nullptr;
    size_t tiles_capacity = 0;
    std::vector<uint64_t> tile_bitmap;  // host scratch of the disjointness check in shm_render_wave
    int n_cu = 256;
    // tuned traversal (k_trace5; the field names date from the retired one-node-step kernels)
    int trace3_blocks[2] = {0, 0};   // persistent grid of the closest-hit [0] / any-hit [1] entry point this scene uses (k_trace.hip: K5Shape)
    int spill3_levels[2] = {1, 1};   // stack levels beyond the entry point's LDS levels (HBM spill)
    int leaf_min = 16;             // closest-hit: lanes with a pending leaf before the triangle phase runs (SHM_LEAF_MIN)
    int leaf_min_any = 8;          // any-hit (SHM_LEAF_MIN_ANY)
    uint32_t* d_spill3 = nullptr;
    LdsTables lds_tables = {};        // the small tables the shading kernels stage in LDS within the full budget (render.hip: wf_lds_tables; SHM_LDS_TABLES=0: nothing)
    LdsTables lds_tables_small = {};  // ... within the 1.5 KB the material-sorted triangle vertex kernel has to spare
    uint32_t* d_q_emit = nullptr;      // paths of the current fused-kernel launch that hit an emitter (k_emit_jobs)
    uint32_t* d_big_leaf_n = nullptr;  // n_prims by first primitive slot, only in scenes with a leaf of >= 7 primitives (LINK_COUNT_MAX: the link word holds smaller counts)
    float4* d_rw = nullptr;          // RandomWalk: (le, f cos) per depth per path, 2 * (max_depth + 1) * capacity float4
    size_t rw_floats4 = 0;
    uint32_t* d_spill3_any = nullptr;  // the any-hit kernel may run concurrently with the closest-hit one (second stream)
    float4* d_gen_save[2] = {nullptr, nullptr};  // k_trace5<., GEN> (scenes with spheres / patches / instances): per resident lane two 48-byte areas for the ray state (closest, any)
    // scenes without a coated material: ONE fused all-materials launch per bounce from this bounce on (SHM_TAIL_FUSED_BOUNCE, negative = never: the staged kernels before
    // it — the test instrument that holds the two pipelines against each other), its chunks sorted by material
    int tail_fused_bounce = 0;
    bool gen_heavy = false;        // a quarter or more of the primitive records are spheres / bilinear patches (a quad PLY file: every face a patch): the traversal kernels'
                                   // five-wave instantiations (k_trace.hip, K5_GEN_HEAVY_WAVES; SHM_GEN_HEAVY=0 / 1 overrides: the suite runs both)
    int other_min = 16, other_min_any = 16;     // ... and the parked non-triangle tests a wave collects before it runs them (SHM_OTHER_MIN, SHM_OTHER_MIN_ANY)
    hipStream_t stream2 = nullptr;
    hipStream_t stream_cls[4] = {nullptr, nullptr, nullptr, nullptr};  // staged shading: the scatter kernels of the 2nd .. 4th BxDF class of a bounce run beside the first one's
    uint64_t overlap_paths = 96ull << 20;  // batches below this many paths run K3(b) beside K2(b+1) (SHM_OVERLAP_PATHS; 0 = never)
    // idle lanes before a traversal wave refills: the kernels set a ray up with the root test and six IEEE divisions (200 VALU instructions), so that fewer, fuller refills
    // win although a quarter of the lanes idle (r04 sweep on the headline frame, closest / any ms: 24: 97.4 / 62.3, 40: 96.5 / 60.9, 48: 102.1 / 63.5, 56: 119.1 / 78.1)
    int refill_min_any = 40;       // the any-hit kernel's threshold (SHM_REFILL_MIN_ANY)
    int refill_min = 40;           // (SHM_REFILL_MIN)
    int trace_rays_per_lane = 4;   // a traversal launch uses as much of its persistent grid as gives each resident lane about this many rays (SHM_TRACE_RAYS_PER_LANE; 0 = always
                                   // the whole grid). profiles/r03_trace_rays_per_lane_sweep.txt: C2 16.8 / 16.1 / 15.8 / 15.8 / 16.5 ms at 0 / 4 / 8 / 16 / 32, C4's K2 276.7 / 276.8 / 282 / 307 / 362
    uint32_t pix_group = 1024;      // path-slot order [tile][sample][pixel in tile] (>= n_pix would be sample-major; swept in round 1: profiles/r01_*)
    int queue_parts = 8;           // the traversal queue's partitions, one per XCD with stealing
    uint32_t* d_heads3 = nullptr;  // [2 (closest, any)][8 partitions][32 dwords: one 128-B line per head word]
    std::vector<hipEvent_t> events;
    struct DistState* dist = nullptr;  // multi-GPU state (dist.hip): communicator, this rank's tile shard, the gather plan
};
// dist.hip: releases s->dist (RCCL communicator included); called by shm_scene_destroy
__attribute__((visibility("hidden"))) void wf_dist_release(ShmScene* s);

struct EventPool {
    ShmScene* s;
    size_t used = 0;
    bool failed = false;  // hipEventCreate failed: checked once per render (events are only used for timing / stream ordering)
    hipEvent_t get() {
        if (used == s->events.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) { failed = true; return nullptr; }
            s->events.push_back(e);
        }
        return s->events[used++];
    }
};

// ---- launchers exported by the kernel translation units (hidden visibility: library-internal) ----
#define WF_INTERNAL __attribute__((visibility("hidden")))
// k_trace.hip: BvhAggregate::intersect (any = false) / intersect_predicate (any = true) over a queue of path slots
WF_INTERNAL LdsTables wf_lds_tables(const ShmScene* s, uint32_t budget);  // render.hip: which small tables fit `budget` bytes of LDS
WF_INTERNAL void wf_trace_census();  // k_trace.hip: prints the per-phase lane census of a -DK5_CENSUS development build (a no-op otherwise)
WF_INTERNAL void wf_layered_census();  // k_scatter_layered_staged_tri.hip: the same for a -DLJ_CENSUS build of the staged LayeredBxDF kernel
WF_INTERNAL int wf_trace_prepare(ShmScene* s);  // grid sizes + stack spill buffers of the two traversal kernels (at scene creation)
WF_INTERNAL int wf_launch_trace(ShmScene* s, bool any, hipStream_t stream, const uint32_t* queue, const uint32_t* n_ptr, uint32_t n_direct,
                                const ShmRay* rays, ShmHit* hits, uint8_t* occluded, float4* L, const float4* contrib, int hit16 = 0);
// shading of one path vertex of PathIntegrator::li for every entry of q_active[cur]
struct ShadeArgs {
    hipStream_t stream;
    int cur;
    ShmRenderParams params;
    int shadow_parity;
    int blocks;
    int first_bounce = 0;  // 1: bounce 0 of a render whose k_generate left the constants out (beta = 1, p_b = eta_scale = 1, flags = 0, the identity queue): the fused kernel knows them
    int hit_kept = 0;      // 1: the hit records are double-buffered by bounce parity (PathArrays::hit_prev is the previous bounce's): the fused kernel leaves nothing for the next vertex
    const uint32_t* q_in = nullptr;  // k_vertex: the queue to work through instead of q_active[cur], and its count (the split pass's q_split)
    const uint32_t* n_in = nullptr;
};
WF_INTERNAL int wf_launch_shade_lean(ShmScene* s, const ShadeArgs& a);  // the fused kernel: all-diffuse triangle scenes without textures
WF_INTERNAL int wf_launch_shade_lean_diverted(ShmScene* s, const ShadeArgs& a);
WF_INTERNAL int wf_launch_shade_lean_gen(ShmScene* s, const ShadeArgs& a);           // the same kernel with the quadric / patch / instance code: scenes that hold such shapes (k_shade_lean_gen.hip)
WF_INTERNAL int wf_launch_shade_lean_gen_diverted(ShmScene* s, const ShadeArgs& a);
WF_INTERNAL int wf_launch_shade_lean_env(ShmScene* s, const ShadeArgs& a);      // the lean fused kernel with an ImageInfinitelight compiled in (k_shade_lean_env.hip)
WF_INTERNAL int wf_launch_shade_lean_gen_env(ShmScene* s, const ShadeArgs& a);  // ... for general geometry (k_shade_lean_gen_env.hip)
// the staged kernels of a scene whose only image is an ImageInfinitelight (K_ENV_LIGHT units: k_vertex_env.hip, k_scatter_*_env.hip, k_scatter_layered*_env.hip)
WF_INTERNAL int wf_launch_vertex_tri_env(ShmScene* s, const ShadeArgs& a);
WF_INTERNAL int wf_launch_vertex_gen_env(ShmScene* s, const ShadeArgs& a);
WF_INTERNAL int wf_launch_scatter_diffuse_env(ShmScene* s, const ShadeArgs& a, bool tri_only);
WF_INTERNAL int wf_launch_scatter_conductor_env(ShmScene* s, const ShadeArgs& a, bool tri_only);
WF_INTERNAL int wf_launch_scatter_dielectric_env(ShmScene* s, const ShadeArgs& a, bool tri_only);
WF_INTERNAL int wf_launch_scatter_layered_staged_tri_env(ShmScene* s, const ShadeArgs& a);
WF_INTERNAL int wf_launch_scatter_layered_staged_gen_env(ShmScene* s, const ShadeArgs& a);
WF_INTERNAL int wf_launch_shade_lean_env_diverted(ShmScene* s, const ShadeArgs& a);
WF_INTERNAL int wf_launch_shade_lean_gen_env_diverted(ShmScene* s, const ShadeArgs& a);
WF_INTERNAL int wf_launch_shade_tail_sorted_env(ShmScene* s, const ShadeArgs& a);  // the sorted fused all-materials kernel with an ImageInfinitelight compiled in (k_shade_tail_sorted_env.hip)
WF_INTERNAL int wf_launch_shade_fused_gen_env(ShmScene* s, const ShadeArgs& a);    // ... for general geometry (k_shade_fused_gen_env.hip)
WF_INTERNAL int wf_launch_shade_fused_gen(ShmScene* s, const ShadeArgs& a);    // ... for scenes with spheres / patches / instances (k_shade_fused_gen.hip)
WF_INTERNAL int wf_launch_shade_fused_gen_tex(ShmScene* s, const ShadeArgs& a);  // ... and those with textures (k_shade_fused_gen_tex.hip)
WF_INTERNAL int wf_launch_shade_fused_tex(ShmScene* s, const ShadeArgs& a);    // ... and for triangle scenes with textures, no coated material (k_shade_fused_tex.hip)
WF_INTERNAL int wf_launch_shade_tail_sorted(ShmScene* s, const ShadeArgs& a);  // ... with material-sorted chunks (k_shade_tail_sorted.hip)
// (the unsorted tail kernel of rounds 3-4, k_shade_tail.hip, left the library in round 6: k_shade_tail_sorted.hip below takes its scenes)
// staged shading (k_vertex_*.hip, k_scatter_*.hip): the hit half of a vertex (interaction, emission + MIS, get_bsdf with its texture
// evaluation -> BxDF parameter block, pushed to the queue of its BxDF class), then per class the scattering half (NEE, sample_f, RR)
WF_INTERNAL int wf_launch_vertex_tri(ShmScene* s, const ShadeArgs& a);
WF_INTERNAL int wf_launch_vertex_gen(ShmScene* s, const ShadeArgs& a);
WF_INTERNAL int wf_launch_vertex_tex(ShmScene* s, const ShadeArgs& a);
WF_INTERNAL int wf_launch_scatter_diffuse(ShmScene* s, const ShadeArgs& a, bool tri_only, bool has_tex);
WF_INTERNAL int wf_launch_scatter_conductor(ShmScene* s, const ShadeArgs& a, bool tri_only, bool has_tex);
WF_INTERNAL int wf_launch_scatter_dielectric(ShmScene* s, const ShadeArgs& a, bool tri_only, bool has_tex);
WF_INTERNAL int wf_launch_scatter_layered_tri(ShmScene* s, const ShadeArgs& a);
WF_INTERNAL int wf_launch_scatter_layered_gen(ShmScene* s, const ShadeArgs& a);
WF_INTERNAL int wf_launch_scatter_layered_tex(ShmScene* s, const ShadeArgs& a);
// the same class as dense per-wave stages (k_scatter_layered.inl): every render but options.force_diffuse; SHM_LAYERED_STAGED=0 for A/B
WF_INTERNAL int wf_launch_scatter_layered_staged_tri(ShmScene* s, const ShadeArgs& a);
WF_INTERNAL int wf_launch_scatter_layered_staged_gen(ShmScene* s, const ShadeArgs& a);
WF_INTERNAL int wf_launch_scatter_layered_staged_tex(ShmScene* s, const ShadeArgs& a);
WF_INTERNAL int wf_launch_shade_simple(ShmScene* s, const ShadeArgs& a);
WF_INTERNAL int wf_launch_shade_randomwalk(ShmScene* s, const ShadeArgs& a, uint32_t cap_eff);
WF_INTERNAL int wf_launch_fold_randomwalk(ShmScene* s, hipStream_t stream, uint32_t cap_eff, uint32_t total);
