// shimmer_hip.hip — the MI355X (gfx950, wave64) wavefront path tracer behind include/shimmer_hip.h.
//
// Replaces the tile-parallel loop of the reference (paths relative to /root/reference/src):
//   integrator.rs:226-322  ImageTileIntegrator::render      -> shm_render / shm_render_device / shm_render_wave (host loop below)
//   integrator.rs:326-396  evaluate_pixel_sample            -> K1 k_generate
//   aggregate.rs:71-139    BvhAggregate::intersect          -> K2 k_trace3<false, TRI_ONLY> (persistent waves, LDS stack)
//   aggregate.rs:141-203   BvhAggregate::intersect_predicate-> K3 k_trace3<true, TRI_ONLY>
//   integrator.rs:772-892  PathIntegrator::li loop body     -> K4+K5 k_shade<HAS_LAYERED, TRI_ONLY> (one path vertex per launch)
//   integrator.rs:897-963  PathIntegrator::sample_ld        -> inside k_shade (shadow ray deferred to K3)
//   film.rs:548-574        RgbFilm::add_sample              -> K6 k_film (per-pixel ordered f64 sums)
// Leaf arithmetic is the single-source header library csrc/shm/*.h (compiled with -ffp-contract=off).
//
// Execution model: one (pixel, sample) per lane; paths live in SoA arrays in HBM; each bounce is
// trace_closest -> shade -> trace_any over index queues compacted with wave-aggregated atomics; queue
// sizes stay on the device (persistent / grid-stride kernels read them), so a whole render (all fused spp-waves)
// is enqueued on one HIP stream without host round trips. There is no CPU fallback anywhere in this file.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "host/flatten.h"
#include "host/integrator.hpp"
#include "shm/path.h"

using namespace shm;

namespace {

thread_local std::string g_err;

#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            g_err = std::string(#expr) + ": " + hipGetErrorString(_e);                           \
            return SHM_ERR_DEVICE;                                                               \
        }                                                                                        \
    } while (0)

// every kernel launch is followed by LAUNCH_TRY: a failed launch (bad configuration, missing code object) is reported by the call that
// made it, not by the stream synchronisation at the end
#define LAUNCH_TRY(what)                                                                         \
    do {                                                                                         \
        hipError_t _e = hipGetLastError();                                                       \
        if (_e != hipSuccess) {                                                                  \
            g_err = std::string("launch of ") + (what) + ": " + hipGetErrorString(_e);           \
            return SHM_ERR_DEVICE;                                                               \
        }                                                                                        \
    } while (0)

static bool dbg_on() { static int v = -1; if (v < 0) v = getenv("SHM_DEBUG") ? 1 : 0; return v == 1; }
#define DBG(...) do { if (dbg_on()) { fprintf(stderr, "[shm] " __VA_ARGS__); fprintf(stderr, "\n"); fflush(stderr); } } while (0)

constexpr int WAVE = 64;
constexpr int TRACE_BLOCK = 256;  // 4 waves per workgroup
constexpr int SHADE_BLOCK = 128;

struct DeviceCounters {
    unsigned long long rays_closest, rays_any, nodes_closest, tris_closest, nodes_any, tris_any, paths;
};

// Queue bookkeeping that lives in HBM so that no launch needs a host round trip.
struct QueueState {
    uint32_t n_active[2];   // entries in q_active[0/1]
    uint32_t n_shadow[2];   // entries in q_shadow, double-buffered by bounce parity: K3 of bounce b may still be reading its count
                            // while the counters of bounce b+1 are recycled (K3(b) overlaps K2(b+1) on a second stream)
    uint32_t pad[4];
};

// Path state, structure of arrays (DESIGN.md §"Data layout in HBM"). All arrays have `capacity` entries.
struct PathArrays {
    ShmRay* ray;            // 32 B: o, d, t_max — input of K2
    ShmHit* hit;            // 32 B: output of K2
    ShmRay* shadow_ray;     // 32 B: input of K3
    float4* shadow_contrib; // beta * Ld, added to L by K3 when unoccluded
    float4* L;
    float4* beta;
    float4* lambda;
    float4* lambda_pdf;
    float4* ctx0;           // prev_intr_ctx: pi.low.xyz, pi.high.x
    float4* ctx1;           //                pi.high.yz, n.xy
    float4* ctx2;           //                n.z, ns.xyz
    float2* pb_eta;         // p_b, eta_scale
    uint2* rng;             // PCG32 state (inc is re-derived from pixel+seed)
    uint32_t* pixel;        // x | y << 16 (absolute pixel coordinates, < 65536)
    uint32_t* flags;        // depth | specular_bounce << 8 | any_non_specular << 9 | ray has auxiliary rays << 10
    // scenes with image textures only (null otherwise): the ray's AuxiliaryRays (ray.rs:104-135)
    float4* aux0;           // rx_origin.xyz, rx_direction.x
    float4* aux1;           // rx_direction.yz, ry_origin.xy
    float4* aux2;           // ry_origin.z, ry_direction.xyz
};

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

// ---------------------------------------------------------------------------------------------
// K0: expand the tile list into a pixel list (reference loop order inside a tile: x outer, y inner;
// integrator.rs:257-258).  One thread per tile; tiny.
// ---------------------------------------------------------------------------------------------
__global__ void k_expand_tiles(const ShmTile* tiles, const uint32_t* tile_offset, uint32_t n_tiles, uint32_t* pixels) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    ShmTile tl = tiles[t];
    uint32_t k = tile_offset[t];
    for (int x = tl.x0; x < tl.x1; ++x)
        for (int y = tl.y0; y < tl.y1; ++y) pixels[k++] = (uint32_t)x | ((uint32_t)y << 16);
}

// ---------------------------------------------------------------------------------------------
// K1: camera rays for one batch. Path slots are ordered [pixel group][sample][pixel in group] with groups of `pix_group`
// consecutive pixels of the tile-ordered pixel list (pix_group >= n_pix is the plain sample-major order
// slot = s_local * n_pix + p_local; pix_group = 64 keeps all samples of one 8x8 tile adjacent in the queues, so that a
// wave's private queue range, and an XCD's queue partition, is a compact image region).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t slot_of(uint32_t p_local, uint32_t s_local, uint32_t n_pix, uint32_t n_samples, uint32_t pix_group) {
    uint32_t g = p_local / pix_group;
    uint32_t g0 = g * pix_group;
    uint32_t pg = min(pix_group, n_pix - g0);
    return g0 * n_samples + s_local * pg + (p_local - g0);
}

__global__ void __launch_bounds__(SHADE_BLOCK) k_generate(SceneView sv, PathArrays pa, const uint32_t* pixels, uint32_t n_pix,
                                                        int sample_begin, int n_samples, ShmRenderParams params,
                                                        uint32_t* q_active, QueueState* qs, uint32_t pix_group) {
    uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t total = n_pix * (uint32_t)n_samples;
    if (slot >= total) return;
    uint32_t g = slot / (pix_group * (uint32_t)n_samples);
    uint32_t g0 = g * pix_group;
    uint32_t pg = min(pix_group, n_pix - g0);
    uint32_t rem = slot - g0 * (uint32_t)n_samples;
    uint32_t s_local = rem / pg;
    uint32_t p_local = g0 + (rem - s_local * pg);
    uint32_t pix = pixels[p_local];
    int px = (int)(pix & 0xffffu), py = (int)(pix >> 16);
    Rng rng = sampler_start_pixel_sample(px, py, sample_begin + (int)s_local, params.seed);
    Wavelengths lambda;
    Float weight;
    const bool has_tex = pa.aux0 != nullptr;
    AuxRays aux = aux_none();
    Ray r = generate_camera_ray(sv, px, py, rng, params.disable_wavelength_jitter != 0, params.disable_pixel_jitter != 0,
                                lambda, weight, has_tex ? &aux : nullptr, params.samples_per_pixel);
    if (has_tex) st_aux(pa, slot, aux);
    ShmRay ray;
    ray.o[0] = r.o.x; ray.o[1] = r.o.y; ray.o[2] = r.o.z;
    ray.d[0] = r.d.x; ray.d[1] = r.d.y; ray.d[2] = r.d.z;
    ray.t_max = infinity();
    ray.pad = 0.0f;
    pa.ray[slot] = ray;
    pa.L[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    pa.beta[slot] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    pa.lambda[slot] = make_float4(lambda.lambda[0], lambda.lambda[1], lambda.lambda[2], lambda.lambda[3]);
    pa.lambda_pdf[slot] = make_float4(lambda.pdf[0], lambda.pdf[1], lambda.pdf[2], lambda.pdf[3]);
    // (ctx0..2, the previous vertex's LightSampleContext, are first read at depth >= 1, after k_shade has written them)
    pa.pb_eta[slot] = make_float2(1.0f, 1.0f);
    pa.rng[slot] = make_uint2((uint32_t)rng.state, (uint32_t)(rng.state >> 32));
    pa.pixel[slot] = pix;
    pa.flags[slot] = has_tex ? (1u << 10) : 0u;  // camera rays always carry auxiliary rays (camera.rs:1070-1078)
    q_active[slot] = slot;  // first bounce: identity queue
    if (slot == 0) {
        qs->n_active[0] = total;
        qs->n_active[1] = 0;
        qs->n_shadow[0] = 0;
        qs->n_shadow[1] = 0;
    }
}

// ---------------------------------------------------------------------------------------------
// K2 / K3: BVH traversal. The reference's traversal (aggregate.rs:71-203: test the current node, push the far child
// untested, enter the near child; pop on a miss or after a leaf) executed as UNIFORM steps, because the profile of the
// first, reference-shaped kernel (one ray per lane from root to done; profiles/r01_v1) showed ~10 of 64 lanes active per
// VALU instruction: issue-bound by divergence, not by HBM (FETCH_SIZE is 3-6x below the algorithmic bytes).  (A variant
// that fetched and tested both children at the parent was measured and dropped: same results, more loads, no gain.)
//   ANY       intersect_predicate (early out, no hit record) vs intersect (closest hit)
//   TRI_ONLY  scenes made of triangles only; otherwise the leaf phase also carries Sphere::intersect (sphere.rs:95-196)
//  * every loop iteration is one identical step for every lane that has a node to test: [pop if requested] -> fetch the
//    32-B record -> slab test -> push far / enter near, or mark the leaf pending, or request a pop.  No nested loops;
//  * leaf (triangle) tests are POSTPONED: a lane that reached a leaf waits until at least `leaf_min` lanes of its wave
//    have a pending leaf (or no lane can take a node step), then the watertight test runs for all of them at once;
//  * finished lanes are refilled from the queue (one wave-aggregated atomic) once `refill_min` lanes are idle;
//  * stack levels [0, LDS_N) live in LDS as [level][lane]; any deeper level goes to a per-lane HBM region laid out the
//    same way.
// Node and primitive visit counts equal the reference's in both modes (it is the same algorithm, node for node).
// ---------------------------------------------------------------------------------------------
#ifndef K3_CHUNK_MAX
#define K3_CHUNK_MAX 1024
#endif
constexpr int K3_LDS_N = 26;   // stack levels [0, K3_LDS_N) -> LDS (6.5 KiB per wave), deeper levels -> HBM spill
enum : uint32_t { ST_IDLE = 0, ST_NODE = 1, ST_LEAF = 2, ST_DONE = 3 };

template <bool ANY, bool TRI_ONLY>
__global__ void __launch_bounds__(TRACE_BLOCK) k_trace3(SceneView sv, const uint32_t* __restrict__ queue, const uint32_t* __restrict__ n_ptr,
                                                       uint32_t n_direct, uint32_t* head, const ShmRay* __restrict__ rays,
                                                       ShmHit* __restrict__ hits, uint8_t* __restrict__ occluded_out,
                                                       float4* __restrict__ L, const float4* __restrict__ contrib,
                                                       DeviceCounters* counters, uint32_t* __restrict__ spill, int spill_levels,
                                                       int refill_min, int leaf_min, int queue_parts) {
    __shared__ uint32_t lds_stack[(TRACE_BLOCK / WAVE) * K3_LDS_N * WAVE];
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint32_t wave_in_block = threadIdx.x / WAVE;
    uint32_t* const st_lds = lds_stack + wave_in_block * K3_LDS_N * WAVE + lane;
    uint32_t* const st_spill = spill + ((size_t)blockIdx.x * (TRACE_BLOCK / WAVE) + wave_in_block) * (size_t)spill_levels * WAVE + lane;
    const uint32_t n = n_ptr ? *n_ptr : n_direct;
    const char* __restrict__ node_base = reinterpret_cast<const char*>(sv.nodes);
    const char* __restrict__ prim_base = reinterpret_cast<const char*>(sv.prim_recs);
    uint32_t c_nodes = 0, c_prims = 0, c_rays = 0;

    uint32_t state = ST_IDLE;
    bool exhausted = false;  // wave-uniform
    uint32_t w_next = 0, w_end = 0;  // wave-uniform private range of the queue
    // chunk size: large enough that the single head word sees few atomics (it saturates near 88 dequeues/us,
    // MI355X_MICROARCH.md "dequeue"), small enough that the last chunks balance across the resident waves
    const uint32_t n_waves = gridDim.x * (TRACE_BLOCK / WAVE);
    uint32_t chunk = n / (n_waves * 8u);
    chunk = chunk < 64u ? 64u : (chunk > (uint32_t)K3_CHUNK_MAX ? (uint32_t)K3_CHUNK_MAX : chunk);
    chunk = (chunk + 63u) & ~63u;
    bool want_pop = false;
    uint32_t path = 0;
    V3 ro = v3s(0.0f), inv_dir = v3s(0.0f);
    V3 rd_full = v3s(0.0f);  // the direction itself is only kept for non-triangle shapes (TRI_ONLY = false)
    bool negx = false, negy = false, negz = false;
    RayShear rs;
    rs.kx = 0; rs.ky = 1; rs.kz = 2; rs.d = v3s(0.0f); rs.sx = rs.sy = rs.sz = 0.0f;
    Float t_max = 0.0f;
    int32_t hit_prim = -1;
    Float hit_t = 0.0f, hit_b0 = 0.0f, hit_b1 = 0.0f, hit_b2 = 0.0f, hit_phi = 0.0f;
    // TransformedPrimitive (TRI_ONLY = false only): the leaf slot of the instance being traversed (-1: the top-level tree), the
    // instance the current closest hit was found through, t_max as it was outside, and whether this visit found a hit
    int32_t inst_slot = -1, hit_inst = -1;
    Float t_outer = 0.0f;
    bool inst_hit = false;
    constexpr uint32_t INST_SENTINEL = 0xffffffffu;  // stack entry that marks the way back out of an instance
    int sp = 0;
    uint32_t cur = 0;
    uint32_t leaf_off = 0, leaf_n = 0;

    // Stack storage by level: [0, K3_LDS_N) in LDS, anything deeper in the per-lane HBM spill. (The first version kept
    // the LDS window at levels 6..31 and spilled the bottom levels: those are written at the start of every ray and again
    // whenever the traversal comes back near the root, and as HBM stores they made WRITE_SIZE 7x the algorithmic hit writes
    // — profiles/r01_v4. Holding them in registers through select chains was measured too: slower than LDS.)
    // Queue partitions: the queue is cut into `queue_parts` contiguous ranges, each with its own head word (own 128-B
    // line). A wave starts in the partition of the XCD it runs on — queue order is image order (pix_group), so one XCD's L2
    // serves one image region's part of the BVH, and 8 head words see 1/8 of the atomics each (one word saturates near
    // 88 dequeues/us: MI355X_MICROARCH.md "dequeue") — and moves on to the next partition when its own has run dry.
    const uint32_t n_parts = (uint32_t)queue_parts;
    const uint32_t part_size = ((n + n_parts * 64u - 1u) / (n_parts * 64u)) * 64u;
    uint32_t part = (n_parts > 1u) ? (__builtin_amdgcn_s_getreg(6164 /* HW_REG_XCC_ID, bits [3:0] */) & (n_parts - 1u)) : 0u;
    uint32_t parts_left = n_parts;
    for (;;) {
        // ---- refill idle lanes from the wave-private chunk [w_next, w_end); one atomic per `chunk` rays ----
        unsigned long long idle = __ballot(state == ST_IDLE);
        if (idle != 0ull) {
            int n_idle = __popcll(idle);
            if (!exhausted && (n_idle >= refill_min || idle == ~0ull)) {
                while (w_next >= w_end && !exhausted) {
                    const uint32_t p_begin = part * part_size;
                    const uint32_t p_end = (p_begin < n) ? ((n - p_begin < part_size) ? n : p_begin + part_size) : p_begin;
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(head + part * 32u, chunk);
                    base = __shfl(base, 0);
                    if (base < p_end - p_begin) {
                        w_next = p_begin + base;
                        w_end = (p_end - w_next < chunk) ? p_end : w_next + chunk;
                    } else {
                        part = (part + 1u == n_parts) ? 0u : part + 1u;
                        if (--parts_left == 0u) exhausted = true;
                    }
                }
                if (!exhausted) {
                    uint32_t take = min((uint32_t)n_idle, w_end - w_next);
                    if (state == ST_IDLE) {
                        uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                        if (rank < take) {
                            uint32_t qi = w_next + rank;
                            path = queue ? queue[qi] : qi;
                            const float4* rp = reinterpret_cast<const float4*>(rays + path);
                            float4 r0 = rp[0], r1 = rp[1];
                            ro = v3(r0.x, r0.y, r0.z);
                            V3 rd = v3(r0.w, r1.x, r1.y);
                            t_max = r1.z;
                            inv_dir = v3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);  // aggregate.rs:76-81
                            negx = inv_dir.x < 0.0f;
                            negy = inv_dir.y < 0.0f;
                            negz = inv_dir.z < 0.0f;
                            rs = ray_shear(rd);
                            if (!TRI_ONLY) rd_full = rd;
                            hit_prim = -1;
                            hit_inst = -1;
                            inst_slot = -1;
                            sp = 0;
                            cur = 0;
                            want_pop = false;
                            state = ST_NODE;
                            c_rays++;
                        }
                    }
                    w_next += take;
                }
            }
            if (__ballot(state != ST_IDLE) == 0ull) {
                if (exhausted) break;
                continue;  // private chunk was empty: fetch the next one
            }
        }
        // ---- one uniform node step ----
        if (state == ST_NODE) {
            bool go = true;
            if (want_pop) {
                want_pop = false;
                if (sp == 0) { state = ST_DONE; go = false; }
                else {
                    sp--;
                    // two different load flavours, so that the compiler cannot merge them into one flat_load of a selected
                    // pointer (which waits on both the LDS and the vector-memory counter)
                    if (sp < K3_LDS_N) cur = st_lds[sp * WAVE];
                    else cur = __builtin_nontemporal_load(st_spill + (size_t)(sp - K3_LDS_N) * WAVE);
                    if (!TRI_ONLY && cur == INST_SENTINEL) {
                        // the instanced aggregate is exhausted: back to the ray of the enclosing tree (primitive.rs:158-171 returns);
                        // t_max is the hit found inside (in the instance's parameterisation, as the reference keeps it) or what it was
                        const float4* rp = reinterpret_cast<const float4*>(rays + path);
                        float4 r0 = rp[0], r1 = rp[1];
                        ro = v3(r0.x, r0.y, r0.z);
                        V3 rd = v3(r0.w, r1.x, r1.y);
                        inv_dir = v3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
                        negx = inv_dir.x < 0.0f;
                        negy = inv_dir.y < 0.0f;
                        negz = inv_dir.z < 0.0f;
                        rs = ray_shear(rd);
                        rd_full = rd;
                        if (!inst_hit) t_max = t_outer;
                        inst_slot = -1;
                        want_pop = true;
                        go = false;
                    }
                }
            }
            if (go) {
                const float4* np = reinterpret_cast<const float4*>(node_base + ((size_t)cur << 5));
                float4 na = np[0], nb = np[1];
                c_nodes++;
                // Bounds3f::intersect_p_cached (bounding_box.rs:520-563), signs held as lane masks
                const Float g = 1.0f + 2.0f * gamma(3);
                Float t0 = ((negx ? na.w : na.x) - ro.x) * inv_dir.x;
                Float t1 = ((negx ? na.x : na.w) - ro.x) * inv_dir.x;
                Float ty0 = ((negy ? nb.x : na.y) - ro.y) * inv_dir.y;
                Float ty1 = ((negy ? na.y : nb.x) - ro.y) * inv_dir.y;
                t1 *= g;
                ty1 *= g;
                bool hit_box = !(t0 > ty1 || ty0 > t1);
                if (ty0 > t0) t0 = ty0;
                if (ty1 < t1) t1 = ty1;
                Float tz0 = ((negz ? nb.y : na.z) - ro.z) * inv_dir.z;
                Float tz1 = ((negz ? na.z : nb.y) - ro.z) * inv_dir.z;
                tz1 *= g;
                hit_box = hit_box && !(t0 > tz1 || tz0 > t1);
                if (tz0 > t0) t0 = tz0;
                if (tz1 < t1) t1 = tz1;
                hit_box = hit_box && (t0 < t_max) && (t1 > 0.0f);
                uint32_t offset = __float_as_uint(nb.z);
                uint32_t meta = __float_as_uint(nb.w);
                uint32_t n_prims = meta & 0xffffu;
                if (!hit_box) {
                    want_pop = true;
                } else if (n_prims > 0) {
                    leaf_off = offset;
                    leaf_n = n_prims;
                    state = ST_LEAF;
                } else {
                    uint32_t axis = (meta >> 16) & 0xffu;
                    bool neg = (axis == 0) ? negx : ((axis == 1) ? negy : negz);
                    uint32_t far_child = neg ? cur + 1 : offset;   // aggregate.rs:119-127
                    uint32_t near_child = neg ? offset : cur + 1;
                    if (sp < K3_LDS_N) st_lds[sp * WAVE] = far_child;
                    else st_spill[(size_t)(sp - K3_LDS_N) * WAVE] = far_child;
                    sp++;
                    cur = near_child;
                }
            }
        }
        // ---- postponed leaf phase ----
        unsigned long long leaf_mask = __ballot(state == ST_LEAF);
        if (leaf_mask != 0ull) {
            unsigned long long node_mask = __ballot(state == ST_NODE);
            if (__popcll(leaf_mask) >= leaf_min || node_mask == 0ull) {
                if (state == ST_LEAF) {
                    bool found_any = false;
                    bool entered = false;
                    for (uint32_t i = 0; i < leaf_n; ++i) {
                        uint32_t slot = leaf_off + i;
                        c_prims++;
                        const float4* pr = reinterpret_cast<const float4*>(prim_base + (size_t)slot * 48u);
                        float4 q0 = pr[0], q1 = pr[1], q2 = pr[2];
                        bool got;
                        if (!TRI_ONLY && (__float_as_uint(q2.y) & PRIM_INSTANCE_BIT)) {
                            // TransformedPrimitive (primitive.rs:158-176; alone in its leaf, flatten.h): leave a marker on the stack, take
                            // the ray into the instance's space — apply_ray_inverse for intersect, the FORWARD apply_ray for
                            // intersect_predicate, as the reference writes them — and go on in the instanced aggregate's tree.
                            const ShmInstance& in = sv.instances[__float_as_uint(q2.y) & PRIM_INDEX_MASK];
                            if (sp < K3_LDS_N) st_lds[sp * WAVE] = INST_SENTINEL;
                            else st_spill[(size_t)(sp - K3_LDS_N) * WAVE] = INST_SENTINEL;
                            sp++;
                            t_outer = t_max;
                            inst_slot = (int32_t)slot;
                            inst_hit = false;
                            Ray r;
                            if (ANY) { Ray w; w.o = ro; w.d = rd_full; r = xf_ray(in.render_from_primitive, w); }
                            else r = xf_ray_inverse(in.primitive_from_render, ro, rd_full, t_max);
                            ro = r.o;
                            rd_full = r.d;
                            inv_dir = v3(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
                            negx = inv_dir.x < 0.0f;
                            negy = inv_dir.y < 0.0f;
                            negz = inv_dir.z < 0.0f;
                            rs = ray_shear(r.d);
                            cur = in.root_node;
                            entered = true;
                            break;
                        }
                        if (!TRI_ONLY && (__float_as_uint(q2.y) & PRIM_SPHERE_BIT)) {
                            // Sphere::intersect (sphere.rs:95-196) through the same leaf phase; the hit record carries p_obj and phi
                            QuadricIntersection qi;
                            got = sphere_basic_intersect(sv.spheres[__float_as_uint(q2.y) & PRIM_INDEX_MASK], ro, rd_full, t_max, qi);
                            if (got) { hit_prim = (int32_t)slot; hit_t = qi.t_hit; hit_b0 = qi.p_obj.x; hit_b1 = qi.p_obj.y; hit_b2 = qi.p_obj.z; hit_phi = qi.phi; }
                        } else if (!TRI_ONLY && (__float_as_uint(q2.y) & PRIM_PATCH_BIT)) {
                            // BilinearPatch::intersect (bilinear_patch.rs:144-236): the record holds p00, p10, p01; (u, v) go in b0, b1
                            BilinearIntersection bi;
                            got = blp_intersect(ro, rd_full, t_max, v3(q0.x, q0.y, q0.z), v3(q0.w, q1.x, q1.y), v3(q1.z, q1.w, q2.x),
                                                ld3(sv.patches[__float_as_uint(q2.y) & PRIM_INDEX_MASK].p11), bi);
                            if (got) { hit_prim = (int32_t)slot; hit_t = bi.t; hit_b0 = bi.u; hit_b1 = bi.v; hit_b2 = 0.0f; hit_phi = 0.0f; }
                        } else {
                            TriangleIntersection ti;
                            got = intersect_triangle_pre(ro, rs, t_max, v3(q0.x, q0.y, q0.z), v3(q0.w, q1.x, q1.y), v3(q1.z, q1.w, q2.x), ti);
                            if (got) { hit_prim = (int32_t)slot; hit_t = ti.t; hit_b0 = ti.b0; hit_b1 = ti.b1; hit_b2 = ti.b2; hit_phi = 0.0f; }
                        }
                        if (got) {
                            if (ANY) { found_any = true; break; }
                            t_max = hit_t;  // aggregate.rs:105-109
                            if (!TRI_ONLY) { hit_inst = inst_slot; inst_hit = true; }
                        }
                    }
                    if (!TRI_ONLY && entered) {
                        state = ST_NODE;
                        want_pop = false;
                    } else if (ANY && found_any) {
                        state = ST_DONE;
                    } else {
                        state = ST_NODE;
                        want_pop = true;
                    }
                }
            }
        }
        // ---- retire finished rays ----
        if (state == ST_DONE) {
            if (ANY) {
                bool occl = hit_prim >= 0;
                if (occluded_out) occluded_out[path] = occl ? 1 : 0;
                if (L && !occl) {
                    float4 l = L[path], c = contrib[path];
                    l.x += c.x; l.y += c.y; l.z += c.z; l.w += c.w;
                    L[path] = l;
                }
            } else {
                float4* hp = reinterpret_cast<float4*>(hits + path);
                hp[0] = make_float4(__int_as_float(hit_prim), hit_t, hit_b0, hit_b1);
                hp[1] = make_float4(hit_b2, TRI_ONLY ? 0.0f : hit_phi, TRI_ONLY ? 0.0f : __int_as_float(hit_inst + 1), 0.0f);
            }
            state = ST_IDLE;
        }
    }
    unsigned long long w_nodes = c_nodes, w_prims = c_prims, w_rays = c_rays;
    for (int off = 32; off > 0; off >>= 1) {
        w_nodes += __shfl_down(w_nodes, off);
        w_prims += __shfl_down(w_prims, off);
        w_rays += __shfl_down(w_rays, off);
    }
    if (lane == 0 && w_rays) {
        if (ANY) {
            atomicAdd(&counters->rays_any, w_rays);
            atomicAdd(&counters->nodes_any, w_nodes);
            atomicAdd(&counters->tris_any, w_prims);
        } else {
            atomicAdd(&counters->rays_closest, w_rays);
            atomicAdd(&counters->nodes_closest, w_nodes);
            atomicAdd(&counters->tris_closest, w_prims);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K4+K5: one path vertex (integrator.rs:772-892 for the vertex found by K2).
// ---------------------------------------------------------------------------------------------
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
// HAS_LAYERED = false is the instantiation for scenes without Coated* materials: the LayeredBxDF random walks (three per
// vertex: f and pdf for NEE, sample_f) are compiled out of it.
//   TRI_ONLY = true is the instantiation for scenes made of triangles only: no quadric / bilinear-patch interaction and light
//   sampling code (with it the kernel needs 264 VGPRs, one wave per SIMD; without it 243, two waves).
//   HAS_TEX = true is the instantiation for scenes that bind image textures: the path carries ray differentials (texture.h),
//   get_bsdf filters the MIP pyramids. One general instantiation <true, false, true>.
//   DIFFUSE_ONLY = true is the instantiation for scenes whose materials are all DiffuseMaterial (the headline scene class): the
//   conductor / dielectric BxDFs and the material dispatch are compiled out of it.
template <bool HAS_LAYERED, bool TRI_ONLY, bool HAS_TEX = false, bool DIFFUSE_ONLY = false>
__global__ void __launch_bounds__(SHADE2_BLOCK) K_SHADE_ATTR k_shade(SceneView sv, PathArrays pa, const uint32_t* __restrict__ q_cur, uint32_t* __restrict__ q_next,
                                                     uint32_t* __restrict__ q_shadow, QueueState* qs, int cur, ShmRenderParams params,
                                                     DeviceCounters* counters, int shadow_parity) {
    const uint32_t n = qs->n_active[cur];
    __shared__ uint32_t s_next[SHADE_CHUNK], s_shadow[SHADE_CHUNK];
    __shared__ uint32_t s_cnt[2], s_base[2];
    for (uint32_t chunk0 = blockIdx.x * SHADE_CHUNK; chunk0 < n; chunk0 += gridDim.x * SHADE_CHUNK) {
      if (threadIdx.x == 0) { s_cnt[0] = 0; s_cnt[1] = 0; }
      __syncthreads();
      for (uint32_t k = 0; k < SHADE_CHUNK / SHADE2_BLOCK; ++k) {
        const uint32_t i = chunk0 + k * SHADE2_BLOCK + threadIdx.x;
        bool active = i < n;
        bool push_next = false, push_shadow = false;
        uint32_t path = 0;
        if (active) {
            path = q_cur[i];
            const float4* hp = reinterpret_cast<const float4*>(pa.hit + path);
            float4 h0 = hp[0], h1 = hp[1];
            Hit hit;
            hit.prim = __float_as_int(h0.x); hit.t = h0.y; hit.b0 = h0.z; hit.b1 = h0.w; hit.b2 = h1.x; hit.phi = h1.y; hit.inst = __float_as_int(h1.z) - 1;
            const float4* rp = reinterpret_cast<const float4*>(pa.ray + path);
            float4 r0 = rp[0], r1 = rp[1];
            V3 ray_d = v3(r0.w, r1.x, r1.y);
            // L is only touched by a vertex that adds emission (most do not): loaded and stored inside add_l
            auto add_l = [&](const Spec& c) { pa.L[path] = st_spec(ld_spec(pa.L[path]) + c); };
            Spec beta = ld_spec(pa.beta[path]);
            Wavelengths lambda;
            float4 pdf_in;
            {
                float4 a = pa.lambda[path], b = pa.lambda_pdf[path];
                pdf_in = b;
                lambda.lambda[0] = a.x; lambda.lambda[1] = a.y; lambda.lambda[2] = a.z; lambda.lambda[3] = a.w;
                lambda.pdf[0] = b.x; lambda.pdf[1] = b.y; lambda.pdf[2] = b.z; lambda.pdf[3] = b.w;
            }
            uint32_t fl = pa.flags[path];
            int depth = (int)(fl & 0xffu);
            bool specular_bounce = (fl >> 8) & 1u;
            bool any_non_specular_bounces = (fl >> 9) & 1u;
            float2 pe = pa.pb_eta[path];
            Float p_b = pe.x, eta_scale = pe.y;
            // the previous vertex's context is only needed for the MIS weight of an emitter that was hit
            auto load_prev_ctx = [&]() {
                LightSampleContext c;
                float4 c0 = pa.ctx0[path], c1 = pa.ctx1[path], c2 = pa.ctx2[path];
                c.pi.x = iv2(c0.x, c0.w);
                c.pi.y = iv2(c0.y, c1.x);
                c.pi.z = iv2(c0.z, c1.y);
                c.n = v3(c1.z, c1.w, c2.x);
                c.ns = v3(c2.y, c2.z, c2.w);
                return c;
            };
            if (hit.prim < 0) {
                // integrator.rs:776-794: escaped ray, infinite lights
                for (uint32_t k = 0; k < sv.n_infinite_lights; ++k) {
                    const ShmLight& light = sv.lights[sv.infinite_lights[k]];
                    Spec le = infinite_light_le<HAS_TEX>(sv, light, ray_d, lambda);
                    if (depth == 0 || specular_bounce) {
                        add_l(beta * le);
                    } else {
                        Float p_l = light_sampler_pmf(sv) * light_pdf_li<TRI_ONLY, HAS_TEX>(sv, light, load_prev_ctx(), ray_d);
                        Float w_b = power_heuristic(1, p_b, 1, p_l);
                        add_l(beta * w_b * le);
                    }
                }
            } else {
                SurfaceInteraction si = hit_interaction<TRI_ONLY>(sv, hit, -ray_d);
                const ShmPrimitive prim = sv.primitives[hit.prim];
                // integrator.rs:798-813: emission at the hit
                if (prim.area_light >= 0) {
                    const ShmLight& light = sv.lights[prim.area_light];
                    Spec le = area_light_l(sv, light, si.n, -ray_d, lambda);
                    if (!is_zero(le)) {
                        if (depth == 0 || specular_bounce) {
                            add_l(beta * le);
                        } else {
                            Float p_l = light_sampler_pmf(sv) * light_pdf_li<TRI_ONLY, HAS_TEX>(sv, light, load_prev_ctx(), ray_d);
                            Float w_l = power_heuristic(1, p_b, 1, p_l);
                            add_l(beta * w_l * le);
                        }
                    }
                }
                // get_bsdf starts with compute_differentials(ray, camera, spp) (interaction.rs:197)
                Differentials df;
                AuxRays aux = aux_none();
                if (HAS_TEX) {
                    if (fl & (1u << 10)) aux = ld_aux(pa, path);
                    df = compute_differentials(sv, si, aux, params.samples_per_pixel, params.disable_pixel_jitter != 0, params.disable_texture_filtering != 0);
                }
                const ShmMaterial& mat = sv.materials[prim.material];
                if (DIFFUSE_ONLY) __builtin_assume(mat.kind == SHM_MATERIAL_DIFFUSE);
                BSDF bsdf = get_bsdf<HAS_TEX>(sv, si, mat, lambda, &df);
                if (!HAS_LAYERED) __builtin_assume(bsdf.bxdf.kind <= SHM_MATERIAL_THIN_DIELECTRIC);
                if (DIFFUSE_ONLY) __builtin_assume(bsdf.bxdf.kind == SHM_MATERIAL_DIFFUSE);
                Rng rng;
                auto load_rng = [&]() {
                    uint32_t pix = pa.pixel[path];
                    uint2 rs = pa.rng[path];
                    rng.state = (uint64_t)rs.x | ((uint64_t)rs.y << 32);
                    // inc is a pure function of (pixel, seed): re-derive instead of storing 8 more bytes per path
                    uint64_t h = mix_bits(((uint64_t)(pix & 0xffffu) << 32) | (uint64_t)(pix >> 16));
                    h = mix_bits(h ^ (params.seed + 0x9e3779b97f4a7c15ULL));
                    rng.inc = (h << 1u) | 1u;
                };
                // options.force_diffuse (interaction.rs:256-275) draws inside get_bsdf, before the depth test; only the general
                // instantiation carries it (the host launches that one when the flag is set)
                const bool forced = HAS_TEX && params.force_diffuse != 0;
                if (forced) {
                    load_rng();
                    Float uc = sampler_get_1d(rng);
                    V2 u2f = sampler_get_2d(rng);
                    bsdf_force_diffuse(bsdf, si.wo, uc, u2f);
                }
                if (params.regularize && any_non_specular_bounces) bxdf_regularize(bsdf.bxdf);
                bool alive = (depth != params.max_depth);  // integrator.rs:830-834
                if (alive) {
                    depth += 1;
                    if (!forced) load_rng();
                    // integrator.rs:837-841 + 897-963: next-event estimation; the visibility test is deferred to K3
                    if (flags_is_non_specular(bsdf_flags(bsdf))) {
                        LightSampleContext ctx = light_ctx_from(si);
                        uint32_t bf = bsdf_flags(bsdf);
                        if (flags_is_reflective(bf) && !flags_is_transmissive(bf)) ctx.pi = p3i_exact(offset_ray_origin(si.pi, si.n, si.wo));
                        else if (flags_is_transmissive(bf) && !flags_is_reflective(bf)) ctx.pi = p3i_exact(offset_ray_origin(si.pi, si.n, -si.wo));
                        Float u = sampler_get_1d(rng);
                        Float p_sel = 0.0f;
                        int li = light_sampler_sample(sv, u, p_sel);
                        V2 u_light = sampler_get_2d(rng);
                        if (li >= 0) {
                            const ShmLight& light = sv.lights[li];
                            LightLiSample ls;
                            if (light_sample_li<TRI_ONLY, HAS_TEX>(sv, light, ctx, u_light, lambda, ls) && !(is_zero(ls.l) || ls.pdf == 0.0f)) {
                                V3 wo = si.wo;
                                V3 wi = ls.wi;
                                Spec f = bsdf_f(bsdf, wo, wi) * abs_dot(wi, si.shading.n);
                                if (!is_zero(f)) {
                                    Ray sr = spawn_ray_to_both_offset(si.pi, si.n, ls.p_light_pi, ls.p_light_n);
                                    Float p_l = p_sel * ls.pdf;
                                    Spec ld;
                                    if (light_is_delta(light)) {
                                        ld = ls.l * f / p_l;
                                    } else {
                                        Float pb2 = bsdf_pdf(bsdf, wo, wi, REFLTRANS_ALL);
                                        Float w_l = power_heuristic(1, p_l, 1, pb2);
                                        ld = w_l * ls.l * f / p_l;
                                    }
                                    ShmRay s;
                                    s.o[0] = sr.o.x; s.o[1] = sr.o.y; s.o[2] = sr.o.z;
                                    s.d[0] = sr.d.x; s.d[1] = sr.d.y; s.d[2] = sr.d.z;
                                    s.t_max = 1.0f - 0.0001f;  // 1 - SHADOW_EPSILON, integrator.rs:66,115
                                    s.pad = 0.0f;
                                    pa.shadow_ray[path] = s;
                                    pa.shadow_contrib[path] = st_spec(beta * ld);
                                    push_shadow = true;
                                }
                            }
                        }
                    }
                    // integrator.rs:843-857: sample the BSDF
                    V3 wo = -ray_d;
                    Float u = sampler_get_1d(rng);
                    V2 u2 = sampler_get_2d(rng);
                    BSDFSample bs;
                    if (!bsdf_sample_f(bsdf, wo, u, u2, REFLTRANS_ALL, bs)) {
                        alive = false;
                    } else {
                        // integrator.rs:859-872
                        beta = beta * (bs.f * abs_dot(bs.wi, si.shading.n) / bs.pdf);
                        p_b = bs.pdf_is_proportional ? bsdf_pdf(bsdf, wo, bs.wi, REFLTRANS_ALL) : bs.pdf;
                        specular_bounce = flags_is_specular(bs.flags);
                        any_non_specular_bounces |= !specular_bounce;
                        if (flags_is_transmissive(bs.flags)) eta_scale *= sqr(bs.eta);
                        LightSampleContext nctx = light_ctx_from(si);
                        V3 no = offset_ray_origin(si.pi, si.n, bs.wi);  // integrator.rs:875 -> interaction.rs:68-75
                        // integrator.rs:878-891: Russian roulette
                        if (is_finite(eta_scale)) {
                            Spec rr_beta = beta * eta_scale;
                            if (max_component_value(rr_beta) < 1.0f && depth > 1) {
                                Float q = max(0.0f, 1.0f - max_component_value(rr_beta));
                                if (sampler_get_1d(rng) < q) alive = false;
                                else beta = beta / (1.0f - q);
                            }
                        }
                        if (alive) {
                            ShmRay nr;
                            nr.o[0] = no.x; nr.o[1] = no.y; nr.o[2] = no.z;
                            nr.d[0] = bs.wi.x; nr.d[1] = bs.wi.y; nr.d[2] = bs.wi.z;
                            nr.t_max = infinity();
                            nr.pad = 0.0f;
                            pa.ray[path] = nr;
                            pa.beta[path] = st_spec(beta);
                            pa.pb_eta[path] = make_float2(p_b, eta_scale);
                            pa.ctx0[path] = make_float4(nctx.pi.x.low, nctx.pi.y.low, nctx.pi.z.low, nctx.pi.x.high);
                            pa.ctx1[path] = make_float4(nctx.pi.y.high, nctx.pi.z.high, nctx.n.x, nctx.n.y);
                            pa.ctx2[path] = make_float4(nctx.n.z, nctx.ns.x, nctx.ns.y, nctx.ns.z);
                            pa.rng[path] = make_uint2((uint32_t)rng.state, (uint32_t)(rng.state >> 32));
                            uint32_t aux_bit = 0u;
                            if (HAS_TEX) {  // spawn_ray_with_differentials, interaction.rs:430-514
                                AuxRays na = spawn_ray_differentials(si, df, aux, bs.wi, bs.flags, bs.eta);
                                if (na.has) { st_aux(pa, path, na); aux_bit = 1u << 10; }
                            }
                            pa.flags[path] = (uint32_t)depth | ((uint32_t)specular_bounce << 8) | ((uint32_t)any_non_specular_bounces << 9) | aux_bit;
                            push_next = true;
                        }
                    }
                }
                // terminate_secondary may have changed the pdfs (material.rs:609-619): written back only then
                if (lambda.pdf[1] != pdf_in.y || lambda.pdf[2] != pdf_in.z || lambda.pdf[3] != pdf_in.w || lambda.pdf[0] != pdf_in.x)
                    pa.lambda_pdf[path] = make_float4(lambda.pdf[0], lambda.pdf[1], lambda.pdf[2], lambda.pdf[3]);
            }
        }
        // stage the queue entries of this chunk in LDS (wave-aggregated LDS atomics)
        uint32_t s1 = queue_push_slot(&s_cnt[0], push_next);
        if (push_next) s_next[s1] = path;
        uint32_t s2 = queue_push_slot(&s_cnt[1], push_shadow);
        if (push_shadow) s_shadow[s2] = path;
      }
      __syncthreads();
      if (threadIdx.x == 0) {
          s_base[0] = s_cnt[0] ? atomicAdd(&qs->n_active[cur ^ 1], s_cnt[0]) : 0u;
          s_base[1] = s_cnt[1] ? atomicAdd(&qs->n_shadow[shadow_parity], s_cnt[1]) : 0u;
      }
      __syncthreads();
      for (uint32_t j = threadIdx.x; j < s_cnt[0]; j += SHADE2_BLOCK) q_next[s_base[0] + j] = s_next[j];
      for (uint32_t j = threadIdx.x; j < s_cnt[1]; j += SHADE2_BLOCK) q_shadow[s_base[1] + j] = s_shadow[j];
      __syncthreads();
    }
    (void)counters;
}

// get_bsdf for the two general (non-throughput) integrator kernels below: with image textures bound it starts with
// compute_differentials; only camera rays carry auxiliary rays there (every later ray is interaction.spawn_ray(wi),
// integrator.rs:547, 686, 716), flag bit 10 as in k_shade.
__device__ BSDF get_bsdf_general(const SceneView& sv, const PathArrays& pa, uint32_t path, uint32_t fl, SurfaceInteraction& si,
                                 const ShmMaterial& m, Wavelengths& lambda, const ShmRenderParams& params) {
    if (pa.aux0 == nullptr) return get_bsdf(sv, si, m, lambda);
    AuxRays aux = (fl & (1u << 10)) ? ld_aux(pa, path) : aux_none();
    Differentials df = compute_differentials(sv, si, aux, params.samples_per_pixel, params.disable_pixel_jitter != 0, params.disable_texture_filtering != 0);
    return get_bsdf<true>(sv, si, m, lambda, &df);
}

// ---------------------------------------------------------------------------------------------
// SimplePathIntegrator::li (integrator.rs:586-733), one vertex per launch, same queues and path state as k_shade. One general
// instantiation (every material and shape kind): it is the reference's debugging integrator, not a throughput path.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(SHADE2_BLOCK) K_SHADE_ATTR k_shade_simple(SceneView sv, PathArrays pa, const uint32_t* __restrict__ q_cur,
                                                                           uint32_t* __restrict__ q_next, uint32_t* __restrict__ q_shadow,
                                                                           QueueState* qs, int cur, ShmRenderParams params, int shadow_parity) {
    const uint32_t n = qs->n_active[cur];
    const bool sample_lights = params.sample_lights != 0, sample_bsdf = params.sample_bsdf != 0;
    __shared__ uint32_t s_next[SHADE_CHUNK], s_shadow[SHADE_CHUNK];
    __shared__ uint32_t s_cnt[2], s_base[2];
    for (uint32_t chunk0 = blockIdx.x * SHADE_CHUNK; chunk0 < n; chunk0 += gridDim.x * SHADE_CHUNK) {
      if (threadIdx.x == 0) { s_cnt[0] = 0; s_cnt[1] = 0; }
      __syncthreads();
      for (uint32_t k = 0; k < SHADE_CHUNK / SHADE2_BLOCK; ++k) {
        const uint32_t i = chunk0 + k * SHADE2_BLOCK + threadIdx.x;
        bool push_next = false, push_shadow = false;
        uint32_t path = 0;
        if (i < n) {
            path = q_cur[i];
            const float4* hp = reinterpret_cast<const float4*>(pa.hit + path);
            float4 h0 = hp[0], h1 = hp[1];
            Hit hit;
            hit.prim = __float_as_int(h0.x); hit.t = h0.y; hit.b0 = h0.z; hit.b1 = h0.w; hit.b2 = h1.x; hit.phi = h1.y; hit.inst = __float_as_int(h1.z) - 1;
            const float4* rp = reinterpret_cast<const float4*>(pa.ray + path);
            float4 r0 = rp[0], r1 = rp[1];
            V3 ray_d = v3(r0.w, r1.x, r1.y);
            auto add_l = [&](const Spec& c) { pa.L[path] = st_spec(ld_spec(pa.L[path]) + c); };
            Spec beta = ld_spec(pa.beta[path]);
            Wavelengths lambda;
            float4 pdf_in;
            {
                float4 a = pa.lambda[path], b = pa.lambda_pdf[path];
                pdf_in = b;
                lambda.lambda[0] = a.x; lambda.lambda[1] = a.y; lambda.lambda[2] = a.z; lambda.lambda[3] = a.w;
                lambda.pdf[0] = b.x; lambda.pdf[1] = b.y; lambda.pdf[2] = b.z; lambda.pdf[3] = b.w;
            }
            uint32_t fl = pa.flags[path];
            int depth = (int)(fl & 0xffu);
            // k_generate leaves flags = 0: the reference starts with specular_bounce = true (integrator.rs:601), so bit 8 holds its
            // negation here ("the last bounce was NOT specular")
            bool specular_bounce = ((fl >> 8) & 1u) == 0u;
            if (hit.prim < 0) {
                if (!sample_lights || specular_bounce)
                    for (uint32_t li = 0; li < sv.n_infinite_lights; ++li) {
                        const ShmLight& light = sv.lights[sv.infinite_lights[li]];
                        add_l(beta * infinite_light_le<true>(sv, light, ray_d, lambda));
                    }
            } else {
                SurfaceInteraction si = hit_interaction<false>(sv, hit, -ray_d);
                const ShmPrimitive prim = sv.primitives[hit.prim];
                if (!sample_lights || specular_bounce) {
                    if (prim.area_light >= 0) add_l(beta * area_light_l(sv, sv.lights[prim.area_light], si.n, -ray_d, lambda));
                    else add_l(beta * spec_const(0.0f));  // isect.le() of a non-emitter: a zero spectrum that is still added
                }
                if (depth != params.max_depth) {
                    depth += 1;
                    BSDF bsdf = get_bsdf_general(sv, pa, path, fl, si, sv.materials[prim.material], lambda, params);
                    V3 wo = -ray_d;
                    uint32_t pix = pa.pixel[path];
                    uint2 rs = pa.rng[path];
                    Rng rng;
                    rng.state = (uint64_t)rs.x | ((uint64_t)rs.y << 32);
                    {
                        uint64_t h = mix_bits(((uint64_t)(pix & 0xffffu) << 32) | (uint64_t)(pix >> 16));
                        h = mix_bits(h ^ (params.seed + 0x9e3779b97f4a7c15ULL));
                        rng.inc = (h << 1u) | 1u;
                    }
                    if (params.force_diffuse) {  // interaction.rs:256-275: rho_hd(wo, [get_1d()], [get_2d()]) inside get_bsdf
                        Float uc = sampler_get_1d(rng);
                        V2 u2f = sampler_get_2d(rng);
                        bsdf_force_diffuse(bsdf, si.wo, uc, u2f);
                    }
                    if (sample_lights) {
                        Float p_sel = 0.0f;
                        int li = light_sampler_sample(sv, sampler_get_1d(rng), p_sel);
                        if (li >= 0) {
                            V2 u_light = sampler_get_2d(rng);
                            LightSampleContext ctx = light_ctx_from(si);
                            const ShmLight& light = sv.lights[li];
                            LightLiSample ls;
                            if (light_sample_li<false, true>(sv, light, ctx, u_light, lambda, ls, false) && !is_zero(ls.l) && ls.pdf > 0.0f) {
                                V3 wi = ls.wi;
                                Spec f = bsdf_f(bsdf, wo, wi) * abs_dot(wi, si.shading.n);
                                if (!is_zero(f)) {
                                    Ray sr = spawn_ray_to_both_offset(si.pi, si.n, ls.p_light_pi, ls.p_light_n);
                                    ShmRay sh;
                                    sh.o[0] = sr.o.x; sh.o[1] = sr.o.y; sh.o[2] = sr.o.z;
                                    sh.d[0] = sr.d.x; sh.d[1] = sr.d.y; sh.d[2] = sr.d.z;
                                    sh.t_max = 1.0f - 0.0001f;  // 1 - SHADOW_EPSILON
                                    sh.pad = 0.0f;
                                    pa.shadow_ray[path] = sh;
                                    pa.shadow_contrib[path] = st_spec(beta * f * ls.l / (p_sel * ls.pdf));  // added to L by K3 if unoccluded
                                    push_shadow = true;
                                }
                            }
                        }
                    }
                    bool alive = true;
                    V3 wi_next = v3s(0.0f);
                    if (sample_bsdf) {
                        Float u = sampler_get_1d(rng);
                        V2 u2 = sampler_get_2d(rng);
                        BSDFSample bs;
                        if (!bsdf_sample_f(bsdf, wo, u, u2, REFLTRANS_ALL, bs)) {
                            alive = false;
                        } else {
                            beta = beta * (bs.f * abs_dot(bs.wi, si.shading.n) / bs.pdf);
                            specular_bounce = flags_is_specular(bs.flags);
                            wi_next = bs.wi;
                        }
                    } else {
                        uint32_t flags = bsdf_flags(bsdf);
                        Float pdf;
                        if (flags_is_reflective(flags) && flags_is_transmissive(flags)) {
                            wi_next = sample_uniform_sphere(sampler_get_2d(rng));
                            pdf = uniform_sphere_pdf();
                        } else {
                            wi_next = sample_uniform_hemisphere(sampler_get_2d(rng));
                            pdf = uniform_hemisphere_pdf();
                            if ((flags_is_reflective(flags) && dot(wo, si.n) * dot(wi_next, si.n) < 0.0f) ||
                                (flags_is_transmissive(flags) && dot(wo, si.n) * dot(wi_next, si.n) > 0.0f))
                                wi_next = -wi_next;
                        }
                        beta = beta * (bsdf_f(bsdf, wo, wi_next) * abs_dot(wi_next, si.shading.n) / pdf);
                        specular_bounce = false;
                    }
                    if (alive && !is_zero(beta)) {  // `while !beta.is_zero()` at the top of the next iteration
                        V3 no = offset_ray_origin(si.pi, si.n, wi_next);
                        ShmRay nr;
                        nr.o[0] = no.x; nr.o[1] = no.y; nr.o[2] = no.z;
                        nr.d[0] = wi_next.x; nr.d[1] = wi_next.y; nr.d[2] = wi_next.z;
                        nr.t_max = infinity();
                        nr.pad = 0.0f;
                        pa.ray[path] = nr;
                        pa.beta[path] = st_spec(beta);
                        pa.rng[path] = make_uint2((uint32_t)rng.state, (uint32_t)(rng.state >> 32));
                        pa.flags[path] = (uint32_t)depth | ((specular_bounce ? 0u : 1u) << 8);
                        push_next = true;
                    }
                    if (lambda.pdf[1] != pdf_in.y || lambda.pdf[2] != pdf_in.z || lambda.pdf[3] != pdf_in.w || lambda.pdf[0] != pdf_in.x)
                        pa.lambda_pdf[path] = make_float4(lambda.pdf[0], lambda.pdf[1], lambda.pdf[2], lambda.pdf[3]);
                }
            }
        }
        uint32_t s1 = queue_push_slot(&s_cnt[0], push_next);
        if (push_next) s_next[s1] = path;
        uint32_t s2 = queue_push_slot(&s_cnt[1], push_shadow);
        if (push_shadow) s_shadow[s2] = path;
      }
      __syncthreads();
      if (threadIdx.x == 0) {
          s_base[0] = s_cnt[0] ? atomicAdd(&qs->n_active[cur ^ 1], s_cnt[0]) : 0u;
          s_base[1] = s_cnt[1] ? atomicAdd(&qs->n_shadow[shadow_parity], s_cnt[1]) : 0u;
      }
      __syncthreads();
      for (uint32_t j = threadIdx.x; j < s_cnt[0]; j += SHADE2_BLOCK) q_next[s_base[0] + j] = s_next[j];
      for (uint32_t j = threadIdx.x; j < s_cnt[1]; j += SHADE2_BLOCK) q_shadow[s_base[1] + j] = s_shadow[j];
      __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// RandomWalkIntegrator (integrator.rs:445-563). Its estimator L_k = le_k + f_k cos_k L_{k+1} / (1/4pi) is recursive and
// evaluated innermost-first there; a wavefront walks the path forwards, so each vertex records (le_k, f_k cos_k) in
// rw[depth][path] and k_fold_randomwalk evaluates the recursion backwards from the terminal vertex, in the reference's
// operation order. No light sampling, no shadow rays.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(SHADE2_BLOCK) K_SHADE_ATTR k_shade_randomwalk(SceneView sv, PathArrays pa, const uint32_t* __restrict__ q_cur,
                                                                               uint32_t* __restrict__ q_next, QueueState* qs, int cur,
                                                                               ShmRenderParams params, float4* __restrict__ rw, uint32_t capacity) {
    const uint32_t n = qs->n_active[cur];
    __shared__ uint32_t s_next[SHADE_CHUNK];
    __shared__ uint32_t s_cnt, s_base;
    for (uint32_t chunk0 = blockIdx.x * SHADE_CHUNK; chunk0 < n; chunk0 += gridDim.x * SHADE_CHUNK) {
      if (threadIdx.x == 0) s_cnt = 0;
      __syncthreads();
      for (uint32_t k = 0; k < SHADE_CHUNK / SHADE2_BLOCK; ++k) {
        const uint32_t i = chunk0 + k * SHADE2_BLOCK + threadIdx.x;
        bool push_next = false;
        uint32_t path = 0;
        if (i < n) {
            path = q_cur[i];
            const float4* hp = reinterpret_cast<const float4*>(pa.hit + path);
            float4 h0 = hp[0], h1 = hp[1];
            Hit hit;
            hit.prim = __float_as_int(h0.x); hit.t = h0.y; hit.b0 = h0.z; hit.b1 = h0.w; hit.b2 = h1.x; hit.phi = h1.y; hit.inst = __float_as_int(h1.z) - 1;
            const float4* rp = reinterpret_cast<const float4*>(pa.ray + path);
            float4 r0 = rp[0], r1 = rp[1];
            V3 ray_d = v3(r0.w, r1.x, r1.y);
            Wavelengths lambda;
            float4 pdf_in;
            {
                float4 a = pa.lambda[path], b = pa.lambda_pdf[path];
                pdf_in = b;
                lambda.lambda[0] = a.x; lambda.lambda[1] = a.y; lambda.lambda[2] = a.z; lambda.lambda[3] = a.w;
                lambda.pdf[0] = b.x; lambda.pdf[1] = b.y; lambda.pdf[2] = b.z; lambda.pdf[3] = b.w;
            }
            const uint32_t fl = pa.flags[path];
            const int depth = (int)(fl & 0xffu);
            float4* rec = rw + (size_t)(2 * depth) * capacity + path;  // le at 2*depth, f cos at 2*depth + 1
            Spec le = spec_const(0.0f);
            if (hit.prim < 0) {
                for (uint32_t li = 0; li < sv.n_infinite_lights; ++li) {
                    const ShmLight& light = sv.lights[sv.infinite_lights[li]];
                    le = le + infinite_light_le<true>(sv, light, ray_d, lambda);
                }
                rec[0] = st_spec(le);  // terminal vertex: flags keeps this depth
            } else {
                SurfaceInteraction si = hit_interaction<false>(sv, hit, -ray_d);
                const ShmPrimitive prim = sv.primitives[hit.prim];
                V3 wo = -ray_d;
                if (prim.area_light >= 0) le = area_light_l(sv, sv.lights[prim.area_light], si.n, wo, lambda);
                rec[0] = st_spec(le);
                if (depth != params.max_depth) {
                    BSDF bsdf = get_bsdf_general(sv, pa, path, fl, si, sv.materials[prim.material], lambda, params);
                    uint32_t pix = pa.pixel[path];
                    uint2 rs = pa.rng[path];
                    Rng rng;
                    rng.state = (uint64_t)rs.x | ((uint64_t)rs.y << 32);
                    {
                        uint64_t h = mix_bits(((uint64_t)(pix & 0xffffu) << 32) | (uint64_t)(pix >> 16));
                        h = mix_bits(h ^ (params.seed + 0x9e3779b97f4a7c15ULL));
                        rng.inc = (h << 1u) | 1u;
                    }
                    if (params.force_diffuse) {
                        Float uc = sampler_get_1d(rng);
                        V2 u2f = sampler_get_2d(rng);
                        bsdf_force_diffuse(bsdf, si.wo, uc, u2f);
                    }
                    V3 wp = sample_uniform_sphere(sampler_get_2d(rng));
                    Spec f = bsdf_f(bsdf, wo, wp);
                    if (!is_zero(f)) {
                        rec[capacity] = st_spec(f * abs_dot(wp, si.shading.n));
                        V3 no = offset_ray_origin(si.pi, si.n, wp);
                        ShmRay nr;
                        nr.o[0] = no.x; nr.o[1] = no.y; nr.o[2] = no.z;
                        nr.d[0] = wp.x; nr.d[1] = wp.y; nr.d[2] = wp.z;
                        nr.t_max = infinity();
                        nr.pad = 0.0f;
                        pa.ray[path] = nr;
                        pa.rng[path] = make_uint2((uint32_t)rng.state, (uint32_t)(rng.state >> 32));
                        pa.flags[path] = (uint32_t)(depth + 1);
                        push_next = true;
                    }
                    if (lambda.pdf[1] != pdf_in.y || lambda.pdf[2] != pdf_in.z || lambda.pdf[3] != pdf_in.w || lambda.pdf[0] != pdf_in.x)
                        pa.lambda_pdf[path] = make_float4(lambda.pdf[0], lambda.pdf[1], lambda.pdf[2], lambda.pdf[3]);
                }
            }
        }
        uint32_t s1 = queue_push_slot(&s_cnt, push_next);
        if (push_next) s_next[s1] = path;
      }
      __syncthreads();
      if (threadIdx.x == 0) s_base = s_cnt ? atomicAdd(&qs->n_active[cur ^ 1], s_cnt) : 0u;
      __syncthreads();
      for (uint32_t j = threadIdx.x; j < s_cnt; j += SHADE2_BLOCK) q_next[s_base + j] = s_next[j];
      __syncthreads();
    }
}
// L = le_T; L = le_k + f_k cos_k * L / (1 / (4 pi)) for k = T-1 .. 0 (integrator.rs:549-562)
__global__ void __launch_bounds__(SHADE_BLOCK) k_fold_randomwalk(PathArrays pa, const float4* __restrict__ rw, uint32_t capacity, uint32_t total) {
    uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= total) return;
    int t = (int)(pa.flags[slot] & 0xffu);
    Spec l = ld_spec(rw[(size_t)(2 * t) * capacity + slot]);
    for (int k = t - 1; k >= 0; --k) {
        Spec le = ld_spec(rw[(size_t)(2 * k) * capacity + slot]);
        Spec fcos = ld_spec(rw[(size_t)(2 * k + 1) * capacity + slot]);
        l = le + fcos * l / (1.0f / (4.0f * PI_F));
    }
    pa.L[slot] = st_spec(l);
}

// Between bounces: recycle the counters (1 thread).
__global__ void k_next_bounce(QueueState* qs, int cur, int next_shadow_parity) {
    qs->n_active[cur] = 0;
    qs->n_shadow[next_shadow_parity] = 0;  // the one the NEXT shade launch fills; this bounce's count stays for its K3
}
__global__ void k_reset_heads3(uint32_t* heads) { for (uint32_t i = threadIdx.x; i < 8 * 32; i += blockDim.x) heads[i] = 0; }

// ---------------------------------------------------------------------------------------------
// K6: RgbFilm::add_sample for every sample of the batch, per pixel in sample order (f64 sums are
// order dependent; the reference adds samples of a pixel in increasing sample_index).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(SHADE_BLOCK) k_film(SceneView sv, PathArrays pa, const uint32_t* pixels, uint32_t n_pix, int n_samples,
                                                    ShmFilmPixel* film, DeviceCounters* counters, uint32_t pix_group) {
    uint32_t p_local = blockIdx.x * blockDim.x + threadIdx.x;
    if (p_local >= n_pix) return;
    uint32_t pix = pixels[p_local];
    int px = (int)(pix & 0xffffu), py = (int)(pix >> 16);
    int width = sv.pixel_bounds[2] - sv.pixel_bounds[0];
    ShmFilmPixel* fp = film + (size_t)(py - sv.pixel_bounds[1]) * (size_t)width + (size_t)(px - sv.pixel_bounds[0]);
    double r = fp->rgb_sum[0], g = fp->rgb_sum[1], b = fp->rgb_sum[2], w = fp->weight_sum;
    for (int s = 0; s < n_samples; ++s) {
        uint32_t slot = slot_of(p_local, (uint32_t)s, n_pix, (uint32_t)n_samples, pix_group);
        Spec L = ld_spec(pa.L[slot]);
        Wavelengths lambda;
        float4 a = pa.lambda[slot], c = pa.lambda_pdf[slot];
        lambda.lambda[0] = a.x; lambda.lambda[1] = a.y; lambda.lambda[2] = a.z; lambda.lambda[3] = a.w;
        lambda.pdf[0] = c.x; lambda.pdf[1] = c.y; lambda.pdf[2] = c.z; lambda.pdf[3] = c.w;
        V3 rgb = film_sample_rgb(sv, L, lambda);
        const Float weight = 1.0f;  // BoxFilter::sample weight (filter.rs:104)
        r += (double)(weight * rgb.x);
        g += (double)(weight * rgb.y);
        b += (double)(weight * rgb.z);
        w += (double)weight;
    }
    fp->rgb_sum[0] = r; fp->rgb_sum[1] = g; fp->rgb_sum[2] = b; fp->weight_sum = w;
    if (p_local == 0) atomicAdd(&counters->paths, (unsigned long long)n_pix * (unsigned long long)n_samples);
}

}  // namespace

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
    QueueState* d_qs = nullptr;
    DeviceCounters* d_counters = nullptr;
    uint32_t* d_pixels = nullptr;
    size_t pixels_capacity = 0;
    ShmTile* d_tiles = nullptr;
    uint32_t* d_tile_offset = nullptr;
    size_t tiles_capacity = 0;
    std::vector<uint64_t> tile_bitmap;  // host scratch of the disjointness check in shm_render_wave
    int n_cu = 256;
    // tuned traversal (k_trace3)
    int trace3_blocks = 0;
    int spill3_levels = 1;
    int leaf_min = 16;             // closest-hit: lanes with a pending leaf before the triangle phase runs (SHM_LEAF_MIN)
    int leaf_min_any = 8;          // any-hit (SHM_LEAF_MIN_ANY)
    uint32_t* d_spill3 = nullptr;
    float4* d_rw = nullptr;          // RandomWalk: (le, f cos) per depth per path, 2 * (max_depth + 1) * capacity float4
    size_t rw_floats4 = 0;
    uint32_t* d_spill3_any = nullptr;  // the any-hit kernel may run concurrently with the closest-hit one (second stream)
    hipStream_t stream2 = nullptr;
    uint64_t overlap_paths = 96ull << 20;  // batches below this many paths run K3(b) beside K2(b+1) (SHM_OVERLAP_PATHS; 0 = never)
    int refill_min = 16;
    uint32_t pix_group = 1024;      // path-slot order [tile][sample][pixel in tile] (SHM_PIX_GROUP; >= n_pix: sample-major)
    int queue_parts = 8;           // k_trace3 queue partitions, one per XCD with stealing (SHM_QUEUE_PARTS: 1 or 8)
    uint32_t* d_heads3 = nullptr;  // [2 (closest, any)][8 partitions][32 dwords: one 128-B line per head word]
    std::vector<hipEvent_t> events;
};

namespace {

template <typename T>
int dev_upload(ShmScene* s, const std::vector<T>& v, const T** out) {
    size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, bytes));
    s->allocs.push_back(d);
    if (!v.empty()) HIP_TRY(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = reinterpret_cast<const T*>(d);
    return SHM_OK;
}
template <typename T>
int dev_alloc(ShmScene* s, size_t n, T** out) {
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, std::max<size_t>(n, 1) * sizeof(T)));
    s->allocs.push_back(d);
    *out = reinterpret_cast<T*>(d);
    return SHM_OK;
}

// Path workspace sized to the work: up to SHM_BATCH_PATHS (default 256 Mi paths = 71 GB of the 288 GB) so that all 256 spp
// of the 1024^2 frame are ONE batch (6 closest + 5 any launches for the whole frame). Small batches starve the persistent traversal kernels: with ~400 K resident lanes a
// 1 M-ray launch gives each lane ~3 rays and the launch time is set by the longest ray, not by throughput (profiles/r01_*).
static uint64_t max_batch_paths() {
    uint64_t max_cap = 1ull << 28;
    if (const char* e = getenv("SHM_BATCH_PATHS")) {
        long long v = atoll(e);
        if (v >= 4096) max_cap = (uint64_t)v;
    }
    return max_cap;
}

// The batch limit on THIS device right now: SHM_BATCH_PATHS, bounded by 80 % of the memory that is free (plus what the
// current workspace already holds), so that a GPU shared with other allocations degrades to more batches, not to an error.
static uint64_t workspace_cap(const ShmScene* s) {
    const uint64_t BYTES_PER_PATH = 264 + 3 * 4 + (s->flat.has_textures ? 48 : 0);  // path state + three queues (+ auxiliary rays)
    uint64_t cap = max_batch_paths();
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        uint64_t avail = (uint64_t)free_b + (uint64_t)s->capacity * BYTES_PER_PATH;
        uint64_t by_mem = (uint64_t)((double)avail * 0.8) / BYTES_PER_PATH;
        if (by_mem < cap) cap = by_mem;
    }
    return std::max<uint64_t>(cap, 4096);
}

int ensure_workspace(ShmScene* s, uint64_t needed_paths) {
    uint64_t max_cap = workspace_cap(s);
    uint64_t want = std::min<uint64_t>(std::max<uint64_t>(needed_paths, 4096), max_cap);
    want = (want + 4095ull) & ~4095ull;
    if (want > 0xfffff000ull) want = 0xfffff000ull;
    if (s->capacity >= want) return SHM_OK;
    for (void* p : s->ws_allocs) hipFree(p);
    s->ws_allocs.clear();
    s->capacity = 0;
    uint32_t cap = (uint32_t)want;
    auto ws_alloc = [&](size_t bytes, void** out) -> int {
        void* d = nullptr;
        if (hipMalloc(&d, bytes) != hipSuccess) { g_err = "hipMalloc of the path workspace failed"; return SHM_ERR_OUT_OF_MEMORY; }
        s->ws_allocs.push_back(d);
        *out = d;
        return SHM_OK;
    };
    int rc;
#define WS(field, type) if ((rc = ws_alloc((size_t)cap * sizeof(type), (void**)&s->pa.field)) != SHM_OK) return rc
    WS(ray, ShmRay); WS(hit, ShmHit); WS(shadow_ray, ShmRay); WS(shadow_contrib, float4); WS(L, float4); WS(beta, float4);
    WS(lambda, float4); WS(lambda_pdf, float4); WS(ctx0, float4); WS(ctx1, float4); WS(ctx2, float4); WS(pb_eta, float2);
    WS(rng, uint2); WS(pixel, uint32_t); WS(flags, uint32_t);
    s->pa.aux0 = s->pa.aux1 = s->pa.aux2 = nullptr;
    if (s->flat.has_textures) { WS(aux0, float4); WS(aux1, float4); WS(aux2, float4); }
#undef WS
    if ((rc = ws_alloc((size_t)cap * 4, (void**)&s->d_q_active[0])) != SHM_OK) return rc;
    if ((rc = ws_alloc((size_t)cap * 4, (void**)&s->d_q_active[1])) != SHM_OK) return rc;
    if ((rc = ws_alloc((size_t)cap * 4, (void**)&s->d_q_shadow)) != SHM_OK) return rc;
    s->capacity = cap;
    DBG("workspace: %u paths (%.2f GB)", cap, (double)cap * 276.0 / 1e9);
    return SHM_OK;
}

template <bool ANY>
int launch_trace(ShmScene* s, hipStream_t stream, const uint32_t* queue, const uint32_t* n_ptr, uint32_t n_direct, const ShmRay* rays,
                  ShmHit* hits, uint8_t* occluded, float4* L, const float4* contrib) {
    uint32_t* heads = s->d_heads3 + (ANY ? 8 * 32 : 0);
    uint32_t* spill = ANY ? s->d_spill3_any : s->d_spill3;
    hipLaunchKernelGGL(k_reset_heads3, dim3(1), dim3(64), 0, stream, heads);
    const int leaf_min = ANY ? s->leaf_min_any : s->leaf_min;
    if (s->flat.has_spheres)
        hipLaunchKernelGGL((k_trace3<ANY, false>), dim3(s->trace3_blocks), dim3(TRACE_BLOCK), 0, stream, s->dsv, queue, n_ptr, n_direct, heads, rays,
                           hits, occluded, L, contrib, s->d_counters, spill, s->spill3_levels, s->refill_min, leaf_min, s->queue_parts);
    else
        hipLaunchKernelGGL((k_trace3<ANY, true>), dim3(s->trace3_blocks), dim3(TRACE_BLOCK), 0, stream, s->dsv, queue, n_ptr, n_direct, heads, rays,
                           hits, occluded, L, contrib, s->d_counters, spill, s->spill3_levels, s->refill_min, leaf_min, s->queue_parts);
    LAUNCH_TRY(ANY ? "k_trace3<any>" : "k_trace3<closest>");
    return SHM_OK;
}

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

}  // namespace

extern "C" {

const char* shm_last_error(void) { return g_err.c_str(); }
// for the host mirror (host_mirror.cpp), which shares this thread-local message; not exported
__attribute__((visibility("hidden"))) void shm_set_last_error(const char* msg) { g_err = msg ? msg : ""; }

int shm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void shm_scene_destroy(ShmScene* s) {
    if (!s) return;
    hipSetDevice(s->device);
    for (void* p : s->allocs) hipFree(p);
    for (void* p : s->ws_allocs) hipFree(p);
    if (s->d_rw) hipFree(s->d_rw);
    if (s->d_tiles) hipFree(s->d_tiles);
    if (s->d_tile_offset) hipFree(s->d_tile_offset);
    if (s->d_pixels) hipFree(s->d_pixels);
    for (hipEvent_t e : s->events) hipEventDestroy(e);
    if (s->stream2) hipStreamDestroy(s->stream2);
    if (s->stream) hipStreamDestroy(s->stream);
    delete s;
}

int shm_scene_create(const ShmSceneDesc* desc, int device, ShmScene** out) {
    if (!out) { g_err = "out is null"; return SHM_ERR_INVALID_ARGUMENT; }
    *out = nullptr;
    ShmScene* s = new ShmScene();
    int rc = shm_host::flatten_scene(desc, s->flat, g_err);
    if (rc != SHM_OK) { delete s; return rc; }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) {
        g_err = "no HIP device visible (libshimmer_hip has no CPU fallback)";
        delete s;
        return SHM_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n_dev) { g_err = "device ordinal out of range"; delete s; return SHM_ERR_INVALID_ARGUMENT; }
    s->device = device;
    auto fail = [&](int code) { shm_scene_destroy(s); return code; };
    if (hipSetDevice(device) != hipSuccess) { g_err = "hipSetDevice failed"; return fail(SHM_ERR_DEVICE); }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) s->n_cu = prop.multiProcessorCount;
    if (hipStreamCreate(&s->stream) != hipSuccess) { g_err = "hipStreamCreate failed"; return fail(SHM_ERR_DEVICE); }
    if (hipStreamCreate(&s->stream2) != hipSuccess) { g_err = "hipStreamCreate failed"; return fail(SHM_ERR_DEVICE); }
    if (const char* e = getenv("SHM_OVERLAP_PATHS")) { long long v2 = atoll(e); if (v2 >= 0) s->overlap_paths = (uint64_t)v2; }

    const shm_host::FlatScene& f = s->flat;
    SceneView v = f.view();  // scalars + host pointers; pointers replaced below
    if ((rc = dev_upload(s, f.nodes, &v.nodes)) != SHM_OK) return fail(rc);
    if ((rc = dev_upload(s, f.prim_recs, &v.prim_recs)) != SHM_OK) return fail(rc);
    if ((rc = dev_upload(s, f.primitives, &v.primitives)) != SHM_OK) return fail(rc);
    if ((rc = dev_upload(s, f.mesh_flags, &v.mesh_flags)) != SHM_OK) return fail(rc);
    if ((rc = dev_upload(s, f.vi, &v.vi)) != SHM_OK) return fail(rc);
    if ((rc = dev_upload(s, f.vn, &v.vn)) != SHM_OK) return fail(rc);
    if ((rc = dev_upload(s, f.vs, &v.vs)) != SHM_OK) return fail(rc);
    if ((rc = dev_upload(s, f.vuv, &v.vuv)) != SHM_OK) return fail(rc);
    if ((rc = dev_upload(s, f.spheres, &v.spheres)) != SHM_OK) return fail(rc);
    if ((rc = dev_upload(s, f.patches, &v.patches)) != SHM_OK) return fail(rc);
    if ((rc = dev_upload(s, f.patch_vi, &v.patch_vi)) != SHM_OK) return fail(rc);
    if ((rc = dev_upload(s, f.patch_vn, &v.patch_vn)) != SHM_OK) return fail(rc);
    if ((rc = dev_upload(s, f.patch_vuv, &v.patch_vuv)) != SHM_OK) return fail(rc);
    if ((rc = dev_upload(s, f.materials, &v.materials)) != SHM_OK) return fail(rc);
    if ((rc = dev_upload(s, f.lights, &v.lights)) != SHM_OK) return fail(rc);
    if ((rc = dev_upload(s, f.infinite_lights, &v.infinite_lights)) != SHM_OK) return fail(rc);
    if ((rc = dev_upload(s, f.spectrum_data, &v.spectrum_data)) != SHM_OK) return fail(rc);
    if ((rc = dev_upload(s, f.sensor_r, &v.sensor_r_bar)) != SHM_OK) return fail(rc);
    if ((rc = dev_upload(s, f.sensor_g, &v.sensor_g_bar)) != SHM_OK) return fail(rc);
    if ((rc = dev_upload(s, f.sensor_b, &v.sensor_b_bar)) != SHM_OK) return fail(rc);
    if ((rc = dev_upload(s, f.image_textures, &v.image_textures)) != SHM_OK) return fail(rc);
    if ((rc = dev_upload(s, f.image_levels, &v.image_levels)) != SHM_OK) return fail(rc);
    if ((rc = dev_upload(s, f.texel_data, &v.texel_data)) != SHM_OK) return fail(rc);
    if ((rc = dev_upload(s, f.rgb2spec_scale, &v.rgb2spec_scale)) != SHM_OK) return fail(rc);
    if ((rc = dev_upload(s, f.rgb2spec_data, &v.rgb2spec_data)) != SHM_OK) return fail(rc);
    if ((rc = dev_upload(s, f.cs_illuminant, &v.cs_illuminant)) != SHM_OK) return fail(rc);
    if ((rc = dev_upload(s, f.ewa_lut, &v.ewa_lut)) != SHM_OK) return fail(rc);
    if ((rc = dev_upload(s, f.instances, &v.instances)) != SHM_OK) return fail(rc);
    if ((rc = dev_upload(s, f.float_textures, &v.float_textures)) != SHM_OK) return fail(rc);
    if ((rc = dev_upload(s, f.ftex_ranges, &v.ftex_ranges)) != SHM_OK) return fail(rc);
    if ((rc = dev_upload(s, f.ftex_ops, &v.ftex_ops)) != SHM_OK) return fail(rc);
    if ((rc = dev_upload(s, f.spectrum_textures, &v.spectrum_textures)) != SHM_OK) return fail(rc);
    if ((rc = dev_upload(s, f.stex_ranges, &v.stex_ranges)) != SHM_OK) return fail(rc);
    if ((rc = dev_upload(s, f.stex_ops, &v.stex_ops)) != SHM_OK) return fail(rc);
    if ((rc = dev_upload(s, f.image_lights, &v.image_lights)) != SHM_OK) return fail(rc);
    if ((rc = dev_upload(s, f.dist_data, &v.dist_data)) != SHM_OK) return fail(rc);
    s->dsv = v;

    size_t w = (size_t)(f.film.pixel_bounds[2] - f.film.pixel_bounds[0]);
    size_t h = (size_t)(f.film.pixel_bounds[3] - f.film.pixel_bounds[1]);
    s->n_film_pixels = w * h;
    if ((rc = dev_alloc<ShmFilmPixel>(s, s->n_film_pixels, &s->d_film)) != SHM_OK) return fail(rc);
    if (hipMemset(s->d_film, 0, s->n_film_pixels * sizeof(ShmFilmPixel)) != hipSuccess) { g_err = "hipMemset film"; return fail(SHM_ERR_DEVICE); }
    if ((rc = dev_alloc<QueueState>(s, 1, &s->d_qs)) != SHM_OK) return fail(rc);
    if ((rc = dev_alloc<DeviceCounters>(s, 1, &s->d_counters)) != SHM_OK) return fail(rc);
    if ((rc = dev_alloc<uint32_t>(s, 2 * 8 * 32, &s->d_heads3)) != SHM_OK) return fail(rc);
    hipMemset(s->d_qs, 0, sizeof(QueueState));
    hipMemset(s->d_counters, 0, sizeof(DeviceCounters));

    // Tuning knobs (development): defaults are the measured optimum on S3
    if (const char* e = getenv("SHM_PIX_GROUP")) { long long v2 = atoll(e); if (v2 >= 1) s->pix_group = (uint32_t)std::min<long long>(v2, 0x7fffffffll); }
    if (const char* e = getenv("SHM_QUEUE_PARTS")) { int v2 = atoi(e); if (v2 == 1 || v2 == 8) s->queue_parts = v2; }
    if (const char* e = getenv("SHM_REFILL_MIN")) { int v2 = atoi(e); if (v2 >= 1 && v2 <= 64) s->refill_min = v2; }
    {
        int per_cu3 = 6;  // persistent grid: 26 KiB of LDS per 256-thread workgroup -> 6 workgroups (24 waves) per CU
        if (const char* e = getenv("SHM_TRACE3_BLOCKS_PER_CU")) { int v2 = atoi(e); if (v2 >= 1 && v2 <= 6) per_cu3 = v2; }
        if (const char* e = getenv("SHM_LEAF_MIN")) { int v2 = atoi(e); if (v2 >= 1 && v2 <= 64) s->leaf_min = v2; }
        if (const char* e = getenv("SHM_LEAF_MIN_ANY")) { int v2 = atoi(e); if (v2 >= 1 && v2 <= 64) s->leaf_min_any = v2; }
        s->trace3_blocks = s->n_cu * per_cu3;
        s->spill3_levels = std::max(0, (int)f.max_leaf_depth + 1 - K3_LDS_N) + 1;
        if ((rc = dev_alloc<uint32_t>(s, (size_t)s->trace3_blocks * (TRACE_BLOCK / WAVE) * (size_t)s->spill3_levels * WAVE, &s->d_spill3)) != SHM_OK) return fail(rc);
        if ((rc = dev_alloc<uint32_t>(s, (size_t)s->trace3_blocks * (TRACE_BLOCK / WAVE) * (size_t)s->spill3_levels * WAVE, &s->d_spill3_any)) != SHM_OK) return fail(rc);
    }
    DBG("scene: %u nodes, depth %u, trace blocks %d, spill levels %d", (unsigned)f.nodes.size(), f.max_leaf_depth, s->trace3_blocks, s->spill3_levels);
    *out = s;
    return SHM_OK;
}

int shm_film_clear(ShmScene* s) {
    if (!s) return SHM_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipMemsetAsync(s->d_film, 0, s->n_film_pixels * sizeof(ShmFilmPixel), s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return SHM_OK;
}

int shm_film_read(ShmScene* s, ShmFilmPixel* film_out) {
    if (!s || !film_out) return SHM_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipMemcpy(film_out, s->d_film, s->n_film_pixels * sizeof(ShmFilmPixel), hipMemcpyDeviceToHost));
    return SHM_OK;
}

int shm_film_device_ptr(ShmScene* s, void** ptr_out, uint64_t* bytes_out) {
    if (!s || !ptr_out || !bytes_out) return SHM_ERR_INVALID_ARGUMENT;
    *ptr_out = s->d_film;
    *bytes_out = (uint64_t)(s->n_film_pixels * sizeof(ShmFilmPixel));
    return SHM_OK;
}

int shm_render_wave(ShmScene* s, const ShmRenderParams* params, const ShmTile* tiles, uint32_t n_tiles, int32_t sample_begin,
                    int32_t sample_end, ShmStats* stats) {
    if (!s || !params || !tiles || n_tiles == 0 || sample_end <= sample_begin) { g_err = "invalid render arguments"; return SHM_ERR_INVALID_ARGUMENT; }
    if (params->max_depth < 0 || params->max_depth > 254) { g_err = "max_depth out of range"; return SHM_ERR_INVALID_ARGUMENT; }
    if (params->integrator > SHM_INTEGRATOR_RANDOM_WALK) { g_err = "unknown integrator"; return SHM_ERR_UNSUPPORTED; }
    HIP_TRY(hipSetDevice(s->device));
    int rc;
    const int32_t* pb = s->flat.film.pixel_bounds;
    // pixel list for these tiles
    std::vector<uint32_t> tile_offset(n_tiles);
    uint64_t n_pixels = 0;
    for (uint32_t t = 0; t < n_tiles; ++t) {
        const ShmTile& tl = tiles[t];
        if (tl.x0 < pb[0] || tl.y0 < pb[1] || tl.x1 > pb[2] || tl.y1 > pb[3] || tl.x1 <= tl.x0 || tl.y1 <= tl.y0 || tl.x1 > 65535 || tl.y1 > 65535 ||
            tl.x0 < 0 || tl.y0 < 0) {
            g_err = "tile outside pixel bounds";
            return SHM_ERR_INVALID_ARGUMENT;
        }
        tile_offset[t] = (uint32_t)n_pixels;
        n_pixels += (uint64_t)(tl.x1 - tl.x0) * (uint64_t)(tl.y1 - tl.y0);
    }
    if (n_pixels > 0xffffffffull) { g_err = "too many pixels"; return SHM_ERR_INVALID_ARGUMENT; }
    // Tiles must be disjoint: the film update is one unsynchronised read-modify-write per pixel, as in the reference
    // (integrator.rs:277-295 relies on Tile::tile's exclusive ownership). One bit per film pixel, one masked word per tile row.
    {
        const uint32_t fw = (uint32_t)(pb[2] - pb[0]);
        const size_t words_per_row = (fw + 63u) / 64u;
        s->tile_bitmap.assign(words_per_row * (size_t)(pb[3] - pb[1]), 0ull);
        for (uint32_t t = 0; t < n_tiles; ++t) {
            const ShmTile& tl = tiles[t];
            const uint32_t x0 = (uint32_t)(tl.x0 - pb[0]), x1 = (uint32_t)(tl.x1 - pb[0]);
            for (int y = tl.y0; y < tl.y1; ++y) {
                uint64_t* row = s->tile_bitmap.data() + (size_t)(y - pb[1]) * words_per_row;
                for (uint32_t w0 = x0 / 64u; w0 * 64u < x1; ++w0) {
                    const uint32_t lo = std::max(x0, w0 * 64u) - w0 * 64u, hi = std::min(x1, w0 * 64u + 64u) - w0 * 64u;  // bits [lo, hi)
                    const uint64_t mask = (hi - lo == 64u ? ~0ull : ((1ull << (hi - lo)) - 1ull)) << lo;
                    if (row[w0] & mask) { g_err = "tiles overlap (each pixel must belong to at most one tile of a call)"; return SHM_ERR_INVALID_ARGUMENT; }
                    row[w0] |= mask;
                }
            }
        }
    }
    // the tile / pixel lists are regrown on demand; the superseded buffers are released (the stream is idle between calls)
    if (s->tiles_capacity < n_tiles) {
        if (s->d_tiles) hipFree(s->d_tiles);
        if (s->d_tile_offset) hipFree(s->d_tile_offset);
        s->d_tiles = nullptr; s->d_tile_offset = nullptr; s->tiles_capacity = 0;
        if (hipMalloc((void**)&s->d_tiles, (size_t)n_tiles * sizeof(ShmTile)) != hipSuccess ||
            hipMalloc((void**)&s->d_tile_offset, (size_t)n_tiles * sizeof(uint32_t)) != hipSuccess) { g_err = "hipMalloc of the tile list failed"; return SHM_ERR_OUT_OF_MEMORY; }
        s->tiles_capacity = n_tiles;
    }
    if (s->pixels_capacity < n_pixels) {
        if (s->d_pixels) hipFree(s->d_pixels);
        s->d_pixels = nullptr; s->pixels_capacity = 0;
        if (hipMalloc((void**)&s->d_pixels, (size_t)n_pixels * sizeof(uint32_t)) != hipSuccess) { g_err = "hipMalloc of the pixel list failed"; return SHM_ERR_OUT_OF_MEMORY; }
        s->pixels_capacity = n_pixels;
    }
    HIP_TRY(hipMemcpyAsync(s->d_tiles, tiles, n_tiles * sizeof(ShmTile), hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipMemcpyAsync(s->d_tile_offset, tile_offset.data(), n_tiles * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipMemsetAsync(s->d_counters, 0, sizeof(DeviceCounters), s->stream));
    hipLaunchKernelGGL(k_expand_tiles, dim3((n_tiles + 255) / 256), dim3(256), 0, s->stream, s->d_tiles, s->d_tile_offset, n_tiles, s->d_pixels);

    const int n_samples = sample_end - sample_begin;
    const bool random_walk = params->integrator == SHM_INTEGRATOR_RANDOM_WALK;
    // (the random walk keeps 32 B per depth per path beside the path state: its batches are capped at 16 Mi paths)
    if ((rc = ensure_workspace(s, random_walk ? std::min<uint64_t>(n_pixels * (uint64_t)n_samples, 1ull << 24) : n_pixels * (uint64_t)n_samples)) != SHM_OK) return rc;
    uint32_t cap_eff = random_walk ? std::min<uint32_t>(s->capacity, 1u << 24) : s->capacity;  // paths per batch
    if (random_walk) {
        // 32 B per depth per path (up to 8 KB per path at max_depth 254): shrink the batch until the records fit in 80 % of what is free
        const size_t per_path = (size_t)2 * (size_t)(params->max_depth + 1) * sizeof(float4);
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const size_t avail = (size_t)((double)(free_b + s->rw_floats4 * sizeof(float4)) * 0.8);
            while (cap_eff > 4096u && (size_t)cap_eff * per_path > avail) cap_eff = (cap_eff / 2u + 63u) & ~63u;
        }
        size_t need = (size_t)2 * (size_t)(params->max_depth + 1) * (size_t)cap_eff;
        if (s->rw_floats4 < need) {
            if (s->d_rw) hipFree(s->d_rw);
            s->d_rw = nullptr;
            s->rw_floats4 = 0;
            if (hipMalloc((void**)&s->d_rw, need * sizeof(float4)) != hipSuccess) { g_err = "hipMalloc of the random-walk records failed"; return SHM_ERR_OUT_OF_MEMORY; }
            s->rw_floats4 = need;
        }
    }
    uint32_t pix_per_batch = cap_eff / (uint32_t)n_samples;
    if (pix_per_batch == 0) { g_err = "spp-wave larger than the path workspace"; return SHM_ERR_INVALID_ARGUMENT; }
    if (pix_per_batch > 64) pix_per_batch &= ~63u;  // whole 8x8 tiles per wavefront
    EventPool ev{s};
    hipEvent_t e_begin = ev.get(), e_end = ev.get();
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_closest, ev_any, ev_shade;
    bool used_overlap = false;
    HIP_TRY(hipEventRecord(e_begin, s->stream));
    const int shade_blocks = s->n_cu * 4;
    for (uint64_t p0 = 0; p0 < n_pixels; p0 += pix_per_batch) {
        uint32_t n_pix = (uint32_t)std::min<uint64_t>(pix_per_batch, n_pixels - p0);
        uint32_t total = n_pix * (uint32_t)n_samples;
        const uint32_t* pixels = s->d_pixels + p0;
        hipLaunchKernelGGL(k_generate, dim3((total + SHADE_BLOCK - 1) / SHADE_BLOCK), dim3(SHADE_BLOCK), 0, s->stream, s->dsv, s->pa, pixels, n_pix,
                           sample_begin, n_samples, *params, s->d_q_active[0], s->d_qs, s->pix_group);
        LAUNCH_TRY("k_generate");
        int cur = 0;
        // Small batches are tail-dominated (the last rays of a persistent traversal launch take ~0.5 ms whatever its size): there
        // K3 of bounce b runs on a second stream beside K2 of bounce b+1 — they are independent: K3 reads the shadow buffers and
        // adds into L, K2 reads the extension rays and writes hit records — and the next shade launch waits for both. Large
        // batches (the 1-GPU headline frame) keep everything on one stream, so each kernel has the device to itself.
        const bool overlap = s->overlap_paths > 0 && (uint64_t)total < s->overlap_paths && params->max_depth > 0;
        hipStream_t any_stream = overlap ? s->stream2 : s->stream;
        used_overlap = used_overlap || overlap;
        hipEvent_t k3_done = nullptr;
        for (int bounce = 0; bounce <= params->max_depth; ++bounce) {
            const int sh = bounce & 1;
            hipEvent_t a = ev.get(), b = ev.get();
            hipEventRecord(a, s->stream);
            if ((rc = launch_trace<false>(s, s->stream, s->d_q_active[cur], &s->d_qs->n_active[cur], 0, s->pa.ray, s->pa.hit, nullptr, nullptr, nullptr)) != SHM_OK) return rc;
            hipEventRecord(b, s->stream);
            ev_closest.push_back({a, b});
            if (overlap && k3_done) hipStreamWaitEvent(s->stream, k3_done, 0);  // shade(b) touches L and refills the shadow buffers
            {
                hipEvent_t s0 = ev.get(), s1 = ev.get();
                hipEventRecord(s0, s->stream);
                auto launch_shade = [&](auto kernel) {
                    hipLaunchKernelGGL(kernel, dim3(shade_blocks), dim3(SHADE2_BLOCK), 0, s->stream, s->dsv, s->pa, s->d_q_active[cur],
                                       s->d_q_active[cur ^ 1], s->d_q_shadow, s->d_qs, cur, *params, s->d_counters, sh);
                };
                const bool tri_only = !s->flat.has_spheres;
                if (random_walk)
                    hipLaunchKernelGGL(k_shade_randomwalk, dim3(shade_blocks), dim3(SHADE2_BLOCK), 0, s->stream, s->dsv, s->pa, s->d_q_active[cur],
                                       s->d_q_active[cur ^ 1], s->d_qs, cur, *params, s->d_rw, cap_eff);
                else if (params->integrator == SHM_INTEGRATOR_SIMPLE_PATH)
                    hipLaunchKernelGGL(k_shade_simple, dim3(shade_blocks), dim3(SHADE2_BLOCK), 0, s->stream, s->dsv, s->pa, s->d_q_active[cur],
                                       s->d_q_active[cur ^ 1], s->d_q_shadow, s->d_qs, cur, *params, sh);
                else if (s->flat.has_textures || params->force_diffuse) {  // the general instantiations (textures, image lights, force_diffuse)
                    if (s->flat.has_layered) launch_shade(k_shade<true, false, true>); else launch_shade(k_shade<false, false, true>);
                }
                else if (s->flat.has_layered) { if (tri_only) launch_shade(k_shade<true, true>); else launch_shade(k_shade<true, false>); }
                else if (tri_only && s->flat.diffuse_only && !getenv("SHM_NO_DIFFUSE_ONLY")) launch_shade(k_shade<false, true, false, true>);
                else { if (tri_only) launch_shade(k_shade<false, true>); else launch_shade(k_shade<false, false>); }
                LAUNCH_TRY("k_shade");
                hipEventRecord(s1, s->stream);
                ev_shade.push_back({s0, s1});
            }
            if (bounce < params->max_depth && !random_walk) {
                hipEvent_t c = ev.get(), d = ev.get();
                if (overlap) {
                    hipEvent_t shaded = ev.get();
                    hipEventRecord(shaded, s->stream);
                    hipStreamWaitEvent(any_stream, shaded, 0);
                }
                hipEventRecord(c, any_stream);
                if ((rc = launch_trace<true>(s, any_stream, s->d_q_shadow, &s->d_qs->n_shadow[sh], 0, s->pa.shadow_ray, nullptr, nullptr, s->pa.L, s->pa.shadow_contrib)) != SHM_OK) return rc;
                hipEventRecord(d, any_stream);
                ev_any.push_back({c, d});
                k3_done = d;
            }
            if (dbg_on()) {  // queue sizes per bounce (costs a sync: debug only)
                QueueState q;
                hipStreamSynchronize(s->stream);
                hipStreamSynchronize(any_stream);
                hipMemcpy(&q, s->d_qs, sizeof(q), hipMemcpyDeviceToHost);
                DBG("bounce %d: traced %u, next %u, shadow %u", bounce, q.n_active[cur], q.n_active[cur ^ 1], q.n_shadow[sh]);
            }
            hipLaunchKernelGGL(k_next_bounce, dim3(1), dim3(1), 0, s->stream, s->d_qs, cur, sh ^ 1);
            cur ^= 1;
        }
        if (overlap && k3_done) hipStreamWaitEvent(s->stream, k3_done, 0);  // the film reads L
        if (random_walk)
            hipLaunchKernelGGL(k_fold_randomwalk, dim3((total + SHADE_BLOCK - 1) / SHADE_BLOCK), dim3(SHADE_BLOCK), 0, s->stream, s->pa, s->d_rw, cap_eff, total);
        hipLaunchKernelGGL(k_film, dim3((n_pix + SHADE_BLOCK - 1) / SHADE_BLOCK), dim3(SHADE_BLOCK), 0, s->stream, s->dsv, s->pa, pixels, n_pix, n_samples,
                           s->d_film, s->d_counters, s->pix_group);
        LAUNCH_TRY("k_film");
        if (ev.failed) { g_err = "hipEventCreate failed"; return SHM_ERR_DEVICE; }
    }
    HIP_TRY(hipEventRecord(e_end, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipGetLastError());
    if (stats) {
        DeviceCounters c;
        HIP_TRY(hipMemcpy(&c, s->d_counters, sizeof(c), hipMemcpyDeviceToHost));
        stats->paths += c.paths;
        stats->rays_closest += c.rays_closest;
        stats->rays_any += c.rays_any;
        stats->nodes_closest += c.nodes_closest;
        stats->tris_closest += c.tris_closest;
        stats->nodes_any += c.nodes_any;
        stats->tris_any += c.tris_any;
        float ms = 0.0f;
        hipEventElapsedTime(&ms, e_begin, e_end);
        stats->ms_total += ms;
        double mc = 0.0, ma = 0.0;
        for (auto& p : ev_closest) { hipEventElapsedTime(&ms, p.first, p.second); mc += ms; DBG("closest launch %.3f ms", ms); }
        for (auto& p : ev_any) { hipEventElapsedTime(&ms, p.first, p.second); ma += ms; DBG("any launch %.3f ms", ms); }
        double msh = 0.0;
        for (auto& p : ev_shade) { hipEventElapsedTime(&ms, p.first, p.second); msh += ms; DBG("shade launch %.3f ms", ms); }
        stats->ms_trace_closest += mc;
        stats->ms_trace_any += ma;
        float tot = 0.0f;
        hipEventElapsedTime(&tot, e_begin, e_end);
        // everything that is not traversal: shade + generate + film. Without overlap that is the rest of the wall time; with K3
        // running beside K2 the kernels' own durations add up to more than the wall time, so the shade launches are summed instead.
        stats->ms_shade += used_overlap ? msh : (double)tot - mc - ma;
        stats->launches_closest += (uint32_t)ev_closest.size();
        stats->launches_any += (uint32_t)ev_any.size();
    }
    return SHM_OK;
}

int shm_render_device(ShmScene* s, const ShmRenderParams* params, const ShmTile* tiles, uint32_t n_tiles, ShmStats* stats) {
    if (!s || !params) { g_err = "invalid render arguments"; return SHM_ERR_INVALID_ARGUMENT; }
    // ImageTileIntegrator::render's wave schedule (integrator.rs:231-233, 306-308: 1,1,2,4,...,64,64,...). The waves only
    // exist there to show progress / write intermediate images (TODO at :311); a pixel's samples are added to the film in
    // increasing sample_index whatever the grouping, so consecutive waves are fused into launches of at least 64 spp (the
    // reference's own maximum wave size) and as many more as fit the path workspace in one batch (a rank that owns 1/8 of
    // the tiles takes all 256 spp at once), without changing a single film sum. SHM_FUSE_WAVES=0 keeps one launch per wave.
    bool fuse = true;
    if (const char* e = getenv("SHM_FUSE_WAVES")) fuse = atoi(e) != 0;
    int spp = params->samples_per_pixel;
    uint64_t n_pixels = 0;
    for (uint32_t t = 0; tiles && t < n_tiles; ++t)
        n_pixels += (uint64_t)std::max(0, tiles[t].x1 - tiles[t].x0) * (uint64_t)std::max(0, tiles[t].y1 - tiles[t].y0);
    HIP_TRY(hipSetDevice(s->device));
    const int max_fuse = (int)std::min<uint64_t>(std::max<uint64_t>(64, n_pixels ? workspace_cap(s) / n_pixels : 64), 1u << 20);
    int wave_start = 0, wave_end = 1, next_wave_size = 1;
    int pend_begin = 0, pend_end = 0;
    while (wave_start < spp) {
        if (pend_end == pend_begin) pend_begin = wave_start;
        pend_end = wave_end;
        int nws = wave_end;  // advance the reference's schedule
        wave_start = wave_end;
        wave_end = std::min(spp, nws + next_wave_size);
        next_wave_size = std::min(2 * next_wave_size, 64);
        bool flush = !fuse || wave_start >= spp || (wave_end - pend_begin) > max_fuse;
        if (flush) {
            int rc = shm_render_wave(s, params, tiles, n_tiles, pend_begin, pend_end, stats);
            if (rc != SHM_OK) return rc;
            pend_begin = pend_end;
        }
    }
    return SHM_OK;
}

int shm_render(ShmScene* s, const ShmRenderParams* params, const ShmTile* tiles, uint32_t n_tiles, ShmFilmPixel* film, ShmStats* stats) {
    if (!s || !params || !film) { g_err = "invalid render arguments"; return SHM_ERR_INVALID_ARGUMENT; }
    if (stats) memset(stats, 0, sizeof(*stats));
    int rc = shm_film_clear(s);
    if (rc != SHM_OK) return rc;
    rc = shm_render_device(s, params, tiles, n_tiles, stats);
    if (rc != SHM_OK) return rc;
    std::vector<ShmFilmPixel> tmp(s->n_film_pixels);
    rc = shm_film_read(s, tmp.data());
    if (rc != SHM_OK) return rc;
    for (size_t i = 0; i < tmp.size(); ++i) {
        film[i].rgb_sum[0] += tmp[i].rgb_sum[0];
        film[i].rgb_sum[1] += tmp[i].rgb_sum[1];
        film[i].rgb_sum[2] += tmp[i].rgb_sum[2];
        film[i].weight_sum += tmp[i].weight_sum;
    }
    return SHM_OK;
}

static int trace_device_impl(ShmScene* s, bool any, const void* rays_dev, uint32_t n, void* out_dev, int repeat, ShmStats* stats) {
    if (!s || !rays_dev || !out_dev || n == 0 || repeat < 1) { g_err = "invalid trace arguments"; return SHM_ERR_INVALID_ARGUMENT; }
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipMemsetAsync(s->d_counters, 0, sizeof(DeviceCounters), s->stream));
    EventPool ev{s};
    std::vector<std::pair<hipEvent_t, hipEvent_t>> evs;
    for (int r = 0; r < repeat; ++r) {
        hipEvent_t a = ev.get(), b = ev.get();
        hipEventRecord(a, s->stream);
        int rc = any ? launch_trace<true>(s, s->stream, nullptr, nullptr, n, (const ShmRay*)rays_dev, nullptr, (uint8_t*)out_dev, nullptr, nullptr)
                     : launch_trace<false>(s, s->stream, nullptr, nullptr, n, (const ShmRay*)rays_dev, (ShmHit*)out_dev, nullptr, nullptr, nullptr);
        if (rc != SHM_OK) return rc;
        hipEventRecord(b, s->stream);
        evs.push_back({a, b});
    }
    HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipGetLastError());
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        DeviceCounters c;
        HIP_TRY(hipMemcpy(&c, s->d_counters, sizeof(c), hipMemcpyDeviceToHost));
        stats->rays_closest = c.rays_closest; stats->rays_any = c.rays_any;
        stats->nodes_closest = c.nodes_closest; stats->tris_closest = c.tris_closest;
        stats->nodes_any = c.nodes_any; stats->tris_any = c.tris_any;
        double tot = 0.0;
        for (auto& p : evs) { float ms = 0.0f; hipEventElapsedTime(&ms, p.first, p.second); tot += ms; }
        if (any) { stats->ms_trace_any = tot; stats->launches_any = (uint32_t)repeat; }
        else { stats->ms_trace_closest = tot; stats->launches_closest = (uint32_t)repeat; }
        stats->ms_total = tot;
    }
    return SHM_OK;
}

int shm_integrator_render(const char* name, const ShmSceneDesc* scene, int device, int32_t max_depth, int regularize,
                          int sample_lights, int sample_bsdf, int32_t samples_per_pixel, int32_t seed, int disable_pixel_jitter,
                          int disable_wavelength_jitter, ShmFilmPixel* film_out, ShmStats* stats_out, int32_t* n_waves_out) {
    if (!name || !scene || !film_out) { g_err = "invalid integrator arguments"; return SHM_ERR_INVALID_ARGUMENT; }
    try {
        shimmer::PathIntegratorParameters p;
        p.max_depth = max_depth;
        p.regularize = regularize != 0;
        p.sample_lights = sample_lights != 0;
        p.sample_bsdf = sample_bsdf != 0;
        p.samples_per_pixel = samples_per_pixel;
        std::unique_ptr<shimmer::Integrator> integrator = shimmer::create_integrator(name, p, *scene, device);
        shimmer::Options options;
        options.seed = seed;
        options.disable_pixel_jitter = disable_pixel_jitter != 0;
        options.disable_wavelength_jitter = disable_wavelength_jitter != 0;
        integrator->render(options);
        auto* w = static_cast<shimmer::WavefrontPathIntegrator*>(integrator.get());
        std::copy(w->film().begin(), w->film().end(), film_out);
        if (stats_out) *stats_out = w->stats();
        if (n_waves_out) *n_waves_out = w->waves();
        return SHM_OK;
    } catch (const shimmer::IntegratorError& e) {
        g_err = e.what();
        return SHM_ERR_UNSUPPORTED;
    } catch (const std::exception& e) {  // nothing unwinds across the ABI
        g_err = e.what();
        return SHM_ERR_INTERNAL;
    }
}

int shm_trace_closest_device(ShmScene* s, const void* rays_dev, uint32_t n, void* hits_dev, int repeat, ShmStats* stats) {
    return trace_device_impl(s, false, rays_dev, n, hits_dev, repeat, stats);
}
int shm_trace_any_device(ShmScene* s, const void* rays_dev, uint32_t n, void* occluded_dev, int repeat, ShmStats* stats) {
    return trace_device_impl(s, true, rays_dev, n, occluded_dev, repeat, stats);
}

static int trace_host_impl(ShmScene* s, bool any, const ShmRay* rays, uint32_t n, void* out, ShmStats* stats) {
    if (!s || !rays || !out || n == 0) { g_err = "invalid trace arguments"; return SHM_ERR_INVALID_ARGUMENT; }
    HIP_TRY(hipSetDevice(s->device));
    void *d_rays = nullptr, *d_out = nullptr;
    size_t out_bytes = any ? (size_t)n : (size_t)n * sizeof(ShmHit);
    HIP_TRY(hipMalloc(&d_rays, (size_t)n * sizeof(ShmRay)));
    if (hipMalloc(&d_out, out_bytes) != hipSuccess) { hipFree(d_rays); g_err = "hipMalloc"; return SHM_ERR_OUT_OF_MEMORY; }
    int rc = SHM_OK;
    if (hipMemcpy(d_rays, rays, (size_t)n * sizeof(ShmRay), hipMemcpyHostToDevice) != hipSuccess) { g_err = "hipMemcpy rays"; rc = SHM_ERR_DEVICE; }
    if (rc == SHM_OK) rc = trace_device_impl(s, any, d_rays, n, d_out, 1, stats);
    if (rc == SHM_OK && hipMemcpy(out, d_out, out_bytes, hipMemcpyDeviceToHost) != hipSuccess) { g_err = "hipMemcpy out"; rc = SHM_ERR_DEVICE; }
    hipFree(d_rays);
    hipFree(d_out);
    return rc;
}
int shm_trace_closest(ShmScene* s, const ShmRay* rays, uint32_t n, ShmHit* hits_out, ShmStats* stats) { return trace_host_impl(s, false, rays, n, hits_out, stats); }
int shm_trace_any(ShmScene* s, const ShmRay* rays, uint32_t n, uint8_t* occluded_out, ShmStats* stats) { return trace_host_impl(s, true, rays, n, occluded_out, stats); }

}  // extern "C"
