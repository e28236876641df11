// render.hip — the MI355X (gfx950, wave64) wavefront path tracer behind include/shimmer_hip.h.
//
// Replaces the tile-parallel loop of the reference (paths relative to /root/reference/src):
//   integrator.rs:226-322  ImageTileIntegrator::render      -> shm_render / shm_render_device / shm_render_wave (host loop below)
//   integrator.rs:326-396  evaluate_pixel_sample            -> K1 k_generate
//   aggregate.rs:71-139    BvhAggregate::intersect          -> K2 k_trace5<false, GEN> (persistent waves, LDS stack, both children per step)
//   aggregate.rs:141-203   BvhAggregate::intersect_predicate-> K3 k_trace5<true, GEN>
//   integrator.rs:772-892  PathIntegrator::li loop body     -> K4+K5 k_shade<HAS_LAYERED, TRI_ONLY> (one path vertex per launch)
//   integrator.rs:897-963  PathIntegrator::sample_ld        -> inside k_shade (shadow ray deferred to K3)
//   film.rs:548-574        RgbFilm::add_sample              -> K6 k_film (per-pixel ordered f64 sums)
// Leaf arithmetic is the single-source header library csrc/shm/*.h (compiled with -ffp-contract=off).
//
// Execution model: one (pixel, sample) per lane; paths live in SoA arrays in HBM; each bounce is
// trace_closest -> shade -> trace_any over index queues compacted with wave-aggregated atomics; queue
// sizes stay on the device (persistent / grid-stride kernels read them), so a whole render (all fused spp-waves)
// is enqueued on one HIP stream without host round trips. There is no CPU fallback anywhere in this file.
#include "wavefront.h"
#include "host/integrator.hpp"

std::string& shm_err() {
    thread_local std::string e;
    return e;
}
#define g_err shm_err()

namespace {

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

// HAS_TEX: scenes that bind image textures carry the camera ray's auxiliary rays (a compile-time switch: with a run-time pointer the
// auxiliary-ray record lived in scratch memory, 52 B of stores per path, in every scene)
template <bool HAS_TEX, bool LEAN = false>
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
    constexpr bool has_tex = HAS_TEX;
    AuxRays aux = aux_none();
    Ray r = generate_camera_ray(sv, px, py, rng, params.disable_wavelength_jitter != 0, params.disable_pixel_jitter != 0,
                                lambda, weight, HAS_TEX ? &aux : nullptr, params.samples_per_pixel);
    if (HAS_TEX) st_aux(pa, slot, aux);
    ShmRay ray;
    ray.o[0] = r.o.x; ray.o[1] = r.o.y; ray.o[2] = r.o.z;
    ray.d[0] = r.d.x; ray.d[1] = r.d.y; ray.d[2] = r.d.z;
    ray.t_max = infinity();
    ray.pad = 0.0f;
    pa.ray[slot] = ray;
    pa.L[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    // LEAN (the fused kernel shades bounce 0 and knows it is bounce 0): the constants — beta = 1, p_b = eta_scale = 1, flags = 0, the identity queue — are not
    // written here nor read there (ShadeArgs::first_bounce): 32 of 124 bytes per path
    const float4 lambda4 = make_float4(lambda.lambda[0], lambda.lambda[1], lambda.lambda[2], lambda.lambda[3]);
    pa.lambda[slot] = lambda4;  // (the film's copy: k_film reads wavelengths and pdfs of every sample, and a 64-byte record for 16 of its bytes would triple that)
    pa.lambda_pdf[slot] = make_float4(lambda.pdf[0], lambda.pdf[1], lambda.pdf[2], lambda.pdf[3]);
    // (the CtxRec, the previous vertex's LightSampleContext, is first read at depth >= 1, after k_shade has written them)
    if (LEAN) {
        // no record: the fused kernel's bounce 0 reads these three arrays and writes the path's first record whole (one full 64-byte store instead of this kernel's
        // partial sectors: k_generate 10.5 -> 6 ms per headline frame)
        pa.rng0[slot] = make_uint2((uint32_t)rng.state, (uint32_t)(rng.state >> 32));
        pa.pixel0[slot] = pix;
    } else {
        PathRec r;
        r.lambda = lambda4;
        r.rng = make_uint2((uint32_t)rng.state, (uint32_t)(rng.state >> 32));
        r.pixel = pix;
        r.flags = has_tex ? (1u << 10) : 0u;  // camera rays always carry auxiliary rays (camera.rs:1070-1078)
        r.beta = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
        r.pb_eta = make_float2(1.0f, 1.0f);
        r.pad[0] = 0u; r.pad[1] = 0u;
        pa.rec[slot] = r;
    }
    if (!LEAN) q_active[slot] = slot;  // first bounce: identity queue
    if (slot == 0) {
        qs->n_active[0] = total;
        qs->n_active[1] = 0;
        qs->n_shadow[0] = 0;
        qs->n_shadow[1] = 0;
        qs->n_scatter[0] = qs->n_scatter[1] = qs->n_scatter[2] = qs->n_scatter[3] = 0;
        qs->n_emit = 0;
        qs->n_lean = 0;
        qs->n_split = 0;
    }
}

// Between bounces: recycle the counters (1 thread).
__global__ void k_next_bounce(QueueState* qs, int cur, int next_shadow_parity) {
    qs->n_active[cur] = 0;
    qs->n_scatter[0] = qs->n_scatter[1] = qs->n_scatter[2] = qs->n_scatter[3] = 0;
    qs->n_lean = 0;
    qs->n_split = 0;
    qs->n_emit = 0;
    qs->n_shadow[next_shadow_parity] = 0;  // the one the NEXT shade launch fills; this bounce's count stays for its K3
}
// ---------------------------------------------------------------------------------------------
// Scenes WITH textures (round 5): the split pass in front of the textured vertex kernel. Until then one textured material put every vertex of the scene through the textured
// class's kernels — ray differentials, the 48-byte differential arrays, 256 VGPRs at two waves per SIMD —: the headline scene with a textured material OUT OF SIGHT shaded in
// 207 ms per frame against 86. A vertex on a DiffuseMaterial that binds no texture (ShmMaterial::pad[0], flatten_scene) needs none of it, and a diffuse bounce ends the
// ray differentials (interaction.rs:430-514: only specular bounces carry them on): nothing a later texture look-up reads depends on which kernel shaded it. This pass — a
// few registers, full occupancy — sends such hits to q_lean (the lean fused kernel takes their whole vertex, as in the lean diversion of k_vertex.inl) and everything
// else, escaped rays included, to q_split, which the textured kernels work through. Paths are independent and every later queue is order-agnostic: films and counters
// do not change.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(SHADE2_BLOCK) k_split_plain(SceneView sv, PathArrays pa, const uint32_t* __restrict__ q_cur, uint32_t* __restrict__ q_lean,
                                                             uint32_t* __restrict__ q_split, QueueState* qs, int cur) {
    const uint32_t n = qs->n_active[cur];
    __shared__ uint32_t s_q[2][SHADE_CHUNK];
    __shared__ uint32_t s_cnt[2], s_base[2];
    for (uint32_t chunk0 = blockIdx.x * SHADE_CHUNK; chunk0 < n; chunk0 += gridDim.x * SHADE_CHUNK) {
        if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
        __syncthreads();
        for (uint32_t k = 0; k < SHADE_CHUNK / SHADE2_BLOCK; ++k) {
            const uint32_t i = chunk0 + k * SHADE2_BLOCK + threadIdx.x;
            bool plain = false, rest = false;
            uint32_t path = 0;
            if (i < n) {
                path = q_cur[i];
                const int prim = pa.hit16 ? hit_prim_of(__float_as_int(reinterpret_cast<const float*>(reinterpret_cast<const float4*>(pa.hit) + path)[0])) : __float_as_int(reinterpret_cast<const float*>(pa.hit + path)[0]);
                plain = prim >= 0 && (sv.materials[sv.prim_recs[prim].material].pad[0] & 1u) != 0u;
                rest = !plain;
            }
            const uint32_t a = queue_push_slot(&s_cnt[0], plain);
            if (plain) s_q[0][a] = path;
            const uint32_t b = queue_push_slot(&s_cnt[1], rest);
            if (rest) s_q[1][b] = path;
        }
        __syncthreads();
        if (threadIdx.x == 0) s_base[0] = s_cnt[0] ? atomicAdd(&qs->n_lean, s_cnt[0]) : 0u;
        if (threadIdx.x == 1) s_base[1] = s_cnt[1] ? atomicAdd(&qs->n_split, s_cnt[1]) : 0u;
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < s_cnt[0]; j += SHADE2_BLOCK) q_lean[s_base[0] + j] = s_q[0][j];
        for (uint32_t j = threadIdx.x; j < s_cnt[1]; j += SHADE2_BLOCK) q_split[s_base[1] + j] = s_q[1][j];
        __syncthreads();
    }
}
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
        if (sv.quirks_off && !spec_is_finite(L)) L = spec_const(0.0f);  // integrator.rs:377-382's TODOs, done only with the quirks switched off
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

namespace {

template <typename T>
int dev_upload(ShmScene* s, const std::vector<T>& v, const T** out) {
    size_t bytes = (std::max<size_t>(v.size(), 1) * sizeof(T) + 15u) & ~(size_t)15u;  // (whole 16-byte groups: the LDS staging of the small tables copies uint4s)
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

// Path workspace sized to the work: up to SHM_BATCH_PATHS (default 512 Mi paths: 148 GB of the 288 GB for a lean scene, bounded by 80 % of
// what is free) so that all 256 spp of the 1024^2 frame (268 M paths, 71 GB) and all 256 spp of the 1000x1400 crown-proxy frame (358 M paths:
// its 23 late bounces of < 1 M rays cost ~2.3 ms each whatever the batch holds, once instead of twice) are ONE batch. Small batches starve the persistent traversal kernels: with ~400 K resident lanes a
// 1 M-ray launch gives each lane ~3 rays and the launch time is set by the longest ray, not by throughput (profiles/r01_*).
static uint64_t max_batch_paths() {
    uint64_t max_cap = 1ull << 29;
    if (const char* e = getenv("SHM_BATCH_PATHS")) {
        long long v = atoll(e);
        if (v >= 4096) max_cap = std::min<uint64_t>((uint64_t)v, (1ull << 30) - 4096ull);  // (below 2^30: the layered scatter kernel's jobs carry two flag bits above the path index)
    }
    return max_cap;
}

// The batch limit on THIS device right now: SHM_BATCH_PATHS, bounded by 80 % of the memory that is free (plus what the
// current workspace already holds), so that a GPU shared with other allocations degrades to more batches, not to an error.
// staged shading (k_vertex -> k_scatter<class>): everything but all-diffuse triangle scenes without textures, which keep the fused kernel
// (one BxDF class: nothing to sort, and the parameter block would be pure traffic); options.force_diffuse always takes the staged path.
// (Round-2 A/B against the fused general kernels of round 1, same box: coated S3 1 208 -> 1 724 Mray/s, textured Cornell 768 -> 906,
// crown-proxy C4 1 758 -> 1 735, Cornell with patches 2 840 -> 2 634: the staged pipeline replaced them everywhere.)
// (round 5: whatever the shapes — a scene with spheres / patches / instances runs the kernel's general-geometry instantiation, k_shade_lean_gen.hip)
// (round 5: ... and an ImageInfinitelight alone does not make a scene "textured": the light's look-up, sample and pdf are compiled into the lean kernels' ENV_LIGHT
// instantiations (k_shade_lean_env.hip); ray differentials and auxiliary rays only feed texture filtering, and nothing filters a texture there)
// `env_plain`: the scene's only image is an environment map. No path-integrator render of it reaches a HAS_TEX kernel unless options.force_diffuse asks for that code — and even
// there the differentials are dead values (no material binds a texture), so the auxiliary-ray arrays are never allocated for it. Every class has its instantiation: the lean
// kernel's and the sorted fused kernel's ENV_LIGHT ones (all-diffuse; glass, metal), the K_ENV_LIGHT units of the staged kernels (coated materials)
static bool env_plain_scene(const ShmScene* s) { return s->flat.has_image_light && !s->flat.has_material_textures; }
static bool env_lean_scene(const ShmScene* s) { return env_plain_scene(s) && s->flat.diffuse_only; }
// (the shapes and classes whose every bounce the material-sorted fused all-materials kernel takes: k_shade_tail*.hip, k_shade_fused_*.hip)
static bool fused_all_from_0(const ShmScene* s) { return !s->flat.has_class[CLASS_LAYERED] && s->tail_fused_bounce == 0; }
static bool scene_is_lean(const ShmScene* s) { return s->flat.diffuse_only && (!s->flat.has_textures || env_lean_scene(s)); }
static bool tex_ws(const ShmScene* s) { return s->flat.has_textures && !env_plain_scene(s); }  // the auxiliary-ray arrays (and k_generate<true>)
// (round 5) scenes whose every bounce shades with ONE fused kernel that knows bounce 0's constants (ShadeArgs::first_bounce): the lean class, and — without textures or coated
// materials — every class the material-sorted fused kernel takes from bounce 0 on (k_shade_tail*.hip, k_shade_fused_gen.hip)
static bool first_bounce_candidate(const ShmScene* s) {
    return scene_is_lean(s) || ((!s->flat.has_textures || env_plain_scene(s)) && fused_all_from_0(s));
}
static bool use_staged(const ShmScene* s, const ShmRenderParams* params) {
    if (params->integrator != SHM_INTEGRATOR_PATH) return false;
    // (the lean class through the staged pipeline, measured: shade + generate + film 130 -> 165 ms per headline frame)
    return !scene_is_lean(s) || params->force_diffuse != 0;
}
static uint64_t staging_bytes_per_path(const ShmScene* s) {
    const shm_host::FlatScene& f = s->flat;
    uint64_t b = 128;  // the parameter block, one BxRec per path
    if (f.has_textures) b += 48;                                               // dd0..2
    for (int c = 0; c < N_BXDF_CLASSES; ++c) if (f.has_class[c]) b += 4;       // class queues
    if (s->lean_divert) b += 4;                                                // the lean diversion's queue
    if (s->split_pass) b += 4;                                                  // the split pass's queue
    return b;
}
static bool uses_fused_kernel(const ShmScene* s) { return scene_is_lean(s) || s->lean_divert; }  // k_shade<lean>: deferred emitter hits (PathArrays::e_*)
static uint64_t workspace_cap(const ShmScene* s, bool need_staged) {
    // path state + three queues (+ auxiliary rays) (+ the staging arrays whenever the upcoming render is staged: every scene class but the
    // lean one, and the lean one too under options.force_diffuse — the budget must count them BEFORE the first staged allocation)
    // (ray 32, hit 32, shadow_ray 32, shadow_contrib 16, L 16, the PathRec 64, lambda 16, lambda_pdf 16, the CtxRec 64 = 288)
    const uint64_t BYTES_PER_PATH = 288 + (first_bounce_candidate(s) ? 12 : 0) + 3 * 4 + (tex_ws(s) ? 48 : 0) + (uses_fused_kernel(s) ? 88 : 0) + ((need_staged || s->ws_staged || !scene_is_lean(s)) ? staging_bytes_per_path(s) : 0);
    uint64_t cap = max_batch_paths();
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        uint64_t avail = (uint64_t)free_b + (uint64_t)s->capacity * BYTES_PER_PATH;
        uint64_t by_mem = (uint64_t)((double)avail * 0.8) / BYTES_PER_PATH;
        if (by_mem < cap) cap = by_mem;
    }
    return std::max<uint64_t>(cap, 4096);
}

}  // namespace
// Which of the small scene tables a kernel with `budget` bytes of LDS to spare stages there (wavefront.h, stage_scene_tables): greedily, in the enum's order.
LdsTables wf_lds_tables(const ShmScene* s, uint32_t budget) {
    LdsTables t = {};
    const shm_host::FlatScene& f = s->flat;
    auto pad16 = [](size_t b) { return (size_t)((b + 15u) & ~(size_t)15u); };  // (dev_upload allocates whole 16-byte groups)
    const size_t want[N_LDS_TABLES] = {
        pad16(f.mesh_flags.size() * sizeof(uint32_t)), pad16(f.lights.size() * sizeof(ShmLight)), pad16(f.light_prim_recs.size() * sizeof(shm::PrimRec)), pad16(f.materials.size() * sizeof(ShmMaterial)), pad16(f.spectrum_data.size() * sizeof(float)),
        pad16(f.rgb2spec_scale.size() * sizeof(float)),
        pad16(f.image_textures.size() * sizeof(ShmImageTexture)), pad16(f.image_levels.size() * sizeof(ShmImageLevel)), pad16(f.float_textures.size() * sizeof(ShmFloatTexture)),
        pad16(f.ftex_ranges.size() * sizeof(shm::FloatTexRange)), pad16(f.ftex_ops.size() * sizeof(shm::FloatTexOp)), pad16(f.spectrum_textures.size() * sizeof(ShmSpectrumTexture)),
        pad16(f.stex_ranges.size() * sizeof(shm::FloatTexRange)), pad16(f.stex_ops.size() * sizeof(shm::FloatTexOp)), pad16(f.ewa_lut.size() * sizeof(float))};
    size_t left = budget;
    for (int k = 0; k < N_LDS_TABLES; ++k)
        if (want[k] && want[k] <= left) { t.bytes[k] = (uint32_t)want[k]; left -= want[k]; }
    return t;
}
namespace {
int ensure_workspace(ShmScene* s, uint64_t needed_paths, bool need_staged) {
    uint64_t max_cap = workspace_cap(s, need_staged);
    uint64_t want = std::min<uint64_t>(std::max<uint64_t>(needed_paths, 4096), max_cap);
    // a large request takes the whole budget at once: freeing and re-allocating ~140 GB because the next call needs a few percent more
    // paths costs seconds (measured: 3.7 s per regrow at 500 M paths)
    need_staged = need_staged || s->ws_staged || !scene_is_lean(s);
    // (what decides is whether THIS render fits: the budget below moves by a few paths from call to call — the free-memory reading, the per-path estimate — and a
    //  workspace that already holds the render must not be freed and re-allocated for it: 7 s per frame at 150 GB)
    if (s->ws_staged || !need_staged) {
        if (s->capacity >= want) return SHM_OK;
        // a render that is batched anyway (it needs more than the budget): a workspace within 10 % of the budget is the budget
        if (needed_paths > max_cap && s->capacity * 10ull >= max_cap * 9ull) return SHM_OK;
    }
    if (want > max_cap / 8) want = max_cap;
    want = (want + 4095ull) & ~4095ull;
    if (want > 0xfffff000ull) want = 0xfffff000ull;
    want = std::max<uint64_t>(want, s->capacity);
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
    WS(ray, ShmRay); WS(hit, ShmHit); WS(shadow_ray, ShmRay); WS(shadow_contrib, float4); WS(L, float4); WS(rec, PathRec);
    WS(lambda, float4); WS(lambda_pdf, float4); WS(ctx, CtxRec);
    s->pa.rng0 = nullptr; s->pa.pixel0 = nullptr;
    if (first_bounce_candidate(s)) { WS(rng0, uint2); WS(pixel0, uint32_t); }
    s->pa.e_ray = s->pa.e_beta = s->pa.e_ctx0 = s->pa.e_ctx1 = s->pa.e_ctx2 = nullptr;
    s->pa.e_flags = nullptr;
    s->d_q_emit = nullptr;
    if (uses_fused_kernel(s)) {
        WS(e_ray, float4); WS(e_beta, float4); WS(e_ctx0, float4); WS(e_ctx1, float4); WS(e_ctx2, float4); WS(e_flags, uint32_t);
        if ((rc = ws_alloc((size_t)cap * 4, (void**)&s->d_q_emit)) != SHM_OK) return rc;
    }
    s->pa.aux0 = s->pa.aux1 = s->pa.aux2 = nullptr;
    if (tex_ws(s)) { WS(aux0, float4); WS(aux1, float4); WS(aux2, float4); }
    s->pa.bx = nullptr;
    s->pa.has_layered = s->flat.has_class[CLASS_LAYERED] ? 1u : 0u;
    s->pa.dd0 = s->pa.dd1 = s->pa.dd2 = nullptr;
    for (int c = 0; c < N_BXDF_CLASSES; ++c) s->d_q_scatter[c] = nullptr;
    s->d_q_lean = nullptr;
    s->d_q_split = nullptr;
    s->ws_staged = false;
    if (need_staged) {
        const shm_host::FlatScene& f = s->flat;
        WS(bx, BxRec);
        if (f.has_textures) { WS(dd0, float4); WS(dd1, float4); WS(dd2, float4); }
        for (int c = 0; c < N_BXDF_CLASSES; ++c)
            if (f.has_class[c] && (rc = ws_alloc((size_t)cap * 4, (void**)&s->d_q_scatter[c])) != SHM_OK) return rc;
        if (s->lean_divert && (rc = ws_alloc((size_t)cap * 4, (void**)&s->d_q_lean)) != SHM_OK) return rc;
        if (s->split_pass && (rc = ws_alloc((size_t)cap * 4, (void**)&s->d_q_split)) != SHM_OK) return rc;
        s->ws_staged = true;
    }
#undef WS
    if ((rc = ws_alloc((size_t)cap * 4, (void**)&s->d_q_active[0])) != SHM_OK) return rc;
    if ((rc = ws_alloc((size_t)cap * 4, (void**)&s->d_q_active[1])) != SHM_OK) return rc;
    if ((rc = ws_alloc((size_t)cap * 4, (void**)&s->d_q_shadow)) != SHM_OK) return rc;
    s->capacity = cap;
    DBG("workspace: %u paths", cap);
    return SHM_OK;
}
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
    wf_trace_census();  // (prints in -DK5_CENSUS development builds only)
    wf_layered_census();  // (-DLJ_CENSUS)
    wf_dist_release(s);
    for (void* p : s->allocs) hipFree(p);
    for (void* p : s->ws_allocs) hipFree(p);
    if (s->d_rw) hipFree(s->d_rw);
    if (s->d_tiles) hipFree(s->d_tiles);
    if (s->d_tile_offset) hipFree(s->d_tile_offset);
    if (s->d_pixels) hipFree(s->d_pixels);
    for (hipEvent_t e : s->events) hipEventDestroy(e);
    if (s->stream2) hipStreamDestroy(s->stream2);
    for (hipStream_t st : s->stream_cls) if (st) hipStreamDestroy(st);
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
    for (hipStream_t& st : s->stream_cls) if (hipStreamCreate(&st) != hipSuccess) { g_err = "hipStreamCreate failed"; return fail(SHM_ERR_DEVICE); }
    if (const char* e = getenv("SHM_OVERLAP_PATHS")) { long long v2 = atoll(e); if (v2 >= 0) s->overlap_paths = (uint64_t)v2; }

    const shm_host::FlatScene& f = s->flat;
    SceneView v = f.view();  // scalars + host pointers; pointers replaced below
    // The DEVICE copy of the tree is laid out by sibling pairs (the ABI's array and the oracle's stay in the reference's depth-first order,
    // aggregate.rs:425-467): the two children of a node share one 64-byte block, the block of a node's first child's children follows. Depth first, a
    // node's second child lies behind its sibling's whole subtree, and the fetch that a pop starts — the head of a dependent chain — misses; here it
    // shares the block its sibling brought in. Same nodes, same visit order, same counters: an interior node's `offset` is its first child's index, the
    // second child is offset + 1 (k_trace.hip). Headline frame: K2 136.0 -> 133.5 ms, K3 81.0 -> 77.8 ms (of which the larger-child-next order: 0.5 %).
    std::vector<ShmBvhNode> pair_nodes;
    std::vector<uint32_t> big_leaf_n;
    std::vector<ShmInstance> pair_instances = f.instances;
    for (ShmInstance& in : pair_instances) in.pad[0] = 0xffffffffu;  // (the slot of the instance's leaf, filled in below; 0xffffffff: not seen yet)
    {
        const std::vector<ShmBvhNode>& dn = f.nodes;
        std::vector<uint32_t> new_index(dn.size(), 0xffffffffu);
        std::vector<uint32_t> roots{0u};
        for (const ShmInstance& in : f.instances) roots.push_back(in.root_node);
        std::sort(roots.begin(), roots.end());
        roots.erase(std::unique(roots.begin(), roots.end()), roots.end());
        uint32_t next = 0;
        std::vector<uint32_t> stack;
        for (uint32_t r : roots) {
            if (r >= dn.size()) { g_err = "instance root node out of range"; return fail(SHM_ERR_INVALID_ARGUMENT); }
            // (a root inside another root's tree — an instanced SUB-tree — would get two device indices: named, not mis-traversed)
            if (new_index[r] != 0xffffffffu) { g_err = "an instance's root node lies inside another tree of the node array (instanced sub-trees are not supported: give the object its own tree)"; return fail(SHM_ERR_INVALID_ARGUMENT); }
            new_index[r] = next;  // (a root sits alone in its block: the odd slot stays a zeroed, never-visited record)
            next += 2;
            stack.assign(1, r);
            while (!stack.empty()) {
                const uint32_t o = stack.back();
                stack.pop_back();
                if (dn[o].n_prims != 0) continue;
                const uint32_t c0 = o + 1u, c1 = dn[o].offset;
                if (c0 >= dn.size() || c1 >= dn.size() || new_index[c0] != 0xffffffffu || new_index[c1] != 0xffffffffu) {
                    g_err = "BVH node array is not a depth-first tree"; return fail(SHM_ERR_INVALID_ARGUMENT);
                }
                new_index[c0] = next;
                new_index[c1] = next + 1u;
                next += 2;
                // the child whose block of children comes next (and, half of the time, in the same 128-byte line): the one a ray is more likely to enter
                auto area = [&](const ShmBvhNode& n) {
                    const float dx = n.bmax[0] - n.bmin[0], dy = n.bmax[1] - n.bmin[1], dz = n.bmax[2] - n.bmin[2];
                    return dx * dy + dy * dz + dz * dx;
                };
                const bool first_next = !(area(dn[c1]) > area(dn[c0]));  // (the larger child's; "the first child's block next" measured 1 % slower in round 4)
                stack.push_back(first_next ? c1 : c0);
                stack.push_back(first_next ? c0 : c1);
            }
        }
        ShmBvhNode zero;
        memset(&zero, 0, sizeof(zero));
        pair_nodes.assign(next, zero);
        if (next > LINK_INDEX_MASK || f.prim_recs.size() > LINK_INDEX_MASK) { g_err = "more than 2^27 BVH nodes or primitives (the device link word holds 27-bit indices)"; return fail(SHM_ERR_UNSUPPORTED); }
        for (size_t o = 0; o < dn.size(); ++o) {
            if (new_index[o] == 0xffffffffu) continue;  // (not reachable from any root)
            ShmBvhNode n = dn[o];
            // the link word (wavefront.h): where a traversal goes on from this node
            if (n.n_prims == 0) n.offset = ((uint32_t)n.axis << LINK_AXIS_SHIFT) | new_index[o + 1];
            else {
                if (n.n_prims >= LINK_COUNT_MAX) {
                    if (big_leaf_n.empty()) big_leaf_n.assign(f.prim_recs.size(), 0u);
                    for (uint32_t j = 0; j < n.n_prims; ++j) big_leaf_n[n.offset + j] = n.n_prims - j;  // (the primitives left from each slot on: k_trace5 takes one per phase)
                }
                // a leaf of ONE primitive that is no triangle is marked as such in the link word itself — a lane that reaches it parks for the wave's next round of
                // non-triangle work straight from the node step, without the leaf phase's fetch of a record it cannot test (k_trace5<., GEN>):
                //   an instance (always alone in its leaf, flatten.h): count 0, the INDEX of the instance in place of the slot (its slot rides in the device copy's pad[0]);
                //   a sphere / a bilinear patch: count 1 and the slot, as a lane would have parked on it
                const uint32_t kind1 = n.n_prims == 1 ? f.prim_recs[n.offset].kind_index : 0u;
                if (kind1 & shm::PRIM_INSTANCE_BIT) {
                    const uint32_t idx = kind1 & shm::PRIM_INDEX_MASK;
                    // (one leaf per ShmInstance: the traversal finds the instance's leaf slot — what a hit inside it is named by — in this record. Two instance primitives
                    //  that shared one ShmInstance would overwrite each other's slot: refused, the host gives each primitive its own record)
                    if (pair_instances[idx].pad[0] != 0xffffffffu) { g_err = "two instance primitives share one ShmInstance record (give each TransformedPrimitive its own)"; return fail(SHM_ERR_INVALID_ARGUMENT); }
                    pair_instances[idx].pad[0] = n.offset;
                    n.offset = LINK_LEAF | LINK_OTHER | idx;
                } else if (kind1 & (shm::PRIM_SPHERE_BIT | shm::PRIM_PATCH_BIT)) {
                    n.offset = LINK_LEAF | LINK_OTHER | (1u << LINK_COUNT_SHIFT) | n.offset;
                } else {
                    n.offset = LINK_LEAF | (std::min<uint32_t>(n.n_prims, LINK_COUNT_MAX) << LINK_COUNT_SHIFT) | n.offset;
                }
            }
            pair_nodes[new_index[o]] = n;
        }
        for (ShmInstance& in : pair_instances) in.root_node = new_index[in.root_node];
    }
    std::vector<ShmBvhNode> inst_roots;
    for (const ShmInstance& in : pair_instances) inst_roots.push_back(pair_nodes[in.root_node]);
    if ((rc = dev_upload(s, pair_nodes, &v.nodes)) != SHM_OK) return fail(rc);
    if (!big_leaf_n.empty()) { const uint32_t* d = nullptr; if ((rc = dev_upload(s, big_leaf_n, &d)) != SHM_OK) return fail(rc); s->d_big_leaf_n = const_cast<uint32_t*>(d); }
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
    v.light_prim_recs = nullptr;
    if (!f.light_prim_recs.empty() && (rc = dev_upload(s, f.light_prim_recs, &v.light_prim_recs)) != SHM_OK) return fail(rc);
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
    if ((rc = dev_upload(s, pair_instances, &v.instances)) != SHM_OK) return fail(rc);
    if ((rc = dev_upload(s, inst_roots, &v.inst_roots)) != SHM_OK) return fail(rc);
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
    // the lean diversion (k_vertex.inl): triangle-only scenes without textures that hold plain diffuse materials BESIDE other classes
    // scenes with material textures: the split pass (k_split_plain) where a quarter of the primitives or more carry a plain DiffuseMaterial (the textured Cornell box, whose
    // only plain material is its emitter's, would pay a pass per bounce for a handful of hits; SHM_SPLIT_PASS=0 / 1 overrides)
    // (in scenes WITHOUT textures k_vertex diverts such hits itself, k_vertex.inl — its triangle instantiation runs three waves per SIMD; the pass as a kernel of its own in front of
    //  it, forced with SHM_SPLIT_PASS=1, changes nothing there: coated S3 3 258-3 284 Mray/s either way)
    // (round 6: ... or a quarter of the SURFACE AREA — hits fall by area, not by count: a textured 4.3 M-triangle object in a plain room of 14 triangles sent every wall and
    //  floor hit through the textured class: 2 796 -> 3 127 Mray/s with the pass, profiles/r06_textured_object.txt)
    s->split_pass = (s->flat.has_material_textures && !scene_is_lean(s) &&
                     (s->flat.n_plain_diffuse_prims * 4ull >= (uint64_t)s->flat.prim_recs.size() ||
                      (s->flat.n_plain_diffuse_prims > 0 && s->flat.area_plain_diffuse * 4.0 >= s->flat.area_total))) ? 1 : 0;
    if (const char* e = getenv("SHM_SPLIT_PASS")) s->split_pass = (atoi(e) != 0 && !scene_is_lean(s) && s->flat.n_plain_diffuse_prims > 0) ? 1 : 0;
    s->lean_divert = ((!s->flat.has_textures || env_plain_scene(s)) && s->flat.has_class[CLASS_DIFFUSE] && !scene_is_lean(s)) || s->split_pass;
    // a shallow tree means short rays, and short rays want fewer, fuller waves (C2's 63-node box: 15.5 -> 15.1 ms per frame at 8 rays per lane); a deep
    // tree means long dependent chains per ray, which want every wave the device has (C4: 8 costs 2 %) — profiles/r03_trace_rays_per_lane_sweep.txt
    if (f.nodes.size() < 4096) s->trace_rays_per_lane = 8;
    if (const char* e = getenv("SHM_TRACE_RAYS_PER_LANE")) { int v2 = atoi(e); if (v2 >= 0 && v2 <= 4096) s->trace_rays_per_lane = v2; }
    s->lds_tables = wf_lds_tables(s, LDS_TABLE_BUDGET);
    s->lds_tables_small = wf_lds_tables(s, LDS_TABLE_BUDGET_SMALL);
    // the traversal kernels' instantiation for scenes with spheres / patches: five waves per SIMD where those shapes are a twelfth of the primitive records or more (k_trace.hip,
    // K5_GEN_HEAVY_WAVES; S3 with 100 / 50 / 25 / 10 % of the object's cells as patches — 100 / 33 / 14 / 5 % of the records —, five against seven waves: +73 / +38 / +13 / 0 %,
    // with ONE patch in 4.3 M triangles -7 %). At five waves a refill is cheaper to make early (24 idle lanes: S3 as patches 3 506 -> 3 685 Mray/s; profiles/r06_patch_heavy_scenes.txt)
    s->gen_heavy = f.has_spheres && f.n_quadric_patch_prims * 12ull >= (uint64_t)f.prim_recs.size();
    if (const char* e = getenv("SHM_GEN_HEAVY")) s->gen_heavy = f.has_spheres && atoi(e) != 0;
    if (s->gen_heavy) s->refill_min = s->refill_min_any = 24;
    // parked rounds of a scene whose only non-triangles are instances are ray set-ups in another space (~450 instructions): worth waiting for 32 lanes (S3 instanced
    // 4 263 -> 4 354, its object as 4 x 4 x 4 instances 2 703 -> 2 839 Mray/s; with spheres / patches 16 stays ahead: profiles/r06_instance_grid.txt)
    if (f.has_instances && f.n_quadric_patch_prims == 0) s->other_min = s->other_min_any = 32;
    if (const char* e = getenv("SHM_REFILL_MIN")) { int v2 = atoi(e); if (v2 >= 1 && v2 <= 64) s->refill_min = s->refill_min_any = v2; }
    if (const char* e = getenv("SHM_REFILL_MIN_ANY")) { int v2 = atoi(e); if (v2 >= 1 && v2 <= 64) s->refill_min_any = v2; }
    if (const char* e = getenv("SHM_LEAF_MIN")) { int v2 = atoi(e); if (v2 >= 1 && v2 <= 64) s->leaf_min = v2; }
    if (const char* e = getenv("SHM_LEAF_MIN_ANY")) { int v2 = atoi(e); if (v2 >= 1 && v2 <= 64) s->leaf_min_any = v2; }
    if (const char* e = getenv("SHM_OTHER_MIN")) { int v2 = atoi(e); if (v2 >= 1 && v2 <= 64) s->other_min = s->other_min_any = v2; }
    if (const char* e = getenv("SHM_TAIL_FUSED_BOUNCE")) { const int v2 = atoi(e); s->tail_fused_bounce = v2 >= 0 ? v2 : 1 << 30; }
    if (const char* e = getenv("SHM_OTHER_MIN_ANY")) { int v2 = atoi(e); if (v2 >= 1 && v2 <= 64) s->other_min_any = v2; }
    if ((rc = wf_trace_prepare(s)) != SHM_OK) return fail(rc);
    DBG("scene: %u nodes, depth %u, trace blocks %d / %d, spill levels %d / %d", (unsigned)f.nodes.size(), f.max_leaf_depth, s->trace3_blocks[0], s->trace3_blocks[1],
        s->spill3_levels[0], s->spill3_levels[1]);
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
    s->dsv.quirks_off = params->disable_reference_quirks ? 1u : 0u;  // SHM_REFERENCE_QUIRKS: every kernel of this render takes s->dsv by value
    const bool random_walk = params->integrator == SHM_INTEGRATOR_RANDOM_WALK;
    // (the random walk keeps 32 B per depth per path beside the path state: its batches are capped at 16 Mi paths)
    const bool staged = use_staged(s, params);
    if ((rc = ensure_workspace(s, random_walk ? std::min<uint64_t>(n_pixels * (uint64_t)n_samples, 1ull << 24) : n_pixels * (uint64_t)n_samples, staged)) != SHM_OK) return rc;
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
    const int shade_blocks = s->n_cu * 4;  // (the fused / vertex / scatter launchers scale this by their own waves per SIMD)
    for (uint64_t p0 = 0; p0 < n_pixels; p0 += pix_per_batch) {
        uint32_t n_pix = (uint32_t)std::min<uint64_t>(pix_per_batch, n_pixels - p0);
        uint32_t total = n_pix * (uint32_t)n_samples;
        const uint32_t* pixels = s->d_pixels + p0;
        // the fused kernel's own scene class under the path integrator: bounce 0 runs on known constants (k_generate<., LEAN>, ShadeArgs::first_bounce)
        const bool lean_first = (!staged || (first_bounce_candidate(s) && params->force_diffuse == 0)) && !random_walk && params->integrator != SHM_INTEGRATOR_SIMPLE_PATH && !s->pa.aux0 && s->pa.rng0;
        // triangle scenes without textures under the path integrator: every kernel that reads the render's hit array is a TRI_ONLY one, and none reads a triangle hit's t —
        // the closest-hit launches write {primitive, b0, b1, b2}, 16 bytes per path instead of the 32-byte ShmHit
        // (round 5: in scenes with textures too — their kernels, compiled for general geometry, read either record form: load_hit_tri)
        // (... and — the split form — in scenes with spheres / bilinear patches: a triangle hit is its 16-byte record, a sphere / patch hit flags a second one with t and phi
        //  (HIT_HAS_SECOND, wavefront.h). Not with instances: a hit inside one names it in the 32-byte record, patched when the instance's marker is popped)
        s->pa.hit16 = ((!s->flat.has_spheres || !s->flat.has_instances) && params->integrator == SHM_INTEGRATOR_PATH && !random_walk) ? 1u : 0u;
        s->pa.hit2 = (s->pa.hit16 && s->flat.has_spheres) ? reinterpret_cast<const float4*>(s->pa.hit) + s->capacity : nullptr;
        if (s->pa.aux0)
            hipLaunchKernelGGL(k_generate<true>, dim3((total + SHADE_BLOCK - 1) / SHADE_BLOCK), dim3(SHADE_BLOCK), 0, s->stream, s->dsv, s->pa, pixels, n_pix,
                               sample_begin, n_samples, *params, s->d_q_active[0], s->d_qs, s->pix_group);
        else if (lean_first)
            hipLaunchKernelGGL((k_generate<false, true>), dim3((total + SHADE_BLOCK - 1) / SHADE_BLOCK), dim3(SHADE_BLOCK), 0, s->stream, s->dsv, s->pa, pixels, n_pix,
                               sample_begin, n_samples, *params, s->d_q_active[0], s->d_qs, s->pix_group);
        else
            hipLaunchKernelGGL(k_generate<false>, dim3((total + SHADE_BLOCK - 1) / SHADE_BLOCK), dim3(SHADE_BLOCK), 0, s->stream, s->dsv, s->pa, pixels, n_pix,
                               sample_begin, n_samples, *params, s->d_q_active[0], s->d_qs, s->pix_group);
        LAUNCH_TRY("k_generate");
        int cur = 0;
        // Small batches are tail-dominated (the last rays of a persistent traversal launch take ~0.5 ms whatever its size): there
        // K3 of bounce b runs on a second stream beside K2 of bounce b+1 — they are independent: K3 reads the shadow buffers and
        // adds into L, K2 reads the extension rays and writes hit records — and the next shade launch waits for both. Large
        // batches (the 1-GPU headline frame) keep everything on one stream, so each kernel has the device to itself.
        // (a scene shaded by the diverted fused kernel AND the LayeredBxDF scatter kernel runs in this mode at any batch size: its any-hit share is large and its
        //  two shading kernels are bound differently — coated S3 at 256 spp 447 -> 437.5 ms; the headline frame is indifferent, C4 loses 1 %)
        const bool mixed_lean_layered = staged && s->lean_divert && s->flat.has_class[CLASS_LAYERED] && params->force_diffuse == 0;
        const bool overlap_batch = s->overlap_paths > 0 && ((uint64_t)total < s->overlap_paths || mixed_lean_layered) && params->max_depth > 0;
        // ... and, whatever the batch size, the LATE bounces of a deep render: from bounce `late_overlap_bounce` on the queues hold a few percent of the paths
        // (C4: 7 % at bounce 6, 1 % at 12) and every launch is a tail (profiles/r03_c4_per_bounce.txt): K3 beside the next K2, the class scatter kernels beside each other.
        // C4 frame 541 -> 522 ms at 6 (524-528 at 10, 528-534 at 16, 530 at 4)
        constexpr int late_overlap_bounce = 6;
        bool overlap = overlap_batch;
        hipStream_t any_stream = overlap ? s->stream2 : s->stream;
        used_overlap = used_overlap || overlap;
        hipEvent_t k3_done = nullptr;
        // all-diffuse triangle scenes with 16-byte hit records: the records are double-buffered by bounce parity in the two halves of the ShmHit allocation, so that the
        // previous vertex's record — all the next vertex's emitter MIS weight needs (k_shade.inl, k_emit_jobs) — is still there and no vertex writes anything for it
        ShmHit* const hit_base = s->pa.hit;
        // (s->pa is the scene's one PathArrays: the per-bounce overrides — hit, hit_prev — and the per-render record form, hit16, are taken back when the batch is through,
        //  so that whoever reads s->pa.hit outside a render — the public trace entry points, dist.hip — finds the allocation's base and 32-byte records)
        struct HitRestore { ShmScene* sc; ShmHit* base; ~HitRestore() { sc->pa.hit = base; sc->pa.hit_prev = nullptr; sc->pa.hit16 = 0u; sc->pa.hit2 = nullptr; } } hit_restore{s, hit_base};
        const bool hit_kept = s->pa.hit16 && !s->flat.has_spheres && scene_is_lean(s) && !staged && !random_walk;  // (the split form keeps its second records in the other half)
        for (int bounce = 0; bounce <= params->max_depth; ++bounce) {
            if (hit_kept) {
                float4* const h16 = reinterpret_cast<float4*>(hit_base);
                s->pa.hit = reinterpret_cast<ShmHit*>(h16 + (size_t)(bounce & 1) * s->capacity);
                s->pa.hit_prev = h16 + (size_t)((bounce + 1) & 1) * s->capacity;
            }
            const int sh = bounce & 1;
            if (!overlap && s->overlap_paths > 0 && bounce >= late_overlap_bounce) {
                // (the switch is safe at a bounce boundary: everything so far was ordered on the render stream)
                overlap = true;
                any_stream = s->stream2;
                used_overlap = true;
            }
            hipEvent_t a = ev.get(), b = ev.get();
            hipEventRecord(a, s->stream);
            const bool first_lean = lean_first && bounce == 0;  // (the identity queue was not written: K2 takes slot = queue index, k_shade knows the constants)
            if ((rc = first_lean ? wf_launch_trace(s, false, s->stream, nullptr, nullptr, total, s->pa.ray, s->pa.hit, nullptr, nullptr, nullptr, (int)s->pa.hit16)
                                 : wf_launch_trace(s, false, s->stream, s->d_q_active[cur], &s->d_qs->n_active[cur], 0, s->pa.ray, s->pa.hit, nullptr, nullptr, nullptr, (int)s->pa.hit16)) != SHM_OK) return rc;
            hipEventRecord(b, s->stream);
            ev_closest.push_back({a, b});
            if (overlap && k3_done) hipStreamWaitEvent(s->stream, k3_done, 0);  // shade(b) touches L and refills the shadow buffers
            {
                hipEvent_t s0 = ev.get(), s1 = ev.get();
                hipEventRecord(s0, s->stream);
                const ShadeArgs sa{s->stream, cur, *params, sh, shade_blocks, first_lean ? 1 : 0, hit_kept ? 1 : 0};
                const bool tri_only = !s->flat.has_spheres;
                int n_classes_present = 0;
                for (int c = 0; c < N_BXDF_CLASSES; ++c) n_classes_present += s->flat.has_class[c] ? 1 : 0;
                // a triangle scene with several BxDF classes but without textures or coated materials: ONE fused all-materials launch per bounce instead of the staged
                // four or five, from bounce `tail_fused_bounce` on. Rounds 3-4, chunks unsorted: only the late bounces paid (C4 frame 522-528 ms staged throughout,
                // 510-512 from bounce 6, 511-513 from 8). Round 5, chunks counting-sorted by material (k_shade_tail_sorted.hip): the earlier the better — C4 403.2 ms
                // from bounce 8, 399 from 4, 388 from 2, 378 from 1, 365 from 0: the default (SHM_TAIL_FUSED_BOUNCE, negative = never; read at scene creation)
                // (with textures: where there is more than one BxDF class to sort — one class: the fused textured kernel's 128 spilled VGPRs cost more than the staged pair's
                //  parameter block — and no split pass takes most hits away from the textured kernels; an environment map alone is no texture)
                const bool fused_tex_ok = !s->split_pass && n_classes_present > 1;
                const bool fused_all = staged && bounce >= s->tail_fused_bounce && !s->flat.has_class[CLASS_LAYERED] && params->force_diffuse == 0 &&
                                       (!s->flat.has_textures || env_plain_scene(s) || fused_tex_ok);
                if (fused_all) {
                    if (env_plain_scene(s)) rc = tri_only ? wf_launch_shade_tail_sorted_env(s, sa) : wf_launch_shade_fused_gen_env(s, sa);
                    else rc = tri_only ? (s->flat.has_textures ? wf_launch_shade_fused_tex(s, sa) : wf_launch_shade_tail_sorted(s, sa))
                                       : (s->flat.has_textures ? wf_launch_shade_fused_gen_tex(s, sa) : wf_launch_shade_fused_gen(s, sa));
                } else if (staged) {
                    // hit half (interaction, emission, get_bsdf -> parameter block, class queues), then one scattering kernel per BxDF
                    // class the scene holds, each over its own material-sorted queue
                    const bool env = env_plain_scene(s) && params->force_diffuse == 0;  // (the K_ENV_LIGHT units: the class without textures + the image light)
                    const bool has_tex = s->flat.has_textures && !env;
                    const bool split = s->split_pass && s->d_q_split && s->d_q_lean && params->force_diffuse == 0;
                    ShadeArgs va = sa;  // (k_vertex's arguments: its queue is q_split behind the split pass)
                    if (split) {  // plain-diffuse hits -> q_lean (their whole vertex in the lean fused kernel, below), the rest -> q_split for k_vertex
                        hipLaunchKernelGGL(k_split_plain, dim3(s->n_cu * 8), dim3(SHADE2_BLOCK), 0, s->stream, s->dsv, s->pa, s->d_q_active[cur], s->d_q_lean, s->d_q_split, s->d_qs, cur);
                        LAUNCH_TRY("k_split_plain");
                        va.q_in = s->d_q_split;
                        va.n_in = &s->d_qs->n_split;
                    }
                    rc = has_tex ? wf_launch_vertex_tex(s, va) : (env ? (tri_only ? wf_launch_vertex_tri_env(s, va) : wf_launch_vertex_gen_env(s, va))
                                                                      : (tri_only ? wf_launch_vertex_tri(s, va) : wf_launch_vertex_gen(s, va)));
                    // the hits k_vertex diverted (plain diffuse materials): their whole vertex in the fused kernel — the first member of the group below
                    const bool lean_too = s->lean_divert && s->d_q_lean && params->force_diffuse == 0 && (!has_tex || split);
                    // The classes' scatter kernels are independent of each other (own queue each, disjoint paths, wave-aggregated atomics on the
                    // shared next / shadow queues): the first runs on the render stream, the others beside it on their own streams, and the render
                    // stream waits for them — for small batches only (the same threshold as the K3 / K2 overlap above), where the launches are
                    // tail-dominated: textured Cornell 512^2 x 64 1 376 -> 1 486 Mray/s. With large queues each kernel fills the device by itself and
                    // sharing it costs (coated S3 at 256 spp 1 944 -> 1 864), and C4's late bounces did not gain (2 044 either way).
                    int n_cls = lean_too ? 1 : 0;
                    for (int c = 0; c < N_BXDF_CLASSES; ++c) n_cls += s->flat.has_class[c] ? 1 : 0;
                    hipEvent_t vertex_done = nullptr;
                    // ... and at any batch size where the diverted fused kernel (latency-bound, three waves per SIMD) has the LayeredBxDF class's scatter kernel
                    // (issue-bound, two) to run beside: complementary bounds (coated S3 at 256 spp: see DESIGN.md section 6)
                    const bool group_big = lean_too && s->flat.has_class[CLASS_LAYERED];
                    if ((overlap || group_big) && n_cls > 1) { vertex_done = ev.get(); hipEventRecord(vertex_done, s->stream); }  // (before the first class's launch)
                    std::vector<hipEvent_t> side_done;
                    int k_cls = 0;
                    auto scatter_on = [&](int cls, auto&& launch) {
                        if (rc != SHM_OK || !(cls < 0 ? lean_too : s->flat.has_class[cls])) return;
                        ShadeArgs sc = sa;
                        const bool side = k_cls > 0 && vertex_done != nullptr;
                        if (side) {
                            sc.stream = s->stream_cls[k_cls - 1];
                            hipStreamWaitEvent(sc.stream, vertex_done, 0);
                        }
                        rc = launch(sc);
                        if (side && rc == SHM_OK) {
                            hipEvent_t done = ev.get();
                            hipEventRecord(done, sc.stream);
                            side_done.push_back(done);
                        }
                        ++k_cls;
                    };
                    scatter_on(-1, [&](const ShadeArgs& x) { return s->flat.has_image_light ? (tri_only ? wf_launch_shade_lean_env_diverted(s, x) : wf_launch_shade_lean_gen_env_diverted(s, x))
                                                                                              : (tri_only ? wf_launch_shade_lean_diverted(s, x) : wf_launch_shade_lean_gen_diverted(s, x)); });
                    scatter_on(CLASS_DIFFUSE, [&](const ShadeArgs& x) { return env ? wf_launch_scatter_diffuse_env(s, x, tri_only) : wf_launch_scatter_diffuse(s, x, tri_only, has_tex); });
                    scatter_on(CLASS_CONDUCTOR, [&](const ShadeArgs& x) { return env ? wf_launch_scatter_conductor_env(s, x, tri_only) : wf_launch_scatter_conductor(s, x, tri_only, has_tex); });
                    scatter_on(CLASS_DIELECTRIC, [&](const ShadeArgs& x) { return env ? wf_launch_scatter_dielectric_env(s, x, tri_only) : wf_launch_scatter_dielectric(s, x, tri_only, has_tex); });
                    scatter_on(CLASS_LAYERED, [&](const ShadeArgs& x) {
                        // (options.force_diffuse replaces the BxDF inside this half: the one-pass kernel has that code)
                        if (params->force_diffuse == 0 && s->capacity < (1u << 30)) {  // (its jobs carry two flag bits above the path index)
                            if (env) return tri_only ? wf_launch_scatter_layered_staged_tri_env(s, x) : wf_launch_scatter_layered_staged_gen_env(s, x);
                            return has_tex ? wf_launch_scatter_layered_staged_tex(s, x) : (tri_only ? wf_launch_scatter_layered_staged_tri(s, x) : wf_launch_scatter_layered_staged_gen(s, x));
                        }
                        // (`env` implies params->force_diffuse == 0: the one-pass kernel is reached from here only past 2^30 paths of workspace, which ensure_workspace never grants —
                        //  its K_ENV_LIGHT units left the library in round 6 — and under options.force_diffuse, where the textured class's units run)
                        return has_tex ? wf_launch_scatter_layered_tex(s, x) : (tri_only ? wf_launch_scatter_layered_tri(s, x) : wf_launch_scatter_layered_gen(s, x)); });
                    for (hipEvent_t e : side_done) hipStreamWaitEvent(s->stream, e, 0);
                }
                else if (random_walk) rc = wf_launch_shade_randomwalk(s, sa, cap_eff);
                else if (params->integrator == SHM_INTEGRATOR_SIMPLE_PATH) rc = wf_launch_shade_simple(s, sa);
                else if (env_lean_scene(s)) rc = tri_only ? wf_launch_shade_lean_env(s, sa) : wf_launch_shade_lean_gen_env(s, sa);
                else rc = tri_only ? wf_launch_shade_lean(s, sa) : wf_launch_shade_lean_gen(s, sa);
                if (rc != SHM_OK) return rc;
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
                if ((rc = wf_launch_trace(s, true, any_stream, s->d_q_shadow, &s->d_qs->n_shadow[sh], 0, s->pa.shadow_ray, nullptr, nullptr, s->pa.L, s->pa.shadow_contrib)) != SHM_OK) return rc;
                hipEventRecord(d, any_stream);
                ev_any.push_back({c, d});
                k3_done = d;
            }
            if (dbg_on()) {  // queue sizes per bounce (costs a sync: debug only)
                QueueState q;
                hipStreamSynchronize(s->stream);
                hipStreamSynchronize(any_stream);
                hipMemcpy(&q, s->d_qs, sizeof(q), hipMemcpyDeviceToHost);
                DBG("bounce %d: traced %u, next %u, shadow %u, emitter hits deferred %u, diverted %u", bounce, q.n_active[cur], q.n_active[cur ^ 1], q.n_shadow[sh], q.n_emit, q.n_lean);
            }
            hipLaunchKernelGGL(k_next_bounce, dim3(1), dim3(1), 0, s->stream, s->d_qs, cur, sh ^ 1);
            cur ^= 1;
        }
        if (overlap && k3_done) hipStreamWaitEvent(s->stream, k3_done, 0);  // the film reads L
        if (random_walk)
            if ((rc = wf_launch_fold_randomwalk(s, s->stream, cap_eff, total)) != SHM_OK) return rc;
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
    // the tiles takes all 256 spp at once), without changing a single film sum (shm_render_wave is the one-launch-per-wave entry: tests/test_gpu_parity.py holds the two against each other).
    const bool fuse = true;
    int spp = params->samples_per_pixel;
    uint64_t n_pixels = 0;
    for (uint32_t t = 0; tiles && t < n_tiles; ++t)
        n_pixels += (uint64_t)std::max(0, tiles[t].x1 - tiles[t].x0) * (uint64_t)std::max(0, tiles[t].y1 - tiles[t].y0);
    HIP_TRY(hipSetDevice(s->device));
    const int max_fuse = (int)std::min<uint64_t>(std::max<uint64_t>(64, n_pixels ? workspace_cap(s, use_staged(s, params)) / n_pixels : 64), 1u << 20);
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
        int rc = any ? wf_launch_trace(s, true, s->stream, nullptr, nullptr, n, (const ShmRay*)rays_dev, nullptr, (uint8_t*)out_dev, nullptr, nullptr)
                     : wf_launch_trace(s, false, s->stream, nullptr, nullptr, n, (const ShmRay*)rays_dev, (ShmHit*)out_dev, nullptr, nullptr, nullptr);
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