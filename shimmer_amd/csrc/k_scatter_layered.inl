// k_scatter_layered.inl — the scattering half of a LayeredBxDF vertex (CoatedDiffuse / CoatedConductor; integrator.rs:836-892, sample_ld :897-963, bxdf.rs:883-1620)
// as FOUR DENSE STAGES PER WAVE instead of one pass per vertex. A coated vertex runs up to three random walks through the coating — LayeredBxDF::f and ::pdf for
// next-event estimation, ::sample_f for the next direction — whose lengths differ from lane to lane (sample_f: 2, 4, 5 ... 10 interface events; half of the walks
// leave after two), so a wave that takes its 64 vertices through all of it waits for its longest walk at every step (k_scatter<CLASS_LAYERED>: 37 of 64 lanes).
// Here a wave keeps three job buffers in LDS and always runs the stage that has 64 jobs waiting:
//   A  vertex set-up (64 queue entries): the sampler dimensions in the reference's order, the light sample -> an NEE job {wo, wi, p_l, beta} when its f can be non-zero;
//      the top interface's sample (layered_sample_begin) -> an EPILOGUE job when it reflects, a WALK job {w, f, pdf, z, depth, generator} when it enters the coating
//   B  64 walk jobs advanced TWO steps (layered_sample_step: down to the bottom interface and back up to the top one — with albedo 0 every deposited walk stands at
//      the same parity, so the interface dispatch is wave-uniform); a walk that leaves becomes an epilogue job, a dead one disappears, the others are deposited again
//   C  64 epilogue jobs: LayeredBxDF::pdf of the sampled direction, the throughput update, Russian roulette, the spawned ray (integrator.rs:859-891)
//   N  64 NEE jobs: LayeredBxDF::f and ::pdf, the MIS weight, the deferred shadow ray's contribution (integrator.rs:924-962)
// Every lane of every stage has work (but for the last passes of a wave), and a stage holds ONE walk's state beside its own inputs, not three. Per path the
// arithmetic, the sampler dimensions and the walks' own generators are what k_scatter<CLASS_LAYERED> computes (which stays: options.force_diffuse runs it): films
// and counters are bit-identical. Waves are independent: no barrier after the table staging, queue slots by one wave-level atomic per pass whose result is consumed
// a pass later (its latency hides behind the next pass's gathers).
#pragma once
#include "wavefront.h"

// K_ENV_LIGHT (a translation unit's switch, round 5): compile ImageInfinitelight's look-up / sample / pdf into a kernel of the class WITHOUT textures — a scene whose only
// image is an environment map needs nothing else of the textured class (k_shade_lean_env.hip says why); the *_env.hip units define it
#ifndef K_ENV_LIGHT
#define K_ENV_LIGHT false
#endif
#if K_ENV_LIGHT  // (the *_env.hip units' kernels carry their own names: a kernel trace tells them from the units without the light — tools/kernel_coverage.py)
#define k_scatter_layered k_scatter_layered_env
#endif
namespace {

constexpr int LJ_CAP = 2 * WAVE;  // per buffer: at most 63 waiting + 64 new (a stage runs as soon as 64 wait, and before anything is added)
constexpr int LJ_NEE_WORDS = 12, LJ_WALK_WORDS = 16, LJ_EPI_WORDS = 10;
constexpr int LJ_WAVE_WORDS = (LJ_NEE_WORDS + LJ_WALK_WORDS + LJ_EPI_WORDS) * LJ_CAP;  // 19 KB per wave, 76 KB per workgroup: two workgroups per CU
constexpr uint32_t LJ_PATH_MASK = 0x3fffffffu;  // (a path index is below 2^30: 288 GB / 348 B per path; the two bits above carry a job's flags)

// development build (-DLJ_CENSUS): passes, busy lanes and shader-clock ticks per stage, printed by wf_layered_census() at scene destruction
#ifdef LJ_CENSUS
__device__ unsigned long long g_lj_census[24];
#define LJ_COUNT(stage, lanes_pred)                                                                                                  \
    do { const unsigned long long _m = __ballot(lanes_pred); if (lane == 0) { atomicAdd(&g_lj_census[(stage) * 3], 1ull); atomicAdd(&g_lj_census[(stage) * 3 + 1], (unsigned long long)__popcll(_m)); } } while (0)
#define LJ_TICK0() const unsigned long long _t0 = __builtin_readcyclecounter()
#define LJ_TICK1(stage) do { if (lane == 0) atomicAdd(&g_lj_census[(stage) * 3 + 2], (unsigned long long)(__builtin_readcyclecounter() - _t0)); } while (0)
#define LJ_TALLY(slot, pred) do { const unsigned long long _m = __ballot(pred); if (lane == 0) atomicAdd(&g_lj_census[slot], (unsigned long long)__popcll(_m)); } while (0)
#else
#define LJ_COUNT(stage, lanes_pred) do {} while (0)
#define LJ_TICK0() do {} while (0)
#define LJ_TICK1(stage) do {} while (0)
#define LJ_TALLY(slot, pred) do {} while (0)
#endif

// a queue push whose slot arrives a pass later: the atomic is issued when the entries are known, its result read when the next push (or the end) needs the registers
struct DeferredPush {
    uint32_t base_v = 0, path = 0;
    unsigned long long mask = 0ull;  // wave-uniform
};
__device__ __forceinline__ void push_flush(DeferredPush& d, uint32_t* __restrict__ q, uint32_t lane) {
    if (d.mask != 0ull) {
        const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)d.base_v, __ffsll((long long)d.mask) - 1);
        if ((d.mask >> lane) & 1ull) q[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(d.mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)d.mask, 0u))] = d.path;
        d.mask = 0ull;
    }
}
__device__ __forceinline__ void push_begin(DeferredPush& d, uint32_t* counter, bool pred, uint32_t path, uint32_t lane) {
    d.mask = __ballot(pred);
    d.path = path;
    if (d.mask != 0ull && lane == (uint32_t)(__ffsll((long long)d.mask) - 1)) d.base_v = atomicAdd(counter, (uint32_t)__popcll(d.mask));
}
// the slot of a depositing lane in a job buffer that holds `n` jobs (wave-uniform), and the new count
// Jobs travel between the LANES of a wave through LDS (the job buffers) and, for a few words, through global memory (rec[path].rng, shadow_contrib[path]) with no
// barrier: a wave runs in lockstep. What that leaves to be said is said to the COMPILER: reads of a pass are not to sink below the deposits that reuse their slots, and a
// pass's reads are not to rise above an earlier pass's deposits (advisor, round 4). A wavefront-scope fence + wave barrier: no instruction on wave64, an ordering point.
__device__ __forceinline__ void lj_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ uint32_t deposit_slot(uint32_t& n, bool dep) {
    lj_wave_sync();  // (between a pass's job reads and its deposits)
    const unsigned long long m = __ballot(dep);
    const uint32_t slot = n + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    n = (uint32_t)__builtin_amdgcn_readfirstlane((int)(n + (uint32_t)__popcll(m)));
    return slot;
}

// the BxDF as k_vertex left it in the parameter block (without the shading frame: the walks run in the local frame)
__device__ __forceinline__ void layered_bxdf_of(const SceneView& sv, const PathArrays& pa, uint32_t path, BxDF& b) {
    const float4 p2 = pa.bx[path].bx2;
    const uint32_t meta = __float_as_uint(p2.w);
    b.kind = meta & 0xffu;
    b.max_depth = (int)((meta >> 8) & 0xfffu);
    b.n_samples = (int)(meta >> 20);
    b.strict = (int)sv.quirks_off;
    b.eta = p2.x;
    b.mf.alpha_x = p2.y;
    b.mf.alpha_y = p2.z;
    b.r = ld_spec(pa.bx[path].bx0);
    b.k = ld_spec(pa.bx[path].bx1);
    b.albedo = ld_spec(pa.bx[path].bx3);
    const float4 p4 = pa.bx[path].bx4;
    b.mf2.alpha_x = p4.x; b.mf2.alpha_y = p4.y; b.thickness = p4.z; b.g = p4.w;
    __builtin_assume(b.kind == SHM_MATERIAL_COATED_DIFFUSE || b.kind == SHM_MATERIAL_COATED_CONDUCTOR);  // the class is a property of the queue
}
__device__ __forceinline__ void layered_bsdf_of(const SceneView& sv, const PathArrays& pa, uint32_t path, const float4& c2, BSDF& bsdf, V3& ns) {
    layered_bxdf_of(sv, pa, path, bsdf.bxdf);
    const float4 f = pa.bx[path].fr;
    ns = v3(c2.y, c2.z, c2.w);
    // Frame::from_xz (frame.rs:14-17): y = z cross x
    bsdf.shading_frame.x = v3(f.x, f.y, f.z);
    bsdf.shading_frame.z = ns;
    bsdf.shading_frame.y = cross(ns, bsdf.shading_frame.x);
}
__device__ __forceinline__ Rng path_sampler(const PathArrays& pa, uint32_t path, const ShmRenderParams& params) {
    Rng rng;
    const uint32_t pix = pa.rec[path].pixel;
    const uint2 rs = pa.rec[path].rng;
    rng.state = (uint64_t)rs.x | ((uint64_t)rs.y << 32);
    // inc is a pure function of (pixel, seed): re-derived instead of stored
    uint64_t h = mix_bits(((uint64_t)(pix & 0xffffu) << 32) | (uint64_t)(pix >> 16));
    h = mix_bits(h ^ (params.seed + 0x9e3779b97f4a7c15ULL));
    rng.inc = (h << 1u) | 1u;
    return rng;
}

template <bool TRI_ONLY, bool HAS_TEX>
__device__ __forceinline__ void scatter_layered_staged(const SceneView& sv, const PathArrays& pa, const uint32_t* __restrict__ q_cur, uint32_t* __restrict__ q_next,
                                                       uint32_t* __restrict__ q_shadow, QueueState* qs, int cur, const ShmRenderParams& params, int shadow_parity,
                                                       uint32_t* s_jobs) {
    const uint32_t n = qs->n_scatter[CLASS_LAYERED];
    const uint32_t lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    // the three job buffers of this wave, word-major ([word][slot]: a wave's accesses to one word are consecutive LDS addresses)
    uint32_t* const nee = s_jobs + wave * LJ_WAVE_WORDS;
    uint32_t* const walk = nee + LJ_NEE_WORDS * LJ_CAP;
    uint32_t* const epi = walk + LJ_WALK_WORDS * LJ_CAP;
    uint32_t n_nee = 0, n_walk = 0, n_epi = 0;  // wave-uniform
    uint32_t next_i = (blockIdx.x * (SHADE2_BLOCK / WAVE) + wave) * WAVE;  // this wave's next 64 queue entries
    const uint32_t stride = gridDim.x * SHADE2_BLOCK;
    DeferredPush push_next, push_shadow;
    auto deposit_epilogue = [&](bool dep, uint32_t path, const BSDFSample& bs) {
        const uint32_t slot = deposit_slot(n_epi, dep);
        if (dep) {
            uint32_t* jw = epi + slot;
            jw[0] = path;
            for (int c = 0; c < 4; ++c) jw[(1 + c) * LJ_CAP] = __float_as_uint(bs.f.v[c]);
            jw[5 * LJ_CAP] = __float_as_uint(bs.wi.x); jw[6 * LJ_CAP] = __float_as_uint(bs.wi.y); jw[7 * LJ_CAP] = __float_as_uint(bs.wi.z);
            jw[8 * LJ_CAP] = __float_as_uint(bs.pdf);
            jw[9 * LJ_CAP] = bs.flags;
        }
    };
    auto deposit_walk = [&](bool dep, uint32_t path, const LayeredWalk& k, bool regularized) {
        const uint32_t slot = deposit_slot(n_walk, dep);
        if (dep) {
            uint32_t* jw = walk + slot;
            jw[0] = path;
            jw[1 * LJ_CAP] = __float_as_uint(k.w.x); jw[2 * LJ_CAP] = __float_as_uint(k.w.y); jw[3 * LJ_CAP] = __float_as_uint(k.w.z);
            for (int c = 0; c < 4; ++c) jw[(4 + c) * LJ_CAP] = __float_as_uint(k.f.v[c]);
            jw[8 * LJ_CAP] = __float_as_uint(k.pdf);
            jw[9 * LJ_CAP] = __float_as_uint(k.z);
            jw[10 * LJ_CAP] = __float_as_uint(k.wo_z);
            jw[11 * LJ_CAP] = (uint32_t)k.depth | ((uint32_t)k.specular_path << 12) | ((uint32_t)k.flip_wi << 13) | ((uint32_t)regularized << 14);  // (depth <= max_depth < 4096: 12 bits, as in bx2)
            jw[12 * LJ_CAP] = (uint32_t)k.rng.state; jw[13 * LJ_CAP] = (uint32_t)(k.rng.state >> 32);
            jw[14 * LJ_CAP] = (uint32_t)k.rng.inc; jw[15 * LJ_CAP] = (uint32_t)(k.rng.inc >> 32);
        }
    };
    // what BSDF::sample_f adds behind BxDF::sample_f (bsdf.rs:75-79)
    auto usable = [](const BSDFSample& bs) { return !(is_zero(bs.f) || bs.pdf == 0.0f || bs.wi.z == 0.0f); };

    for (;;) {
        lj_wave_sync();  // (before a pass consumes what an earlier one deposited)
        const bool more = next_i < n;
        if (n_epi >= (uint32_t)WAVE || (!more && n_walk == 0u && n_epi > 0u)) {
            // ---- C: a wave of sampled directions — integrator.rs:859-891 ----
            const uint32_t take = n_epi < (uint32_t)WAVE ? n_epi : (uint32_t)WAVE;
            n_epi -= take;
            LJ_COUNT(2, lane < take);
            LJ_TICK0();
            push_flush(push_next, q_next, lane);
            bool alive = false;
            uint32_t path = 0;
            if (lane < take) {
                const uint32_t* jw = epi + n_epi + lane;
                path = jw[0];
                BSDFSample bs;
                for (int c = 0; c < 4; ++c) bs.f.v[c] = __uint_as_float(jw[(1 + c) * LJ_CAP]);
                bs.wi = v3(__uint_as_float(jw[5 * LJ_CAP]), __uint_as_float(jw[6 * LJ_CAP]), __uint_as_float(jw[7 * LJ_CAP]));
                bs.pdf = __uint_as_float(jw[8 * LJ_CAP]);
                bs.flags = jw[9 * LJ_CAP];
                bs.eta = 1.0f;  // (a walk that leaves carries 1; a reflection at the top interface is not transmissive, its eta is never read)
                bs.pdf_is_proportional = true;
                const float4 c0 = pa.ctx[path].c0, c1 = pa.ctx[path].c1, c2 = pa.ctx[path].c2;
                P3i si_pi;
                si_pi.x = iv2(c0.x, c0.w);
                si_pi.y = iv2(c0.y, c1.x);
                si_pi.z = iv2(c0.z, c1.y);
                const V3 si_n = v3(c1.z, c1.w, c2.x);
                BSDF bsdf;
                V3 ns;
                layered_bsdf_of(sv, pa, path, c2, bsdf, ns);
                const float4* rp = reinterpret_cast<const float4*>(pa.ray + path);
                const float4 r0 = rp[0], r1 = rp[1];
                const V3 wo = -v3(r0.w, r1.x, r1.y);  // li()'s wo = -ray.d (integrator.rs:844)
                Spec beta = ld_spec(pa.rec[path].beta);
                const uint32_t fl = pa.rec[path].flags;
                const int depth = (int)(fl & 0xffu) + 1;
                bool any_non_specular_bounces = (fl >> 9) & 1u;
                if (params.regularize && any_non_specular_bounces) bxdf_regularize(bsdf.bxdf);
                Float eta_scale = pa.rec[path].pb_eta.y;
                Rng rng = path_sampler(pa, path, params);  // (stage A left the state behind this vertex's seven dimensions)
                bs.wi = bsdf.shading_frame.from_local(bs.wi);  // bsdf.rs:80
                // integrator.rs:859-872
                beta = beta * (bs.f * abs_dot(bs.wi, ns) / bs.pdf);
                const Float p_b = bsdf_pdf(bsdf, wo, bs.wi, REFLTRANS_ALL);  // (a LayeredBxDF sample's pdf is always proportional)
                const bool specular_bounce = flags_is_specular(bs.flags);
                any_non_specular_bounces |= !specular_bounce;
                if (flags_is_transmissive(bs.flags)) eta_scale *= sqr(bs.eta);
                const V3 no = offset_ray_origin(si_pi, si_n, bs.wi);  // integrator.rs:875 -> interaction.rs:68-75
                // integrator.rs:878-891: Russian roulette
                alive = true;
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
                    uint32_t aux_bit = 0u;
                    if (HAS_TEX && (fl & (1u << 10)) && (bs.flags == BXDF_SPECULAR_REFLECTION || bs.flags == BXDF_SPECULAR_TRANSMISSION)) {
                        // spawn_ray_with_differentials, interaction.rs:430-514 (only specular bounces carry differentials on)
                        V3 si_wo = wo;
                        if (!TRI_ONLY) { const float4 w4 = pa.bx[path].siwo; si_wo = v3(w4.x, w4.y, w4.z); }
                        const float4 d0 = pa.dd0[path], d1 = pa.dd1[path], d2 = pa.dd2[path];
                        AuxRays na = spawn_ray_differentials_pre(si_pi.mid(), si_wo, ns, v3(d0.x, d0.y, d0.z), v3(d0.w, d1.x, d1.y), v3(d1.z, d1.w, d2.x),
                                                                 v3(d2.y, d2.z, d2.w), ld_aux(pa, path), bs.wi, bs.flags, bs.eta);
                        if (na.has) { st_aux(pa, path, na); aux_bit = 1u << 10; }
                    }
                    pa.ray[path] = nr;
                    pa.rec[path].beta = st_spec(beta);
                    pa.rec[path].pb_eta = make_float2(p_b, eta_scale);
                    // (the CtxRec already holds this vertex's context: the next vertex's prev_intr_ctx)
                    pa.rec[path].rng = make_uint2((uint32_t)rng.state, (uint32_t)(rng.state >> 32));
                    pa.rec[path].flags = (uint32_t)depth | ((uint32_t)specular_bounce << 8) | ((uint32_t)any_non_specular_bounces << 9) | aux_bit;
                }
            }
            push_begin(push_next, &qs->n_active[cur ^ 1], alive, path, lane);
            LJ_TICK1(2);
            continue;
        }
        if (n_walk >= (uint32_t)WAVE || (!more && n_walk > 0u)) {
            // ---- B: a wave of walks, two interface events further (bxdf.rs:1300-1403) ----
            const uint32_t take = n_walk < (uint32_t)WAVE ? n_walk : (uint32_t)WAVE;
            n_walk -= take;
            LJ_COUNT(1, lane < take);
            LJ_TICK0();
            int status = WALK_FAILED;
            uint32_t path = 0;
            bool regularized = false;
            LayeredWalk k;
            BSDFSample bs = bsdf_sample(spec_const(0.0f), v3s(0.0f), 0.0f, 0u);
            if (lane < take) {
                const uint32_t* jw = walk + n_walk + lane;
                path = jw[0];
                k.w = v3(__uint_as_float(jw[1 * LJ_CAP]), __uint_as_float(jw[2 * LJ_CAP]), __uint_as_float(jw[3 * LJ_CAP]));
                for (int c = 0; c < 4; ++c) k.f.v[c] = __uint_as_float(jw[(4 + c) * LJ_CAP]);
                k.pdf = __uint_as_float(jw[8 * LJ_CAP]);
                k.z = __uint_as_float(jw[9 * LJ_CAP]);
                k.wo_z = __uint_as_float(jw[10 * LJ_CAP]);
                const uint32_t meta = jw[11 * LJ_CAP];
                k.depth = (int)(meta & 0xfffu);
                k.specular_path = (meta >> 12) & 1u;
                k.flip_wi = (meta >> 13) & 1u;
                regularized = (meta >> 14) & 1u;
                k.rng.state = (uint64_t)jw[12 * LJ_CAP] | ((uint64_t)jw[13 * LJ_CAP] << 32);
                k.rng.inc = (uint64_t)jw[14 * LJ_CAP] | ((uint64_t)jw[15 * LJ_CAP] << 32);
                BxDF b;
                layered_bxdf_of(sv, pa, path, b);
                if (regularized) bxdf_regularize(b);
                status = WALK_CONTINUES;
#pragma unroll 1
                for (int s = 0; s < 2 && status == WALK_CONTINUES; ++s) status = layered_sample_step(b, MODE_RADIANCE, k, bs);
            }
            // (every job of the pass has been read: the deposits below may reuse their slots)
            deposit_walk(status == WALK_CONTINUES, path, k, regularized);
            deposit_epilogue(status == WALK_DONE && usable(bs), path, bs);
            LJ_TALLY(12, status == WALK_CONTINUES);
            LJ_TALLY(13, status == WALK_DONE);
            LJ_TICK1(1);
            continue;
        }
        if (n_nee >= (uint32_t)WAVE || (!more && n_nee > 0u)) {
            // ---- N: a wave of next-event estimations — integrator.rs:924-962, from the BSDF's f on ----
            const uint32_t take = n_nee < (uint32_t)WAVE ? n_nee : (uint32_t)WAVE;
            n_nee -= take;
            LJ_COUNT(3, lane < take);
            LJ_TICK0();
            push_flush(push_shadow, q_shadow, lane);
            bool queued = false;
            uint32_t path = 0;
            if (lane < take) {
                const uint32_t* jw = nee + n_nee + lane;
                const uint32_t w0 = jw[0];
                path = w0 & LJ_PATH_MASK;
                BSDF bsdf;
                V3 ns;
                layered_bsdf_of(sv, pa, path, pa.ctx[path].c2, bsdf, ns);
                if (w0 & 0x80000000u) bxdf_regularize(bsdf.bxdf);
                const Spec l = ld_spec(pa.shadow_contrib[path]);  // (the light's radiance, parked by stage A where this job leaves its result)
                const V3 si_wo = v3(__uint_as_float(jw[1 * LJ_CAP]), __uint_as_float(jw[2 * LJ_CAP]), __uint_as_float(jw[3 * LJ_CAP]));
                const V3 wi = v3(__uint_as_float(jw[4 * LJ_CAP]), __uint_as_float(jw[5 * LJ_CAP]), __uint_as_float(jw[6 * LJ_CAP]));
                Spec f = bsdf_f(bsdf, si_wo, wi) * abs_dot(wi, ns);
                if (!is_zero(f)) {
                    Spec beta;
                    for (int c = 0; c < 4; ++c) beta.v[c] = __uint_as_float(jw[(8 + c) * LJ_CAP]);
                    const Float p_l = __uint_as_float(jw[7 * LJ_CAP]);
                    Spec ld;
                    if (w0 & 0x40000000u) {  // a delta light
                        ld = l * f / p_l;
                    } else {
                        Float pb2 = bsdf_pdf(bsdf, si_wo, wi, REFLTRANS_ALL);
                        Float w_l = power_heuristic(1, p_l, 1, pb2);
                        ld = w_l * l * f / p_l;
                    }
                    pa.shadow_contrib[path] = st_spec(beta * ld);
                    queued = true;
                }
            }
            push_begin(push_shadow, &qs->n_shadow[shadow_parity], queued, path, lane);
            LJ_TALLY(14, queued);
            LJ_TICK1(3);
            continue;
        }
        if (!more) break;
        // ---- A: 64 vertices as k_vertex left them — the sampler's dimensions, the top interface, the light sample ----
        // The dimensions are drawn in the reference's order (NEE's three, then sample_f's three: integrator.rs:837-857); the work behind them runs in the
        // order that keeps the fewest registers alive: the top interface's sample first (it needs the BxDF), the light sample — the register-hungriest
        // island of the vertex — last, with only the vertex geometry beside it.
        const uint32_t i = next_i + lane;
        next_i += stride;
        LJ_COUNT(0, i < n);
        LJ_TICK0();
        uint32_t path = 0;
        bool regularized = false, do_nee = false;
        int status = WALK_FAILED;
        Float nee_u = 0.0f;
        V2 u_light = v2(0.0f, 0.0f);
        uint32_t bf = 0u, kind = SHM_MATERIAL_COATED_DIFFUSE;
        V3 si_wo = v3s(0.0f);
        {
            LayeredWalk k;
            BSDFSample bs = bsdf_sample(spec_const(0.0f), v3s(0.0f), 0.0f, 0u);
            if (i < n) {
                path = q_cur[i];
                BSDF bsdf;
                V3 ns;
                layered_bsdf_of(sv, pa, path, pa.ctx[path].c2, bsdf, ns);
                const float4* rp = reinterpret_cast<const float4*>(pa.ray + path);
                const float4 r0 = rp[0], r1 = rp[1];
                const V3 wo = -v3(r0.w, r1.x, r1.y);  // li()'s wo = -ray.d (integrator.rs:844)
                // intr.wo, what sample_ld and get_bsdf use: bitwise -ray.d for a top-level triangle; a quadric or an instanced primitive carries its own (k_scatter.inl)
                si_wo = wo;
                if (!TRI_ONLY) { const float4 w4 = pa.bx[path].siwo; si_wo = v3(w4.x, w4.y, w4.z); }
                regularized = params.regularize && ((pa.rec[path].flags >> 9) & 1u);
                Rng rng = path_sampler(pa, path, params);
                if (regularized) bxdf_regularize(bsdf.bxdf);
                bf = bsdf_flags(bsdf);
                kind = bsdf.bxdf.kind;
                do_nee = flags_is_non_specular(bf);
                if (do_nee) {
                    nee_u = sampler_get_1d(rng);
                    u_light = sampler_get_2d(rng);
                }
                // integrator.rs:843-857: sample the BSDF — here BSDF::sample_f's entry (bsdf.rs:60-74) and the top interface's sample
                const Float u = sampler_get_1d(rng);
                const V2 u2 = sampler_get_2d(rng);
                pa.rec[path].rng = make_uint2((uint32_t)rng.state, (uint32_t)(rng.state >> 32));  // stage C draws Russian roulette from here
                const V3 wo_l = bsdf.shading_frame.to_local(wo);
                if (!(wo_l.z == 0.0f || !((bf & REFLTRANS_ALL) != 0u))) status = layered_sample_begin(bsdf.bxdf, wo_l, u, u2, MODE_RADIANCE, bs, k);
            }
            deposit_walk(status == WALK_CONTINUES, path, k, regularized);
            deposit_epilogue(status == WALK_DONE && usable(bs), path, bs);
            LJ_TALLY(15, status == WALK_CONTINUES);
            LJ_TALLY(16, status == WALK_DONE);
        }
        // integrator.rs:837-841 + 897-963: next-event estimation up to the BSDF's f, which becomes a job; the visibility test is K3's
        bool dep_nee = false;
        V3 j_wi = v3s(0.0f);
        Float j_pl = 0.0f;
        uint32_t j_flags = 0u;
        if (do_nee) {
            const float4 c0 = pa.ctx[path].c0, c1 = pa.ctx[path].c1, c2 = pa.ctx[path].c2;
            P3i si_pi;
            si_pi.x = iv2(c0.x, c0.w);
            si_pi.y = iv2(c0.y, c1.x);
            si_pi.z = iv2(c0.z, c1.y);
            const V3 si_n = v3(c1.z, c1.w, c2.x), ns = v3(c2.y, c2.z, c2.w);
            Wavelengths lambda;
            {
                const float4 a = pa.rec[path].lambda, b = pa.lambda_pdf[path];
                lambda.lambda[0] = a.x; lambda.lambda[1] = a.y; lambda.lambda[2] = a.z; lambda.lambda[3] = a.w;
                lambda.pdf[0] = b.x; lambda.pdf[1] = b.y; lambda.pdf[2] = b.z; lambda.pdf[3] = b.w;
            }
            LightSampleContext ctx;
            ctx.pi = si_pi; ctx.n = si_n; ctx.ns = ns;
            if (flags_is_reflective(bf) && !flags_is_transmissive(bf)) ctx.pi = p3i_exact(offset_ray_origin(si_pi, si_n, si_wo));
            else if (flags_is_transmissive(bf) && !flags_is_reflective(bf)) ctx.pi = p3i_exact(offset_ray_origin(si_pi, si_n, -si_wo));
            Float p_sel = 0.0f;
            const int li = light_sampler_sample(sv, nee_u, p_sel);
            if (li >= 0) {
                const ShmLight& light = sv.lights[li];
                LightLiSample ls;
                if (light_sample_li<TRI_ONLY, HAS_TEX || K_ENV_LIGHT>(sv, light, ctx, u_light, lambda, ls) && !(is_zero(ls.l) || ls.pdf == 0.0f)) {
                    // LayeredBxDF::f is zero when wo and wi lie on opposite sides of the shading plane (shm/bxdf.h, layered_f) or wo in it (bsdf.rs:48):
                    // nothing to evaluate, nothing to queue. (Only the z components of the local directions: Frame::to_local's third dot product.)
                    const Float wo_lz = dot(si_wo, ns), wi_lz = dot(ls.wi, ns);
                    if (wo_lz != 0.0f && (layered_bottom_transmits(kind) || wo_lz * wi_lz > 0.0f)) {
                        // the shadow ray does not depend on the BSDF: written now (K3 only reads it if the job queues the path)
                        Ray sr = spawn_ray_to_both_offset(si_pi, si_n, ls.p_light_pi, ls.p_light_n);
                        ShmRay sh;
                        sh.o[0] = sr.o.x; sh.o[1] = sr.o.y; sh.o[2] = sr.o.z;
                        sh.d[0] = sr.d.x; sh.d[1] = sr.d.y; sh.d[2] = sr.d.z;
                        sh.t_max = 1.0f - 0.0001f;  // 1 - SHADOW_EPSILON, integrator.rs:66,115
                        sh.pad = 0.0f;
                        pa.shadow_ray[path] = sh;
                        pa.shadow_contrib[path] = st_spec(ls.l);  // parked for the job
                        dep_nee = true;
                        j_wi = ls.wi;
                        j_pl = p_sel * ls.pdf;
                        j_flags = (light_is_delta(light) ? 0x40000000u : 0u) | (regularized ? 0x80000000u : 0u);
                    }
                }
            }
        }
        {
            const uint32_t slot = deposit_slot(n_nee, dep_nee);
            if (dep_nee) {
                uint32_t* jw = nee + slot;
                jw[0] = path | j_flags;
                jw[1 * LJ_CAP] = __float_as_uint(si_wo.x); jw[2 * LJ_CAP] = __float_as_uint(si_wo.y); jw[3 * LJ_CAP] = __float_as_uint(si_wo.z);
                jw[4 * LJ_CAP] = __float_as_uint(j_wi.x); jw[5 * LJ_CAP] = __float_as_uint(j_wi.y); jw[6 * LJ_CAP] = __float_as_uint(j_wi.z);
                jw[7 * LJ_CAP] = __float_as_uint(j_pl);
                // the throughput before this vertex's update (stage C has not run for this path): what weighs the light's contribution
                const float4 b4 = pa.rec[path].beta;
                jw[8 * LJ_CAP] = __float_as_uint(b4.x); jw[9 * LJ_CAP] = __float_as_uint(b4.y); jw[10 * LJ_CAP] = __float_as_uint(b4.z); jw[11 * LJ_CAP] = __float_as_uint(b4.w);
            }
        }
        LJ_TALLY(17, dep_nee);
#ifdef LJ_CENSUS
        { const unsigned long long _d = __ballot(dep_nee); if (lane == 0) { if (_d == 0ull) atomicAdd(&g_lj_census[18], 1ull); if (__popcll(_d) <= 8) atomicAdd(&g_lj_census[19], 1ull); } }
#endif
        LJ_TICK1(0);
    }
    push_flush(push_next, q_next, lane);
    push_flush(push_shadow, q_shadow, lane);
}

template <bool TRI_ONLY, bool HAS_TEX>
__global__ void __launch_bounds__(SHADE2_BLOCK) K_SHADE_ATTR k_scatter_layered(SceneView sv_global, PathArrays pa, const uint32_t* __restrict__ q_cur, uint32_t* __restrict__ q_next,
                                                                              uint32_t* __restrict__ q_shadow, QueueState* qs, int cur, ShmRenderParams params,
                                                                              int shadow_parity, LdsTables lds_tables) {
    __shared__ uint4 s_tables[LDS_TABLE_BUDGET_SMALL / 16];  // the lights (next-event estimation), staged once per workgroup: what the job buffers leave of 80 KB
    __shared__ uint32_t s_jobs[(SHADE2_BLOCK / WAVE) * LJ_WAVE_WORDS];
    const SceneView sv = stage_scene_tables(sv_global, lds_tables, s_tables);
    scatter_layered_staged<TRI_ONLY, HAS_TEX>(sv, pa, q_cur, q_next, q_shadow, qs, cur, params, shadow_parity, s_jobs);
}

}  // namespace

#define WF_SCATTER_LAYERED_LAUNCH(TRI, TEX)                                                                                                          \
    do {                                                                                                                                             \
        hipLaunchKernelGGL((k_scatter_layered<TRI, TEX>), dim3(a.blocks), dim3(SHADE2_BLOCK), 0, a.stream, s->dsv, s->pa, s->d_q_scatter[CLASS_LAYERED], \
                           s->d_q_active[a.cur ^ 1], s->d_q_shadow, s->d_qs, a.cur, a.params, a.shadow_parity, s->lds_tables_small);                   \
        LAUNCH_TRY("k_scatter_layered");                                                                                                             \
    } while (0)
