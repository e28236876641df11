// k_scatter.inl — staged shading, second half of a path vertex (integrator.rs:836-892 + sample_ld :897-963), ONE KERNEL PER BxDF CLASS
// over the class queue k_vertex filled: next-event estimation (light sample, BSDF f and pdf, deferred shadow ray), BSDF::sample_f, the
// throughput update, Russian roulette and the spawned ray. A wave of k_scatter<CLASS_DIELECTRIC> only ever runs dielectric code, one of
// k_scatter<CLASS_LAYERED> only the LayeredBxDF random walks: the queues are sorted by material class by construction.
// The BSDF is rebuilt from the parameter block (PathArrays::bx — one BxRec — and ctx — one CtxRec) exactly as get_bsdf left it.
#pragma once
#include "wavefront.h"

// K_ENV_LIGHT (a translation unit's switch, round 5): compile ImageInfinitelight's look-up / sample / pdf into a kernel of the class WITHOUT textures — a scene whose only
// image is an environment map needs nothing else of the textured class (k_shade_lean_env.hip says why); the *_env.hip units define it
#ifndef K_ENV_LIGHT
#define K_ENV_LIGHT false
#endif
#if K_ENV_LIGHT  // (the *_env.hip units' kernels carry their own names: a kernel trace tells them from the units without the light — tools/kernel_coverage.py)
#define k_scatter k_scatter_env
#define k_scatter_specular k_scatter_specular_env
#define k_scatter_nonspecular k_scatter_nonspecular_env
#endif
namespace {

// Deferred next-event estimation of the LayeredBxDF class. In a CoatedDiffuse / CoatedConductor vertex the expensive part of NEE is LayeredBxDF::f
// and ::pdf (two random walks through the coating) — and only the lanes whose light sample is usable run them: a third of a wave on the coated S3
// (the others wait for them). NEE feeds nothing else of the vertex: its result is the deferred shadow ray's contribution. So a lane with a usable light
// sample only DEPOSITS a job in its wave's LDS buffer — {path, wo, wi, L, p_l, beta, flags}; the BxDF is re-read from the parameter block — and goes on
// with sample_f; when 64 jobs have gathered (and at the end of the chunk) the wave evaluates them with every lane busy. Same arithmetic per path, same
// sampler dimensions: films are bit-identical. Wave-private buffer, wave-uniform count: no barrier, no atomics.
// (timing experiments only — never set in the product build: bit 0 drops the deferred NEE jobs, 1 the pdf after sample_f, 2 light sampling, 3 sample_f; bit 4 evaluates NEE inline)
#ifndef SHM_EXP_SKIP
#define SHM_EXP_SKIP 0
#endif
#ifndef K_SCATTER_NEE_LAST
#define K_SCATTER_NEE_LAST 1  // (0: next-event estimation evaluated where the reference's text has it: A/B)
#endif
constexpr int NEE_JOB_WORDS = 17;
constexpr int NEE_JOB_CAP = 2 * WAVE;  // at most 63 waiting + 64 new

// SUB (the dielectric class only): 0 = every entry of the class queue; 1 = only the entries whose BxDF is specular — a ThinDielectricBxDF, or a DielectricBxDF whose
// distribution is effectively smooth — with next-event estimation and the rough-interface code compiled OUT (a specular BSDF has no NEE, integrator.rs:837):
// the kernel then needs a fraction of the registers and runs four waves per SIMD instead of two; 2 = only the others. Kernels 1 and 2 walk the same queue and
// each skips the other's entries (decided from the parameter block alone, so only without options.force_diffuse / regularize, which change the BxDF here).
template <int CLASS, bool TRI_ONLY, bool HAS_TEX, int SUB = 0>
__device__ __forceinline__ void scatter_body(const SceneView& sv, const PathArrays& pa, const uint32_t* __restrict__ q_cur, uint32_t* __restrict__ q_next,
                                             uint32_t* __restrict__ q_shadow, QueueState* qs, int cur, const ShmRenderParams& params, int shadow_parity) {
    const uint32_t n = qs->n_scatter[CLASS];
    constexpr bool LAYERED = CLASS == CLASS_LAYERED;
    constexpr uint32_t ROUNDS = SHADE_CHUNK / SHADE2_BLOCK;
    __shared__ uint32_t s_next[SHADE_CHUNK], s_shadow[SHADE_CHUNK];
    __shared__ uint32_t s_cnt[2], s_base[2];
    __shared__ uint32_t s_jobs[LAYERED ? (SHADE2_BLOCK / WAVE) * NEE_JOB_WORDS * NEE_JOB_CAP : 1];
    uint32_t* const jobs = s_jobs + (LAYERED ? (threadIdx.x / WAVE) * (NEE_JOB_WORDS * NEE_JOB_CAP) : 0);
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    // options.force_diffuse replaces the BxDF by a DiffuseBxDF (its f is a constant): nothing worth deferring, NEE stays inline
    const bool defer = LAYERED && params.force_diffuse == 0 && !(SHM_EXP_SKIP & 16);
    uint32_t n_jobs = 0;  // wave-uniform
    // the BxDF and the shading frame as k_vertex left them in the parameter block (c2 = PathArrays::ctx[path].c2: n.z and the shading normal)
    auto build_bsdf = [&](uint32_t path, const float4& c2, BSDF& bsdf, V3& ns) {
        BxDF& b = bsdf.bxdf;
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
        b.k = (CLASS == CLASS_CONDUCTOR || CLASS == CLASS_LAYERED) ? ld_spec(pa.bx[path].bx1) : spec_const(0.0f);
        if (CLASS == CLASS_LAYERED) {
            b.albedo = ld_spec(pa.bx[path].bx3);
            const float4 p4 = pa.bx[path].bx4;
            b.mf2.alpha_x = p4.x; b.mf2.alpha_y = p4.y; b.thickness = p4.z; b.g = p4.w;
        } else {
            b.albedo = spec_const(0.0f);
            b.mf2.alpha_x = 0.0f; b.mf2.alpha_y = 0.0f; b.thickness = 0.0f; b.g = 0.0f;
        }
        // the class is a property of the queue: the dispatch inside bxdf_f / bxdf_pdf / bxdf_sample_f folds to this class's code
        // (plus DiffuseBxDF, which options.force_diffuse substitutes below)
        if (CLASS == CLASS_DIFFUSE) __builtin_assume(bsdf.bxdf.kind == SHM_MATERIAL_DIFFUSE);
        if (CLASS == CLASS_CONDUCTOR) __builtin_assume(bsdf.bxdf.kind == SHM_MATERIAL_CONDUCTOR);
        if (CLASS == CLASS_DIELECTRIC) __builtin_assume(bsdf.bxdf.kind == SHM_MATERIAL_DIELECTRIC || bsdf.bxdf.kind == SHM_MATERIAL_THIN_DIELECTRIC);
        if (CLASS == CLASS_LAYERED) __builtin_assume(bsdf.bxdf.kind == SHM_MATERIAL_COATED_DIFFUSE || bsdf.bxdf.kind == SHM_MATERIAL_COATED_CONDUCTOR);
        const float4 f = pa.bx[path].fr;
        ns = v3(c2.y, c2.z, c2.w);
        // Frame::from_xz (frame.rs:14-17): y = z cross x
        bsdf.shading_frame.x = v3(f.x, f.y, f.z);
        bsdf.shading_frame.z = ns;
        bsdf.shading_frame.y = cross(ns, bsdf.shading_frame.x);
    };
    for (uint32_t chunk0 = blockIdx.x * SHADE_CHUNK; chunk0 < n; chunk0 += gridDim.x * SHADE_CHUNK) {
      if (threadIdx.x == 0) { s_cnt[0] = 0; s_cnt[1] = 0; }
      __syncthreads();
      for (uint32_t k = 0; k < ROUNDS || (LAYERED && n_jobs > 0u);) {
        if (LAYERED && (n_jobs >= (uint32_t)WAVE || k == ROUNDS)) {
            // ---- a full wave of deferred NEE jobs (the newest ones; at the end of the chunk whatever is left) ----
            const uint32_t take = n_jobs < (uint32_t)WAVE ? n_jobs : (uint32_t)WAVE;
            const uint32_t first = n_jobs - take;
            n_jobs = first;
            bool push_shadow = false;
            uint32_t path = 0;
            if (lane < take && !(SHM_EXP_SKIP & 1)) {
                const uint32_t* jw = jobs + first + lane;
                path = jw[0];
                BSDF bsdf;
                V3 ns;
                build_bsdf(path, pa.ctx[path].c2, bsdf, ns);
                const uint32_t jf = jw[16 * NEE_JOB_CAP];
                if (jf & 2u) bxdf_regularize(bsdf.bxdf);
                const V3 si_wo = v3(__uint_as_float(jw[1 * NEE_JOB_CAP]), __uint_as_float(jw[2 * NEE_JOB_CAP]), __uint_as_float(jw[3 * NEE_JOB_CAP]));
                const V3 wi = v3(__uint_as_float(jw[4 * NEE_JOB_CAP]), __uint_as_float(jw[5 * NEE_JOB_CAP]), __uint_as_float(jw[6 * NEE_JOB_CAP]));
                // integrator.rs:924-962, from the BSDF's f on
                Spec f = bsdf_f(bsdf, si_wo, wi) * abs_dot(wi, ns);
                if (!is_zero(f)) {
                    Spec l, beta;
                    for (int c = 0; c < 4; ++c) { l.v[c] = __uint_as_float(jw[(7 + c) * NEE_JOB_CAP]); beta.v[c] = __uint_as_float(jw[(12 + c) * NEE_JOB_CAP]); }
                    const Float p_l = __uint_as_float(jw[11 * NEE_JOB_CAP]);
                    Spec ld;
                    if (jf & 1u) {
                        ld = l * f / p_l;
                    } else {
                        Float pb2 = bsdf_pdf(bsdf, si_wo, wi, REFLTRANS_ALL);
                        Float w_l = power_heuristic(1, p_l, 1, pb2);
                        ld = w_l * l * f / p_l;
                    }
                    pa.shadow_contrib[path] = st_spec(beta * ld);
                    push_shadow = true;
                }
            }
            uint32_t s2 = queue_push_slot(&s_cnt[1], push_shadow);
            if (push_shadow) s_shadow[s2] = path;
            continue;
        }
        const uint32_t i = chunk0 + k * SHADE2_BLOCK + threadIdx.x;
        ++k;
        bool push_next = false, push_shadow = false;
        uint32_t path = 0;
        // a deferred NEE job of this lane (LAYERED): deposited below, where the wave has reconverged
        bool dep = false;
        V3 j_wo = v3s(0.0f), j_wi = v3s(0.0f);
        Spec j_l = spec_const(0.0f), j_beta = spec_const(0.0f);
        Float j_pl = 0.0f;
        uint32_t j_flags = 0u;
        bool mine = i < n;
        if (mine) {
            path = q_cur[i];
            if (SUB != 0) {
                const float4 p2 = pa.bx[path].bx2;
                // (what BxDF::flags calls SPECULAR, bxdf.rs:541-551 / 812-814: an index-matched interface with a rough distribution is GLOSSY — its NEE finds f = 0,
                //  but draws its three sampler dimensions — and stays with the general kernel)
                const bool specular = (__float_as_uint(p2.w) & 0xffu) == SHM_MATERIAL_THIN_DIELECTRIC || (p2.y < 1e-3f && p2.z < 1e-3f);
                mine = specular == (SUB == 1);
            }
        }
        if (mine) {
            // ---- the vertex as k_vertex left it ----
            BSDF bsdf;
            P3i si_pi;
            V3 si_n, ns;
            {
                const float4 c0 = pa.ctx[path].c0, c1 = pa.ctx[path].c1, c2 = pa.ctx[path].c2;
                si_pi.x = iv2(c0.x, c0.w);
                si_pi.y = iv2(c0.y, c1.x);
                si_pi.z = iv2(c0.z, c1.y);
                si_n = v3(c1.z, c1.w, c2.x);
                build_bsdf(path, c2, bsdf, ns);
                // (SUB == 1: the specular branches never read the roughness; as constants they fold TrowbridgeReitz::effectively_smooth and with it the rough code away)
                if (SUB == 1) { bsdf.bxdf.mf.alpha_x = 0.0f; bsdf.bxdf.mf.alpha_y = 0.0f; }
            }
            const float4* rp = reinterpret_cast<const float4*>(pa.ray + path);
            const float4 r0 = rp[0], r1 = rp[1];
            const V3 wo = -v3(r0.w, r1.x, r1.y);  // li()'s wo = -ray.d (integrator.rs:844)
            // intr.wo, what sample_ld and get_bsdf use: bitwise -ray.d for a top-level triangle; a quadric or an instanced primitive
            // builds its interaction in object space and maps it back (sphere.rs:254-270, primitive.rs:165-170), so it is carried
            V3 si_wo = wo;
            if (!TRI_ONLY) { const float4 w4 = pa.bx[path].siwo; si_wo = v3(w4.x, w4.y, w4.z); }
            Spec beta = ld_spec(pa.rec[path].beta);
            Wavelengths lambda;
            {
                const float4 a = pa.rec[path].lambda, b = pa.lambda_pdf[path];
                lambda.lambda[0] = a.x; lambda.lambda[1] = a.y; lambda.lambda[2] = a.z; lambda.lambda[3] = a.w;
                lambda.pdf[0] = b.x; lambda.pdf[1] = b.y; lambda.pdf[2] = b.z; lambda.pdf[3] = b.w;
            }
            const uint32_t fl = pa.rec[path].flags;
            int depth = (int)(fl & 0xffu);
            bool specular_bounce = (fl >> 8) & 1u;
            bool any_non_specular_bounces = (fl >> 9) & 1u;
            Float p_b, eta_scale = pa.rec[path].pb_eta.y;
            Rng rng;
            {
                const uint32_t pix = pa.rec[path].pixel;
                const uint2 rs = pa.rec[path].rng;
                rng.state = (uint64_t)rs.x | ((uint64_t)rs.y << 32);
                // inc is a pure function of (pixel, seed): re-derived instead of stored
                uint64_t h = mix_bits(((uint64_t)(pix & 0xffffu) << 32) | (uint64_t)(pix >> 16));
                h = mix_bits(h ^ (params.seed + 0x9e3779b97f4a7c15ULL));
                rng.inc = (h << 1u) | 1u;
            }
            // options.force_diffuse (interaction.rs:256-275) draws inside get_bsdf, i.e. before anything else of this half
            if (SUB == 0 && params.force_diffuse) {  // (the SUB kernels are only launched without it, and without regularize)
                Float uc = sampler_get_1d(rng);
                V2 u2f = sampler_get_2d(rng);
                bsdf_force_diffuse(bsdf, si_wo, uc, u2f);
            }
            if (SUB == 0 && params.regularize && any_non_specular_bounces) bxdf_regularize(bsdf.bxdf);
            bool alive = true;
            depth += 1;
            // integrator.rs:837-841 + 897-963: next-event estimation; the visibility test is deferred to K3. The sampler dimensions are drawn here, in the reference's
            // order; for every class but the layered one the evaluation runs as the LAST thing of the vertex (as in the fused kernel, k_shade.inl: the light sample is
            // the register-hungriest island, and at the end only its own inputs are live beside it). The layered class keeps it first: its sample_f walk is the
            // hungrier part, and NEE leaves only a deferred job behind.
            const uint32_t bf = bsdf_flags(bsdf);
            const bool do_nee = SUB != 1 && flags_is_non_specular(bf) && !(SHM_EXP_SKIP & 4);
            Float nee_u = 0.0f;
            V2 nee_u_light = v2(0.0f, 0.0f);
            if (do_nee) {
                nee_u = sampler_get_1d(rng);
                nee_u_light = sampler_get_2d(rng);
            }
            const Spec beta_at_vertex = beta;  // (the throughput before this vertex's update: what weighs the light's contribution)
            auto next_event_estimation = [&]() {
                LightSampleContext ctx;
                ctx.pi = si_pi; ctx.n = si_n; ctx.ns = ns;
                if (flags_is_reflective(bf) && !flags_is_transmissive(bf)) ctx.pi = p3i_exact(offset_ray_origin(si_pi, si_n, si_wo));
                else if (flags_is_transmissive(bf) && !flags_is_reflective(bf)) ctx.pi = p3i_exact(offset_ray_origin(si_pi, si_n, -si_wo));
                Float p_sel = 0.0f;
                int li = light_sampler_sample(sv, nee_u, p_sel);
                const V2 u_light = nee_u_light;
                if (li >= 0) {
                    const ShmLight& light = sv.lights[li];
                    LightLiSample ls;
                    if (light_sample_li<TRI_ONLY, HAS_TEX || K_ENV_LIGHT>(sv, light, ctx, u_light, lambda, ls) && !(is_zero(ls.l) || ls.pdf == 0.0f)) {
                        V3 wi = ls.wi;
                        // (LayeredBxDF::f is zero when wo and wi lie on opposite sides of the shading plane — shm/bxdf.h, layered_f — or wo in it,
                        //  bsdf.rs:48: half of the usable light samples on the coated S3; nothing to evaluate, nothing to queue)
                        const bool f_may_be_nonzero = !LAYERED || [&] {
                            const V3 wo_l = bsdf.shading_frame.to_local(si_wo), wi_l = bsdf.shading_frame.to_local(wi);
                            return wo_l.z != 0.0f && (layered_bottom_transmits(bsdf.bxdf.kind) || same_hemisphere(wo_l, wi_l));
                        }();
                        if (defer && !f_may_be_nonzero) {
                        } else if (defer) {
                            // the shadow ray does not depend on the BSDF: written now (K3 only reads it if the job queues the path)
                            Ray sr = spawn_ray_to_both_offset(si_pi, si_n, ls.p_light_pi, ls.p_light_n);
                            ShmRay s;
                            s.o[0] = sr.o.x; s.o[1] = sr.o.y; s.o[2] = sr.o.z;
                            s.d[0] = sr.d.x; s.d[1] = sr.d.y; s.d[2] = sr.d.z;
                            s.t_max = 1.0f - 0.0001f;  // 1 - SHADOW_EPSILON, integrator.rs:66,115
                            s.pad = 0.0f;
                            pa.shadow_ray[path] = s;
                            dep = true;
                            j_wo = si_wo; j_wi = wi; j_l = ls.l; j_beta = beta_at_vertex;
                            j_pl = p_sel * ls.pdf;
                            j_flags = (light_is_delta(light) ? 1u : 0u) | ((params.regularize && any_non_specular_bounces) ? 2u : 0u);
                        } else {
                        if (LAYERED && !(SHM_EXP_SKIP & 16)) __builtin_assume(bsdf.bxdf.kind == SHM_MATERIAL_DIFFUSE);  // (force_diffuse: see `defer`)
                        Spec f = bsdf_f(bsdf, si_wo, wi) * abs_dot(wi, ns);
                        if (!is_zero(f)) {
                            Ray sr = spawn_ray_to_both_offset(si_pi, si_n, ls.p_light_pi, ls.p_light_n);
                            Float p_l = p_sel * ls.pdf;
                            Spec ld;
                            if (light_is_delta(light)) {
                                ld = ls.l * f / p_l;
                            } else {
                                Float pb2 = bsdf_pdf(bsdf, si_wo, wi, REFLTRANS_ALL);
                                Float w_l = power_heuristic(1, p_l, 1, pb2);
                                ld = w_l * ls.l * f / p_l;
                            }
                            ShmRay s;
                            s.o[0] = sr.o.x; s.o[1] = sr.o.y; s.o[2] = sr.o.z;
                            s.d[0] = sr.d.x; s.d[1] = sr.d.y; s.d[2] = sr.d.z;
                            s.t_max = 1.0f - 0.0001f;  // 1 - SHADOW_EPSILON, integrator.rs:66,115
                            s.pad = 0.0f;
                            pa.shadow_ray[path] = s;
                            pa.shadow_contrib[path] = st_spec(beta_at_vertex * ld);
                            push_shadow = true;
                        }
                        }
                    }
                }
            };
            constexpr bool NEE_LAST = !LAYERED && K_SCATTER_NEE_LAST;
            if (!NEE_LAST && do_nee) next_event_estimation();
            // integrator.rs:843-857: sample the BSDF
            Float u = sampler_get_1d(rng);
            V2 u2 = sampler_get_2d(rng);
            BSDFSample bs;
            if ((SHM_EXP_SKIP & 8) || !bsdf_sample_f(bsdf, wo, u, u2, REFLTRANS_ALL, bs)) {
                alive = false;
            } else {
                // integrator.rs:859-872
                beta = beta * (bs.f * abs_dot(bs.wi, ns) / bs.pdf);
                p_b = (bs.pdf_is_proportional && !(SHM_EXP_SKIP & 2)) ? bsdf_pdf(bsdf, wo, bs.wi, REFLTRANS_ALL) : bs.pdf;
                specular_bounce = flags_is_specular(bs.flags);
                any_non_specular_bounces |= !specular_bounce;
                if (flags_is_transmissive(bs.flags)) eta_scale *= sqr(bs.eta);
                V3 no = offset_ray_origin(si_pi, si_n, bs.wi);  // integrator.rs:875 -> interaction.rs:68-75
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
                    pa.rec[path].beta = st_spec(beta);
                    pa.rec[path].pb_eta = make_float2(p_b, eta_scale);
                    // (the CtxRec already holds this vertex's context: the next vertex's prev_intr_ctx)
                    pa.rec[path].rng = make_uint2((uint32_t)rng.state, (uint32_t)(rng.state >> 32));
                    uint32_t aux_bit = 0u;
                    if (HAS_TEX && CLASS != CLASS_DIFFUSE && (fl & (1u << 10)) &&
                        (bs.flags == BXDF_SPECULAR_REFLECTION || bs.flags == BXDF_SPECULAR_TRANSMISSION)) {
                        // spawn_ray_with_differentials, interaction.rs:430-514 (only specular bounces carry differentials on)
                        const float4 d0 = pa.dd0[path], d1 = pa.dd1[path], d2 = pa.dd2[path];
                        AuxRays na = spawn_ray_differentials_pre(si_pi.mid(), si_wo, ns, v3(d0.x, d0.y, d0.z), v3(d0.w, d1.x, d1.y), v3(d1.z, d1.w, d2.x),
                                                                 v3(d2.y, d2.z, d2.w), ld_aux(pa, path), bs.wi, bs.flags, bs.eta);
                        if (na.has) { st_aux(pa, path, na); aux_bit = 1u << 10; }
                    }
                    pa.rec[path].flags = (uint32_t)depth | ((uint32_t)specular_bounce << 8) | ((uint32_t)any_non_specular_bounces << 9) | aux_bit;
                    push_next = true;
                }
            }
            if (NEE_LAST && do_nee) next_event_estimation();
        }
        if (LAYERED) {
            const unsigned long long m = __ballot(dep);
            if (m != 0ull) {
                if (dep) {
                    uint32_t* jw = jobs + n_jobs + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    jw[0] = path;
                    jw[1 * NEE_JOB_CAP] = __float_as_uint(j_wo.x); jw[2 * NEE_JOB_CAP] = __float_as_uint(j_wo.y); jw[3 * NEE_JOB_CAP] = __float_as_uint(j_wo.z);
                    jw[4 * NEE_JOB_CAP] = __float_as_uint(j_wi.x); jw[5 * NEE_JOB_CAP] = __float_as_uint(j_wi.y); jw[6 * NEE_JOB_CAP] = __float_as_uint(j_wi.z);
                    for (int c = 0; c < 4; ++c) { jw[(7 + c) * NEE_JOB_CAP] = __float_as_uint(j_l.v[c]); jw[(12 + c) * NEE_JOB_CAP] = __float_as_uint(j_beta.v[c]); }
                    jw[11 * NEE_JOB_CAP] = __float_as_uint(j_pl);
                    jw[16 * NEE_JOB_CAP] = j_flags;
                }
                n_jobs = (uint32_t)__builtin_amdgcn_readfirstlane((int)(n_jobs + (uint32_t)__popcll(m)));
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
}

// two waves per SIMD (<= 256 VGPRs): every class fits without spilling except the LayeredBxDF walks (586 spilled VGPRs). Measured on the
// coated S3 (1024^2 x 64 spp): two waves with the spills 85.6 ms of shading per frame, one wave per SIMD without them (the retired k_scatter_w1)
// 99.2 ms — latency hiding beats the scratch traffic.
template <int CLASS, bool TRI_ONLY, bool HAS_TEX>
__global__ void __launch_bounds__(SHADE2_BLOCK) K_SHADE_ATTR k_scatter(SceneView sv_global, PathArrays pa, const uint32_t* __restrict__ q_cur, uint32_t* __restrict__ q_next,
                                                                      uint32_t* __restrict__ q_shadow, QueueState* qs, int cur, ShmRenderParams params,
                                                                      int shadow_parity, LdsTables lds_tables) {
    __shared__ uint4 s_tables[LDS_TABLE_BUDGET / 16];  // the small scene tables (lights, spectra: next-event estimation), staged once per workgroup
    const SceneView sv = stage_scene_tables(sv_global, lds_tables, s_tables);
    scatter_body<CLASS, TRI_ONLY, HAS_TEX>(sv, pa, q_cur, q_next, q_shadow, qs, cur, params, shadow_parity);
}
// the specular half of the dielectric class at four waves per SIMD (SUB = 1), and the rest of the class (SUB = 2) as before
template <int CLASS, bool TRI_ONLY, bool HAS_TEX>
__global__ void __launch_bounds__(SHADE2_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4))) k_scatter_specular(SceneView sv_global, PathArrays pa, const uint32_t* __restrict__ q_cur,
                                                                                                             uint32_t* __restrict__ q_next, uint32_t* __restrict__ q_shadow,
                                                                                                             QueueState* qs, int cur, ShmRenderParams params, int shadow_parity, LdsTables lds_tables) {
    __shared__ uint4 s_tables[LDS_TABLE_BUDGET / 16];  // the small scene tables (lights, spectra: next-event estimation), staged once per workgroup
    const SceneView sv = stage_scene_tables(sv_global, lds_tables, s_tables);
    scatter_body<CLASS, TRI_ONLY, HAS_TEX, 1>(sv, pa, q_cur, q_next, q_shadow, qs, cur, params, shadow_parity);
}
template <int CLASS, bool TRI_ONLY, bool HAS_TEX>
__global__ void __launch_bounds__(SHADE2_BLOCK) K_SHADE_ATTR k_scatter_nonspecular(SceneView sv_global, PathArrays pa, const uint32_t* __restrict__ q_cur, uint32_t* __restrict__ q_next,
                                                                                  uint32_t* __restrict__ q_shadow, QueueState* qs, int cur, ShmRenderParams params,
                                                                                  int shadow_parity, LdsTables lds_tables) {
    __shared__ uint4 s_tables[LDS_TABLE_BUDGET / 16];  // the small scene tables (lights, spectra: next-event estimation), staged once per workgroup
    const SceneView sv = stage_scene_tables(sv_global, lds_tables, s_tables);
    scatter_body<CLASS, TRI_ONLY, HAS_TEX, 2>(sv, pa, q_cur, q_next, q_shadow, qs, cur, params, shadow_parity);
}
}  // namespace

#define WF_SCATTER_LAUNCH(CLASS, TRI, TEX)                                                                                                      \
    do {                                                                                                                                        \
        hipLaunchKernelGGL((k_scatter<CLASS, TRI, TEX>), dim3(a.blocks), dim3(SHADE2_BLOCK), 0, a.stream, s->dsv, s->pa, s->d_q_scatter[CLASS], \
                           s->d_q_active[a.cur ^ 1], s->d_q_shadow, s->d_qs, a.cur, a.params, a.shadow_parity, s->lds_tables);                                \
        LAUNCH_TRY("k_scatter");                                                                                                                \
    } while (0)
#define WF_SCATTER_LAUNCH_SUB(KERNEL, BLOCKS, CLASS, TRI, TEX)                                                                                   \
    do {                                                                                                                                        \
        hipLaunchKernelGGL((KERNEL<CLASS, TRI, TEX>), dim3(BLOCKS), dim3(SHADE2_BLOCK), 0, a.stream, s->dsv, s->pa, s->d_q_scatter[CLASS],         \
                           s->d_q_active[a.cur ^ 1], s->d_q_shadow, s->d_qs, a.cur, a.params, a.shadow_parity, s->lds_tables);                                \
        LAUNCH_TRY(#KERNEL);                                                                                                                    \
    } while (0)
// the three scene classes every BxDF class is instantiated for
#define WF_SCATTER_DISPATCH(CLASS)                                                  \
    do {                                                                            \
        if (has_tex) WF_SCATTER_LAUNCH(CLASS, false, true);                         \
        else if (tri_only) WF_SCATTER_LAUNCH(CLASS, true, false);                   \
        else WF_SCATTER_LAUNCH(CLASS, false, false);                                \
    } while (0)
