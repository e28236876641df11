// k_shade.inl — the fused K4+K5 kernel template (one whole path vertex of PathIntegrator::li per launch), instantiated by
// k_shade_lean.hip for the all-diffuse triangle scene class (every other class runs the staged k_vertex -> k_scatter<class>).
#pragma once
#include "wavefront.h"

namespace {

// Three waves per SIMD (<= 168 VGPRs) for the fused kernel: it waits for its gathers more than half of its cycles. Path state that is only needed at
// one place (beta, p_b, eta_scale) is loaded THERE instead of being held across the vertex, which brings the kernel from 205 to 191 VGPRs at two waves
// and to 19 spilled ones at three: shade 128.7 -> 121.6 ms per headline frame (three waves with the early loads: 45 spilled, 130.8 ms; two waves with
// the late loads: 133.1 ms — the extra loads are exposed latency there). The grid is 3 workgroups per CU (6 is the same; 4 leaves a quarter of the
// chunks to a second round: 154.7 ms).
#ifndef K_SHADE_LEAN_WAVES
#define K_SHADE_LEAN_WAVES 3
#endif
#define K_SHADE_LEAN_ATTR __attribute__((amdgpu_waves_per_eu(K_SHADE_LEAN_WAVES, K_SHADE_LEAN_WAVES)))
// timing / register experiments only (tools/exp_variants.sh k_shade_lean SHM_SHADE_SKIP <bits>; results are wrong by design): 1 = no next-event estimation,
// 4 = no BSDF sampling (paths end), 16 = no bump_map / get_bsdf beyond the reflectance
#ifndef SHM_SHADE_SKIP
#define SHM_SHADE_SKIP 0
#endif
#ifndef K_SHADE_NEE_LAST
#define K_SHADE_NEE_LAST 1  // next-event estimation evaluated at the end of the vertex (0: where the reference's text has it: A/B)
#endif

// ---------------------------------------------------------------------------------------------
// K4+K5: one path vertex (integrator.rs:772-892 for the vertex found by K2).
// ---------------------------------------------------------------------------------------------
// HAS_LAYERED = false is the instantiation for scenes without Coated* materials: the LayeredBxDF random walks (three per
// vertex: f and pdf for NEE, sample_f) are compiled out of it.
//   TRI_ONLY = true is the instantiation for scenes made of triangles only: no quadric / bilinear-patch interaction and light
//   sampling code (with it the kernel needs 264 VGPRs, one wave per SIMD; without it 243, two waves).
//   ENV_LIGHT = true (round 5; with HAS_TEX = false): the scene's only image is an ImageInfinitelight — its look-up, sample and pdf (light.rs:805-981) are compiled in, and
//   nothing else of the textured class: no ray differentials, no auxiliary rays (an image light reads the map at a direction; differentials only feed texture filtering).
//   HAS_TEX = true is the instantiation for scenes that bind image textures: the path carries ray differentials (texture.h),
//   get_bsdf filters the MIP pyramids. One general instantiation <true, false, true>.
//   DIFFUSE_ONLY = true is the instantiation for scenes whose materials are all DiffuseMaterial (the headline scene class): the
//   conductor / dielectric BxDFs and the material dispatch are compiled out of it.
//   EMIT_INLINE = false (the lean instantiation): emission at the hit (integrator.rs:798-813) is not evaluated here — a vertex that hit an emitter deposits what
//   the evaluation needs of the state this kernel is about to overwrite (PathArrays::e_*) and its path in q_emit; k_emit_jobs below works the list off after the launch.
//   SORT_CHUNK = true (the tail instantiation: every material class in one kernel, late bounces of deep renders): the workgroup counting-sorts its 2048-entry chunk by the
//   material of the primitive each path hit (as k_vertex does, k_vertex.inl) before it works through it, so that the 64 lanes of a wave mostly run ONE material's BxDF
//   — the queue is in image order, and a wave of the unsorted kernel ran at 11.5 of 64 lanes per instruction on C4's late bounces (profiles/r04_staged_C4.txt).
//   Paths are independent and every later queue is order-agnostic: films and counters do not change.
constexpr int SHADE_SORT_BINS = 64;
template <bool HAS_LAYERED, bool TRI_ONLY, bool HAS_TEX = false, bool DIFFUSE_ONLY = false, bool EMIT_INLINE = true, bool SORT_CHUNK = false, bool ENV_LIGHT = false>
__global__ void __launch_bounds__(SHADE2_BLOCK) K_SHADE_LEAN_ATTR k_shade(SceneView sv_global, PathArrays pa, const uint32_t* __restrict__ q_cur, uint32_t* __restrict__ q_next,
                                                     uint32_t* __restrict__ q_shadow, QueueState* qs, int cur, ShmRenderParams params,
                                                     DeviceCounters* counters, int shadow_parity, const uint32_t* __restrict__ n_in, uint32_t* __restrict__ q_emit,
                                                     LdsTables lds_tables, int bounce_flags) {
    // bit 0: bounce 0 on known constants (ShadeArgs::first_bounce); bit 1: the scene runs this kernel alone (all-diffuse triangles, no diversion), and what the vertex leaves
    // for the MIS weight of an emitter hit at the NEXT vertex is its hit record — primitive + barycentrics, 16 bytes — instead of its LightSampleContext (48): hitting an
    // emitter is rare, and k_emit_jobs rebuilds the context from the record with the very code that built it here (triangle_interaction depends on nothing else)
    // bit 2 (with bit 1): the hit array is double-buffered by bounce parity (16-byte records, the two halves of the ShmHit allocation): the previous vertex's record is still
    // there (PathArrays::hit_prev) and nothing at all is written for the next vertex
    const bool first_bounce = (bounce_flags & 1) != 0, ctx_as_hit = (bounce_flags & 2) != 0, hit_kept = (bounce_flags & 4) != 0;
    __shared__ uint4 s_tables[LDS_TABLE_BUDGET / 16];  // the small scene tables, staged once per workgroup (wavefront.h, stage_scene_tables)
    __shared__ uint4 s_view[HAS_TEX ? SCENE_VIEW_UINT4S : 1];  // with textures: the view itself, for the texture evaluators that are real calls (shm/texture.h)
    const SceneView sv = stage_scene_tables_tex<HAS_TEX>(sv_global, lds_tables, s_tables, s_view);
    const uint32_t n = n_in ? *n_in : qs->n_active[cur];  // (n_in: the lean diversion's queue, whose count is not n_active)
    __shared__ uint32_t s_next[SHADE_CHUNK], s_shadow[SHADE_CHUNK];
    __shared__ uint32_t s_emit[EMIT_INLINE ? 1 : SHADE_CHUNK];
    __shared__ uint32_t s_cnt[3], s_base[3];
    __shared__ uint32_t s_sorted[SORT_CHUNK ? SHADE_CHUNK : 1];
    __shared__ uint32_t s_bin[SORT_CHUNK ? SHADE_SORT_BINS + 1 : 1];
    for (uint32_t chunk0 = blockIdx.x * SHADE_CHUNK; chunk0 < n; chunk0 += gridDim.x * SHADE_CHUNK) {
      if (threadIdx.x == 0) { s_cnt[0] = 0; s_cnt[1] = 0; s_cnt[2] = 0; }
      if (SORT_CHUNK) {
          // counting sort of the chunk's paths by the material of the primitive they hit (64 bins + one for escaped rays): each entry's path and key stay in registers
          // between the counting and the scatter pass, so the q -> hit -> primitive -> material chain of dependent gathers is walked once
          if (threadIdx.x <= SHADE_SORT_BINS) s_bin[threadIdx.x] = 0;
          __syncthreads();
          uint32_t my_path[SHADE_CHUNK / SHADE2_BLOCK];
          // (the packing below holds exactly eight 8-bit keys per lane in two words: both constants are load-bearing)
          static_assert(SHADE_CHUNK / SHADE2_BLOCK == 8 && SHADE_SORT_BINS < 256, "the chunk sort packs 8 keys of 8 bits per lane");
          uint32_t my_keys = 0u, my_keys_hi = 0u;  // 8 keys of 7 bits (0..64)
#pragma unroll
          for (uint32_t k = 0; k < SHADE_CHUNK / SHADE2_BLOCK; ++k) {
              const uint32_t i = chunk0 + k * SHADE2_BLOCK + threadIdx.x;
              my_path[k] = 0u;
              if (i < n) {
                  my_path[k] = first_bounce ? i : q_cur[i];  // (bounce 0 on known constants: the identity queue was not written)
                  const int prim = pa.hit16 ? hit_prim_of(__float_as_int(reinterpret_cast<const float*>(reinterpret_cast<const float4*>(pa.hit) + my_path[k])[0]))
                                            : __float_as_int(reinterpret_cast<const float*>(pa.hit + my_path[k])[0]);
                  uint32_t key = (uint32_t)SHADE_SORT_BINS;
                  if (prim >= 0) { const uint32_t m = sv.prim_recs[prim].material; key = m < (uint32_t)SHADE_SORT_BINS ? m : (uint32_t)SHADE_SORT_BINS - 1u; }
                  if (k < 4) my_keys |= key << (8u * k); else my_keys_hi |= key << (8u * (k - 4u));
                  atomicAdd(&s_bin[key], 1u);
              }
          }
          __syncthreads();
          if (threadIdx.x == 0) {  // exclusive prefix over 65 bins: the bins become cursors
              uint32_t acc = 0;
              for (int b = 0; b <= SHADE_SORT_BINS; ++b) { const uint32_t c = s_bin[b]; s_bin[b] = acc; acc += c; }
          }
          __syncthreads();
#pragma unroll
          for (uint32_t k = 0; k < SHADE_CHUNK / SHADE2_BLOCK; ++k) {
              const uint32_t i = chunk0 + k * SHADE2_BLOCK + threadIdx.x;
              const uint32_t key = ((k < 4 ? my_keys >> (8u * k) : my_keys_hi >> (8u * (k - 4u)))) & 0xffu;
              if (i < n) s_sorted[atomicAdd(&s_bin[key], 1u)] = my_path[k];
          }
      }
      __syncthreads();
      for (uint32_t k = 0; k < SHADE_CHUNK / SHADE2_BLOCK; ++k) {
        const uint32_t i = chunk0 + k * SHADE2_BLOCK + threadIdx.x;
        bool active = i < n;
        bool push_next = false, push_shadow = false, push_emit = false;
        uint32_t path = 0;
        if (active) {
            // first_bounce (wave-uniform; the lean class and, since round 5, every class whose bounce 0 runs this kernel): k_generate left the constants out — the queue is the identity, beta = 1, p_b = eta_scale = 1, flags = 0
            path = SORT_CHUNK ? s_sorted[k * SHADE2_BLOCK + threadIdx.x] : (first_bounce ? i : q_cur[i]);
            Hit hit;
            hit = load_hit_tri(pa, path);  // (the 32-byte ShmHit, or — triangle scenes, whatever instantiation shades them: a textured triangle scene runs the general ones — the compact form)
            const float4* rp = reinterpret_cast<const float4*>(pa.ray + path);
            float4 r0 = rp[0], r1 = rp[1];
            V3 ray_d = v3(r0.w, r1.x, r1.y);
            // L is only touched by a vertex that adds emission (most do not): loaded and stored inside add_l
            auto add_l = [&](const Spec& c) { pa.L[path] = st_spec(ld_spec(pa.L[path]) + c); };
            // (beta is read where it is used — emission, the NEE contribution, the throughput update — instead of being held across the vertex)
            auto load_beta = [&]() { return first_bounce ? spec_const(1.0f) : ld_spec(pa.rec[path].beta); };
            auto load_pb_eta = [&]() { return first_bounce ? make_float2(1.0f, 1.0f) : pa.rec[path].pb_eta; };
            Spec beta;
            Wavelengths lambda;
            float4 pdf_in, lambda4_in;
            uint32_t pix_in = 0u;
            {
                // (bounce 0 on known constants: k_generate<LEAN> leaves no record — the wavelengths are in the film's array, sampler state and pixel in rng0 / pixel0 —
                //  and this vertex writes the path's first one, whole)
                float4 a = first_bounce ? pa.lambda[path] : pa.rec[path].lambda, b = pa.lambda_pdf[path];
                lambda4_in = a;
                pdf_in = b;
                lambda.lambda[0] = a.x; lambda.lambda[1] = a.y; lambda.lambda[2] = a.z; lambda.lambda[3] = a.w;
                lambda.pdf[0] = b.x; lambda.pdf[1] = b.y; lambda.pdf[2] = b.z; lambda.pdf[3] = b.w;
            }
            uint32_t fl = first_bounce ? 0u : pa.rec[path].flags;
            int depth = (int)(fl & 0xffu);
            bool specular_bounce = (fl >> 8) & 1u;
            bool any_non_specular_bounces = (fl >> 9) & 1u;
            // (p_b is only read by the MIS weight of an emitter that was hit, eta_scale after sample_f: loaded there, not held across the vertex)
            Float p_b = 0.0f, eta_scale = 0.0f;
            // the previous vertex's context is only needed for the MIS weight of an emitter that was hit
            auto load_prev_ctx = [&]() {
                LightSampleContext c;
                float4 c0 = pa.ctx[path].c0, c1 = pa.ctx[path].c1, c2 = pa.ctx[path].c2;
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
                    // (flatten_scene lists only the infinite kinds here: light_pdf_li's area-light half — the inverted triangle sampling — folds away)
                    __builtin_assume(light.kind != SHM_LIGHT_DIFFUSE_AREA);
                    Spec le = infinite_light_le<HAS_TEX || ENV_LIGHT>(sv, light, ray_d, lambda);
                    if (depth == 0 || specular_bounce) {
                        add_l(load_beta() * le);
                    } else {
                        Float p_l = light_sampler_pmf(sv) * light_pdf_li<TRI_ONLY, HAS_TEX || ENV_LIGHT>(sv, light, load_prev_ctx(), ray_d);
                        Float w_b = power_heuristic(1, load_pb_eta().x, 1, p_l);
                        add_l(load_beta() * w_b * le);
                    }
                }
            } else {
                SurfaceInteraction si = hit_interaction<TRI_ONLY>(sv, hit, -ray_d);
                const PrimRec& prim = sv.prim_recs[hit.prim];  // (material and emitter ride in the record the interaction fetches anyway)
                // integrator.rs:798-813: emission at the hit
                if (!EMIT_INLINE && prim.area_light >= 0) {
                    // the hit is on an emitter (rare): its `L += beta * Le` — with the MIS weight's inverted light sampling — is k_emit_jobs's, after this launch; what
                    // it needs of the state this vertex overwrites goes to the side arrays now
                    pa.e_ray[path] = make_float4(ray_d.x, ray_d.y, ray_d.z, load_pb_eta().x);
                    pa.e_beta[path] = st_spec(load_beta());
                    pa.e_flags[path] = fl;
                    if (!(depth == 0 || specular_bounce)) {
                        pa.e_ctx0[path] = hit_kept ? pa.hit_prev[path] : pa.ctx[path].c0;
                        if (!ctx_as_hit) { pa.e_ctx1[path] = pa.ctx[path].c1; pa.e_ctx2[path] = pa.ctx[path].c2; }
                    }
                    push_emit = true;
                }
                if (EMIT_INLINE && prim.area_light >= 0) {
                    const ShmLight& light = sv.lights[prim.area_light];
                    Spec le = area_light_l(sv, light, si.n, -ray_d, lambda);
                    if (!is_zero(le)) {
                        if (depth == 0 || specular_bounce) {
                            add_l(load_beta() * le);
                        } else {
                            Float p_l = light_sampler_pmf(sv) * light_pdf_li<TRI_ONLY, HAS_TEX || ENV_LIGHT>(sv, light, load_prev_ctx(), ray_d);
                            Float w_l = power_heuristic(1, load_pb_eta().x, 1, p_l);
                            add_l(load_beta() * w_l * le);
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
                    uint32_t pix = first_bounce ? pa.pixel0[path] : pa.rec[path].pixel;
                    uint2 rs = first_bounce ? pa.rng0[path] : pa.rec[path].rng;
                    pix_in = pix;
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
                    // integrator.rs:837-841 + 897-963: next-event estimation; the visibility test is deferred to K3. The sampler dimensions are drawn HERE, in the
                    // reference's order (light choice, light sample, then the BSDF sample's three, then Russian roulette's) — the evaluation itself runs as the LAST
                    // thing of the vertex (K_SHADE_NEE_LAST): the light sample is the register-hungriest island of the kernel (82 VGPRs on its own), and at the end only
                    // its own inputs are live beside it, not everything the BSDF sampling, Russian roulette and the spawned ray still need. Independent computations:
                    // every value is what it was.
                    const bool do_nee = !(SHM_SHADE_SKIP & 1) && flags_is_non_specular(bsdf_flags(bsdf));
                    Float nee_u = 0.0f;
                    V2 nee_u_light = v2(0.0f, 0.0f);
                    if (do_nee) {
                        nee_u = sampler_get_1d(rng);
                        nee_u_light = sampler_get_2d(rng);
                    }
                    const Spec beta_at_vertex = load_beta();  // (the throughput BEFORE this vertex's update: what weighs the light's contribution)
                    auto next_event_estimation = [&]() {
                        LightSampleContext ctx = light_ctx_from(si);
                        uint32_t bf = bsdf_flags(bsdf);
                        if (flags_is_reflective(bf) && !flags_is_transmissive(bf)) ctx.pi = p3i_exact(offset_ray_origin(si.pi, si.n, si.wo));
                        else if (flags_is_transmissive(bf) && !flags_is_reflective(bf)) ctx.pi = p3i_exact(offset_ray_origin(si.pi, si.n, -si.wo));
                        Float p_sel = 0.0f;
                        int li = light_sampler_sample(sv, nee_u, p_sel);
                        if (li >= 0) {
                            const ShmLight& light = sv.lights[li];
                            LightLiSample ls;
                            if (light_sample_li<TRI_ONLY, HAS_TEX || ENV_LIGHT>(sv, light, ctx, nee_u_light, lambda, ls) && !(is_zero(ls.l) || ls.pdf == 0.0f)) {
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
                                    pa.shadow_contrib[path] = st_spec(beta_at_vertex * ld);
                                    push_shadow = true;
                                }
                            }
                        }
                    };
                    if (!K_SHADE_NEE_LAST && do_nee) next_event_estimation();
                    // integrator.rs:843-857: sample the BSDF
                    V3 wo = -ray_d;
                    Float u = sampler_get_1d(rng);
                    V2 u2 = sampler_get_2d(rng);
                    BSDFSample bs;
                    if ((SHM_SHADE_SKIP & 4) || !bsdf_sample_f(bsdf, wo, u, u2, REFLTRANS_ALL, bs)) {
                        alive = false;
                    } else {
                        // integrator.rs:859-872
                        beta = beta_at_vertex * (bs.f * abs_dot(bs.wi, si.shading.n) / bs.pdf);
                        p_b = bs.pdf_is_proportional ? bsdf_pdf(bsdf, wo, bs.wi, REFLTRANS_ALL) : bs.pdf;
                        specular_bounce = flags_is_specular(bs.flags);
                        any_non_specular_bounces |= !specular_bounce;
                        eta_scale = load_pb_eta().y;
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
                            pa.rec[path].beta = st_spec(beta);
                            pa.rec[path].pb_eta = make_float2(p_b, eta_scale);
                            if (ctx_as_hit) {
                                if (!hit_kept) pa.ctx[path].c0 = make_float4(__int_as_float(hit.prim), hit.b0, hit.b1, hit.b2);
                            } else {
                                pa.ctx[path].c0 = make_float4(nctx.pi.x.low, nctx.pi.y.low, nctx.pi.z.low, nctx.pi.x.high);
                                pa.ctx[path].c1 = make_float4(nctx.pi.y.high, nctx.pi.z.high, nctx.n.x, nctx.n.y);
                                pa.ctx[path].c2 = make_float4(nctx.n.z, nctx.ns.x, nctx.ns.y, nctx.ns.z);
                            }
                            pa.rec[path].rng = make_uint2((uint32_t)rng.state, (uint32_t)(rng.state >> 32));
                            if (first_bounce) { pa.rec[path].lambda = lambda4_in; pa.rec[path].pixel = pix_in; }  // the record's first sector, complete
                            uint32_t aux_bit = 0u;
                            if (HAS_TEX) {  // spawn_ray_with_differentials, interaction.rs:430-514
                                AuxRays na = spawn_ray_differentials(si, df, aux, bs.wi, bs.flags, bs.eta);
                                if (na.has) { st_aux(pa, path, na); aux_bit = 1u << 10; }
                            }
                            pa.rec[path].flags = (uint32_t)depth | ((uint32_t)specular_bounce << 8) | ((uint32_t)any_non_specular_bounces << 9) | aux_bit;
                            push_next = true;
                        }
                    }
                    if (K_SHADE_NEE_LAST && do_nee) next_event_estimation();
                }
                // terminate_secondary may have changed the pdfs (material.rs:609-619): written back only then
                // (only DielectricMaterial terminates wavelengths: nothing to write back in the all-diffuse instantiation)
                if (!DIFFUSE_ONLY && (lambda.pdf[1] != pdf_in.y || lambda.pdf[2] != pdf_in.z || lambda.pdf[3] != pdf_in.w || lambda.pdf[0] != pdf_in.x))
                    pa.lambda_pdf[path] = make_float4(lambda.pdf[0], lambda.pdf[1], lambda.pdf[2], lambda.pdf[3]);
            }
        }
        // stage the queue entries of this chunk in LDS (wave-aggregated LDS atomics)
        uint32_t s1 = queue_push_slot(&s_cnt[0], push_next);
        if (push_next) s_next[s1] = path;
        uint32_t s2 = queue_push_slot(&s_cnt[1], push_shadow);
        if (push_shadow) s_shadow[s2] = path;
        if (!EMIT_INLINE) {
            uint32_t s3 = queue_push_slot(&s_cnt[2], push_emit);
            if (push_emit) s_emit[s3] = path;
        }
      }
      __syncthreads();
      if (threadIdx.x == 0) {
          s_base[0] = s_cnt[0] ? atomicAdd(&qs->n_active[cur ^ 1], s_cnt[0]) : 0u;
          s_base[1] = s_cnt[1] ? atomicAdd(&qs->n_shadow[shadow_parity], s_cnt[1]) : 0u;
          s_base[2] = (!EMIT_INLINE && s_cnt[2]) ? atomicAdd(&qs->n_emit, s_cnt[2]) : 0u;
      }
      __syncthreads();
      for (uint32_t j = threadIdx.x; j < s_cnt[0]; j += SHADE2_BLOCK) q_next[s_base[0] + j] = s_next[j];
      for (uint32_t j = threadIdx.x; j < s_cnt[1]; j += SHADE2_BLOCK) q_shadow[s_base[1] + j] = s_shadow[j];
      if (!EMIT_INLINE) for (uint32_t j = threadIdx.x; j < s_cnt[2]; j += SHADE2_BLOCK) q_emit[s_base[2] + j] = s_emit[j];
      __syncthreads();
    }
    (void)counters;
}

// ---------------------------------------------------------------------------------------------
// Emission at the hit (integrator.rs:798-813: `L += beta * Le`, MIS-weighted against the light sampler after a non-specular bounce) for the vertices the fused
// kernel deferred. Hitting an emitter is rare (S3's window: two triangles), but the branch — `light_pdf_li` inverts the spherical-triangle sampling — sat in the middle
// of the vertex kernel and cost EVERY vertex registers: without it the kernel takes 8.8 ms less per headline frame (timing variant SHM_SHADE_SKIP=8). The few paths
// of q_emit are worked off here with dense lanes, from the hit record (K2's, untouched), the wavelengths, and the e_* copies of what the vertex kernel overwrote. Same
// functions, same operands, same order as the inline branch; `L` receives the sum at the same place in the path's sequence (after the previous bounce's shadow
// contribution, before this bounce's: the launcher puts this kernel between k_shade and K3).
// ---------------------------------------------------------------------------------------------
template <bool TRI_ONLY, bool HAS_TEX>
__global__ void __launch_bounds__(SHADE2_BLOCK) k_emit_jobs(SceneView sv, PathArrays pa, const uint32_t* __restrict__ q_emit, const QueueState* qs, int ctx_as_hit) {
    const uint32_t n = qs->n_emit;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t path = q_emit[i];
        Hit hit;
        hit = load_hit_tri(pa, path);
        const float4 er = pa.e_ray[path];
        const V3 ray_d = v3(er.x, er.y, er.z);
        Wavelengths lambda;
        {
            const float4 a = pa.lambda[path], b = pa.lambda_pdf[path];  // (the film's copy: valid from k_generate on, also for a bounce-0 hit whose record does not exist yet)
            lambda.lambda[0] = a.x; lambda.lambda[1] = a.y; lambda.lambda[2] = a.z; lambda.lambda[3] = a.w;
            lambda.pdf[0] = b.x; lambda.pdf[1] = b.y; lambda.pdf[2] = b.z; lambda.pdf[3] = b.w;
        }
        const uint32_t fl = pa.e_flags[path];
        const int depth = (int)(fl & 0xffu);
        const bool specular_bounce = (fl >> 8) & 1u;
        const SurfaceInteraction si = hit_interaction<TRI_ONLY>(sv, hit, -ray_d);
        const PrimRec& prim = sv.prim_recs[hit.prim];  // (material and emitter ride in the record the interaction fetches anyway)
        const ShmLight& light = sv.lights[prim.area_light];
        const Spec le = area_light_l(sv, light, si.n, -ray_d, lambda);
        if (!is_zero(le)) {
            auto add_l = [&](const Spec& c) { pa.L[path] = st_spec(ld_spec(pa.L[path]) + c); };
            auto load_beta = [&]() { return ld_spec(pa.e_beta[path]); };
            if (depth == 0 || specular_bounce) {
                add_l(load_beta() * le);
            } else {
                LightSampleContext c;
                const float4 c0 = pa.e_ctx0[path];
                if (ctx_as_hit) {
                    // the previous vertex left its hit record: its context is light_ctx_from(its interaction) — the same function of the same inputs as when it was shaded
                    Hit ph;
                    ph.prim = __float_as_int(c0.x); ph.t = 0.0f; ph.b0 = c0.y; ph.b1 = c0.z; ph.b2 = c0.w; ph.phi = 0.0f; ph.inst = -1;
                    // (get_bsdf belongs to it: a material with a displacement — the reference's constant one included — resets the shading geometry, interaction.rs:223-245)
                    SurfaceInteraction sp = hit_interaction<TRI_ONLY>(sv, ph, v3s(0.0f));
                    Wavelengths lw = lambda;
                    (void)get_bsdf<HAS_TEX>(sv, sp, sv.materials[sv.prim_recs[ph.prim].material], lw);
                    c = light_ctx_from(sp);
                } else {
                    const float4 c1 = pa.e_ctx1[path], c2 = pa.e_ctx2[path];
                    c.pi.x = iv2(c0.x, c0.w);
                    c.pi.y = iv2(c0.y, c1.x);
                    c.pi.z = iv2(c0.z, c1.y);
                    c.n = v3(c1.z, c1.w, c2.x);
                    c.ns = v3(c2.y, c2.z, c2.w);
                }
                const Float p_l = light_sampler_pmf(sv) * light_pdf_li<TRI_ONLY, HAS_TEX>(sv, light, c, ray_d);
                const Float w_l = power_heuristic(1, er.w, 1, p_l);
                add_l(load_beta() * w_l * le);
            }
        }
    }
}
}  // namespace


#define WF_EMIT_JOBS_LAUNCH(CTX_AS_HIT)                                                                                                      \
    do {                                                                                                                                     \
        hipLaunchKernelGGL((k_emit_jobs<true, false>), dim3(s->n_cu * 4), dim3(SHADE2_BLOCK), 0, a.stream, s->dsv, s->pa, s->d_q_emit, s->d_qs, (CTX_AS_HIT)); \
        LAUNCH_TRY("k_emit_jobs");                                                                                                           \
    } while (0)
#define WF_SHADE_LAUNCH(KERNEL)                                                                                                              \
    do {                                                                                                                                     \
        hipLaunchKernelGGL(KERNEL, dim3(s->n_cu * K_SHADE_LEAN_WAVES), dim3(SHADE2_BLOCK), 0, a.stream, s->dsv, s->pa, s->d_q_active[a.cur], s->d_q_active[a.cur ^ 1], \
                           s->d_q_shadow, s->d_qs, a.cur, a.params, s->d_counters, a.shadow_parity, (const uint32_t*)nullptr, s->d_q_emit, s->lds_tables, a.first_bounce | (CTX_AS_HIT_FLAG)); \
        LAUNCH_TRY("k_shade");                                                                                                               \
    } while (0)
#define WF_SHADE_LAUNCH_DIVERTED(KERNEL)                                                                                                     \
    do {                                                                                                                                     \
        hipLaunchKernelGGL(KERNEL, dim3(s->n_cu * K_SHADE_LEAN_WAVES), dim3(SHADE2_BLOCK), 0, a.stream, s->dsv, s->pa, s->d_q_lean, s->d_q_active[a.cur ^ 1], \
                           s->d_q_shadow, s->d_qs, a.cur, a.params, s->d_counters, a.shadow_parity, (const uint32_t*)&s->d_qs->n_lean, s->d_q_emit, s->lds_tables, 0); \
        LAUNCH_TRY("k_shade (diverted)");                                                                                                    \
    } while (0)
