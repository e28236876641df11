// k_shade_other.hip — the reference's two debugging integrators on the same queues and path state: SimplePathIntegrator::li
// (integrator.rs:586-733) and RandomWalkIntegrator (integrator.rs:445-563).
#include "wavefront.h"

namespace {

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
    __shared__ uint4 s_view[SCENE_VIEW_UINT4S];  // the texture evaluators that are real calls read the scene through the workgroup's copy (shm/texture.h)
    attach_call_copy(sv, s_view);
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
            Spec beta = ld_spec(pa.rec[path].beta);
            Wavelengths lambda;
            float4 pdf_in;
            {
                float4 a = pa.rec[path].lambda, b = pa.lambda_pdf[path];
                pdf_in = b;
                lambda.lambda[0] = a.x; lambda.lambda[1] = a.y; lambda.lambda[2] = a.z; lambda.lambda[3] = a.w;
                lambda.pdf[0] = b.x; lambda.pdf[1] = b.y; lambda.pdf[2] = b.z; lambda.pdf[3] = b.w;
            }
            uint32_t fl = pa.rec[path].flags;
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
                    uint32_t pix = pa.rec[path].pixel;
                    uint2 rs = pa.rec[path].rng;
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
                            pdf = uniform_hemisphere_pdf(sv.quirks_off != 0);
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
                        pa.rec[path].beta = st_spec(beta);
                        pa.rec[path].rng = make_uint2((uint32_t)rng.state, (uint32_t)(rng.state >> 32));
                        pa.rec[path].flags = (uint32_t)depth | ((specular_bounce ? 0u : 1u) << 8);
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
    __shared__ uint4 s_view[SCENE_VIEW_UINT4S];  // the texture evaluators that are real calls read the scene through the workgroup's copy (shm/texture.h)
    attach_call_copy(sv, s_view);
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
                float4 a = pa.rec[path].lambda, b = pa.lambda_pdf[path];
                pdf_in = b;
                lambda.lambda[0] = a.x; lambda.lambda[1] = a.y; lambda.lambda[2] = a.z; lambda.lambda[3] = a.w;
                lambda.pdf[0] = b.x; lambda.pdf[1] = b.y; lambda.pdf[2] = b.z; lambda.pdf[3] = b.w;
            }
            const uint32_t fl = pa.rec[path].flags;
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
                    uint32_t pix = pa.rec[path].pixel;
                    uint2 rs = pa.rec[path].rng;
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
                        pa.rec[path].rng = make_uint2((uint32_t)rng.state, (uint32_t)(rng.state >> 32));
                        pa.rec[path].flags = (uint32_t)(depth + 1);
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
    int t = (int)(pa.rec[slot].flags & 0xffu);
    Spec l = ld_spec(rw[(size_t)(2 * t) * capacity + slot]);
    for (int k = t - 1; k >= 0; --k) {
        Spec le = ld_spec(rw[(size_t)(2 * k) * capacity + slot]);
        Spec fcos = ld_spec(rw[(size_t)(2 * k + 1) * capacity + slot]);
        l = le + fcos * l / (1.0f / (4.0f * PI_F));
    }
    pa.L[slot] = st_spec(l);
}

}  // namespace

int wf_launch_shade_simple(ShmScene* s, const ShadeArgs& a) {
    hipLaunchKernelGGL(k_shade_simple, dim3(a.blocks), dim3(SHADE2_BLOCK), 0, a.stream, s->dsv, s->pa, s->d_q_active[a.cur], s->d_q_active[a.cur ^ 1],
                       s->d_q_shadow, s->d_qs, a.cur, a.params, a.shadow_parity);
    LAUNCH_TRY("k_shade_simple");
    return SHM_OK;
}
int wf_launch_shade_randomwalk(ShmScene* s, const ShadeArgs& a, uint32_t cap_eff) {
    hipLaunchKernelGGL(k_shade_randomwalk, dim3(a.blocks), dim3(SHADE2_BLOCK), 0, a.stream, s->dsv, s->pa, s->d_q_active[a.cur], s->d_q_active[a.cur ^ 1],
                       s->d_qs, a.cur, a.params, s->d_rw, cap_eff);
    LAUNCH_TRY("k_shade_randomwalk");
    return SHM_OK;
}
int wf_launch_fold_randomwalk(ShmScene* s, hipStream_t stream, uint32_t cap_eff, uint32_t total) {
    hipLaunchKernelGGL(k_fold_randomwalk, dim3((total + SHADE_BLOCK - 1) / SHADE_BLOCK), dim3(SHADE_BLOCK), 0, stream, s->pa, s->d_rw, cap_eff, total);
    LAUNCH_TRY("k_fold_randomwalk");
    return SHM_OK;
}
