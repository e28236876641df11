// k_vertex.inl — staged shading, first half of a path vertex (integrator.rs:772-834 for the vertex K2 found): escaped rays and
// infinite lights, the interaction, emission with its MIS weight, get_bsdf (compute_differentials, MixMaterial resolution, bump /
// normal maps, every texture evaluation -> the BxDF's parameters) and the depth test. What survives is written as a parameter block
// (PathArrays::bx, one BxRec, and ctx, one CtxRec) and pushed to the queue of its BxDF CLASS, so that the second half (k_scatter.inl: NEE, sample_f,
// Russian roulette) runs one kernel per class over material-sorted, wave-coherent queues (north star: "wavefront-sorted queues";
// SURVEY K5). Neither half holds the other's live state: the fused kernel of round 1 spilled up to 942 VGPRs in its general
// instantiations. The arithmetic of a path is the fused kernel's, operation for operation: films stay bit-identical.
#pragma once
#include "wavefront.h"

// K_ENV_LIGHT (a translation unit's switch, round 5): compile ImageInfinitelight's look-up / sample / pdf into a kernel of the class WITHOUT textures — a scene whose only
// image is an environment map needs nothing else of the textured class (k_shade_lean_env.hip says why); the *_env.hip units define it
#ifndef K_ENV_LIGHT
#define K_ENV_LIGHT false
#endif
#if K_ENV_LIGHT  // (the *_env.hip units' kernels carry their own names: a kernel trace tells them from the units without the light — tools/kernel_coverage.py)
#define k_vertex k_vertex_env
#define k_vertex_w3 k_vertex_w3_env
#endif
namespace {

// The queue K2 leaves is in image order: neighbouring lanes hit different materials, and this half of the vertex is where materials differ most
// (which textures are filtered and how, which spectra are looked up). SORT: the workgroup first counting-sorts its 2048-entry chunk by the
// material of the primitive each path hit (64 bins + one for escaped rays; two passes over hit.prim -> primitive.material, LDS atomics), then
// works through the chunk in that order, so that the 64 lanes of a wave mostly evaluate ONE material. Paths are independent and every later
// queue is order-agnostic: films and counters do not change.
constexpr int VERTEX_SORT_BINS = 64;
// q_lean (may be null): the lean diversion. A hit on a plain DiffuseMaterial in a scene without textures is not worked on here at all:
// the path goes to q_lean and the FUSED kernel (k_shade.inl's all-diffuse instantiation) runs its whole vertex — for that class the staged pair
// only adds the parameter block's traffic (headline frame staged: 130 -> 165 ms). Everything else — escaped rays, the other
// materials, MixMaterial (resolved in get_bsdf) — stays on the staged path.
constexpr int N_VERTEX_QUEUES = N_BXDF_CLASSES + 1;
template <bool TRI_ONLY, bool HAS_TEX, bool SORT>
__device__ __forceinline__ void vertex_body(const SceneView& sv, const PathArrays& pa, const uint32_t* __restrict__ q_cur, uint32_t* q_s0, uint32_t* q_s1,
                                            uint32_t* q_s2, uint32_t* q_s3, QueueState* qs, int cur, const ShmRenderParams& params, uint32_t* q_lean, const uint32_t* n_in) {
    const uint32_t n = n_in ? *n_in : qs->n_active[cur];  // (n_in: the split pass's queue, whose count is not n_active)
    const bool divert = !HAS_TEX && q_lean != nullptr;
    __shared__ uint32_t s_q[!HAS_TEX ? N_VERTEX_QUEUES : N_BXDF_CLASSES][SHADE_CHUNK];  // (the fifth queue only where the diversion can happen)
    __shared__ uint32_t s_cnt[N_VERTEX_QUEUES], s_base[N_VERTEX_QUEUES];
    __shared__ uint32_t s_sorted[SORT ? SHADE_CHUNK : 1];
    __shared__ uint32_t s_bin[SORT ? VERTEX_SORT_BINS + 1 : 1];
    __shared__ uint8_t s_sorted_key[SORT ? SHADE_CHUNK : 4];  // the sort key (material, 64 = escaped) of each sorted entry: the diversion test needs nothing else
    uint32_t* const q_out[N_VERTEX_QUEUES] = {q_s0, q_s1, q_s2, q_s3, q_lean};
    for (uint32_t chunk0 = blockIdx.x * SHADE_CHUNK; chunk0 < n; chunk0 += gridDim.x * SHADE_CHUNK) {
      if (threadIdx.x < N_VERTEX_QUEUES) s_cnt[threadIdx.x] = 0;
      if (SORT) {
          auto key_of = [&](uint32_t path) -> uint32_t {
              const int prim = pa.hit16 ? hit_prim_of(__float_as_int(reinterpret_cast<const float*>(reinterpret_cast<const float4*>(pa.hit) + path)[0])) : __float_as_int(reinterpret_cast<const float*>(pa.hit + path)[0]);
              if (prim < 0) return (uint32_t)VERTEX_SORT_BINS;
              const uint32_t m = sv.prim_recs[prim].material;
              return m < (uint32_t)VERTEX_SORT_BINS ? m : (uint32_t)VERTEX_SORT_BINS - 1u;
          };
          if (threadIdx.x <= VERTEX_SORT_BINS) s_bin[threadIdx.x] = 0;
          __syncthreads();
          // (the counting pass keeps each entry's path and key in registers for the scatter pass: the q -> hit -> primitive -> material chain of
          //  dependent gathers is walked once per entry, not twice)
          uint32_t my_path[SHADE_CHUNK / SHADE2_BLOCK];
          uint32_t my_keys = 0u, my_keys_hi = 0u;  // 8 keys of 7 bits (0..64)
#pragma unroll
          for (uint32_t k = 0; k < SHADE_CHUNK / SHADE2_BLOCK; ++k) {
              const uint32_t i = chunk0 + k * SHADE2_BLOCK + threadIdx.x;
              my_path[k] = 0u;
              if (i < n) {
                  my_path[k] = q_cur[i];
                  const uint32_t key = key_of(my_path[k]);
                  if (k < 4) my_keys |= key << (8u * k); else my_keys_hi |= key << (8u * (k - 4u));
                  atomicAdd(&s_bin[key], 1u);
              }
          }
          __syncthreads();
          if (threadIdx.x == 0) {  // exclusive prefix over 65 bins: the bins become cursors
              uint32_t acc = 0;
              for (int b = 0; b <= VERTEX_SORT_BINS; ++b) { const uint32_t c = s_bin[b]; s_bin[b] = acc; acc += c; }
          }
          __syncthreads();
#pragma unroll
          for (uint32_t k = 0; k < SHADE_CHUNK / SHADE2_BLOCK; ++k) {
              const uint32_t i = chunk0 + k * SHADE2_BLOCK + threadIdx.x;
              const uint32_t key = ((k < 4 ? my_keys >> (8u * k) : my_keys_hi >> (8u * (k - 4u)))) & 0xffu;
              if (i < n) { const uint32_t slot = atomicAdd(&s_bin[key], 1u); s_sorted[slot] = my_path[k]; s_sorted_key[slot] = (uint8_t)key; }
          }
      }
      __syncthreads();
      for (uint32_t k = 0; k < SHADE_CHUNK / SHADE2_BLOCK; ++k) {
        const uint32_t i = chunk0 + k * SHADE2_BLOCK + threadIdx.x;
        int push_class = -1;
        uint32_t path = 0;
        // a plain-diffuse hit that the fused kernel takes over is recognised by its sort key alone (the material index, when it is below the last,
        // shared bin): no hit record, no primitive record for it
        bool diverted_by_key = false;
        if (SORT && divert && i < n) {
            const uint32_t key = s_sorted_key[k * SHADE2_BLOCK + threadIdx.x];
            if (key < (uint32_t)VERTEX_SORT_BINS - 1u && sv.materials[key].kind == SHM_MATERIAL_DIFFUSE) {
                diverted_by_key = true;
                path = s_sorted[k * SHADE2_BLOCK + threadIdx.x];
                push_class = N_BXDF_CLASSES;
            }
        }
        if (i < n && !diverted_by_key) {
            path = SORT ? s_sorted[k * SHADE2_BLOCK + threadIdx.x] : q_cur[i];
            Hit hit;
            hit = load_hit_tri(pa, path);  // (the 32-byte ShmHit, or — triangle scenes, whatever instantiation shades them: a textured triangle scene runs the general ones — the compact form)
            const float4* rp = reinterpret_cast<const float4*>(pa.ray + path);
            float4 r0 = rp[0], r1 = rp[1];
            V3 ray_d = v3(r0.w, r1.x, r1.y);
            auto add_l = [&](const Spec& c) { pa.L[path] = st_spec(ld_spec(pa.L[path]) + c); };
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
            const bool specular_bounce = (fl >> 8) & 1u;
            // beta, p_b and the previous vertex's context are only needed when something is emitted towards the path
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
            auto emit = [&](const Spec& le, const ShmLight& light) {  // integrator.rs:779-792 / 802-812
                Spec beta = ld_spec(pa.rec[path].beta);
                if (depth == 0 || specular_bounce) {
                    add_l(beta * le);
                } else {
                    Float p_b = pa.rec[path].pb_eta.x;
                    Float p_l = light_sampler_pmf(sv) * light_pdf_li<TRI_ONLY, HAS_TEX || K_ENV_LIGHT>(sv, light, load_prev_ctx(), ray_d);
                    Float w = power_heuristic(1, p_b, 1, p_l);
                    add_l(beta * w * le);
                }
            };
            if (hit.prim < 0) {
                // integrator.rs:776-794: escaped ray, infinite lights
                for (uint32_t li = 0; li < sv.n_infinite_lights; ++li) {
                    const ShmLight& light = sv.lights[sv.infinite_lights[li]];
                    // (flatten_scene lists only the infinite kinds here: light_pdf_li's area-light half — the inverted triangle sampling — folds away)
                    __builtin_assume(light.kind != SHM_LIGHT_DIFFUSE_AREA);
                    emit(infinite_light_le<HAS_TEX || K_ENV_LIGHT>(sv, light, ray_d, lambda), light);
                }
            } else if (divert && sv.materials[sv.prim_recs[hit.prim].material].kind == SHM_MATERIAL_DIFFUSE) {
                push_class = N_BXDF_CLASSES;  // the fused kernel takes this vertex from its start (emission included)
            } else {
                SurfaceInteraction si = hit_interaction<TRI_ONLY>(sv, hit, -ray_d);
                const PrimRec& prim = sv.prim_recs[hit.prim];  // (material and emitter ride in the record the interaction fetches anyway)
                // integrator.rs:798-813: emission at the hit
                if (prim.area_light >= 0) {
                    const ShmLight& light = sv.lights[prim.area_light];
                    Spec le = area_light_l(sv, light, si.n, -ray_d, lambda);
                    if (!is_zero(le)) emit(le, light);
                }
                // get_bsdf starts with compute_differentials(ray, camera, spp) (interaction.rs:197)
                Differentials df;
                if (HAS_TEX) {
                    AuxRays aux = aux_none();
                    if (fl & (1u << 10)) aux = ld_aux(pa, path);
                    df = compute_differentials(sv, si, aux, params.samples_per_pixel, params.disable_pixel_jitter != 0, params.disable_texture_filtering != 0);
                }
                BSDF bsdf = get_bsdf<HAS_TEX>(sv, si, sv.materials[prim.material], lambda, &df);
                if (depth != params.max_depth) {  // integrator.rs:830-834
                    const BxDF& b = bsdf.bxdf;
                    pa.bx[path].bx0 = st_spec(b.r);
                    pa.bx[path].bx1 = st_spec(b.k);  // (the same 32-byte sector as fr)
                    pa.bx[path].bx2 = make_float4(b.eta, b.mf.alpha_x, b.mf.alpha_y,
                                               __uint_as_float(b.kind | ((uint32_t)b.max_depth << 8) | ((uint32_t)b.n_samples << 20)));  // (b.strict is sv.quirks_off: not carried)
                    if (pa.has_layered) {
                        pa.bx[path].bx3 = st_spec(b.albedo);
                        pa.bx[path].bx4 = make_float4(b.mf2.alpha_x, b.mf2.alpha_y, b.thickness, b.g);
                    }
                    const V3 fx = bsdf.shading_frame.x;
                    pa.bx[path].fr = make_float4(fx.x, fx.y, fx.z, 0.0f);
                    if (!TRI_ONLY) pa.bx[path].siwo = make_float4(si.wo.x, si.wo.y, si.wo.z, 0.0f);  // differs from -ray.d for quadrics / instances
                    // this vertex's LightSampleContext (light.rs:1001-1009): the scatter half's geometry, and the next vertex's prev_intr_ctx
                    pa.ctx[path].c0 = make_float4(si.pi.x.low, si.pi.y.low, si.pi.z.low, si.pi.x.high);
                    pa.ctx[path].c1 = make_float4(si.pi.y.high, si.pi.z.high, si.n.x, si.n.y);
                    pa.ctx[path].c2 = make_float4(si.n.z, si.shading.n.x, si.shading.n.y, si.shading.n.z);
                    if (HAS_TEX) {  // the surface part of spawn_ray_with_differentials (interaction.rs:436-440)
                        V3 dndx = si.shading.dndu * df.dudx + si.shading.dndv * df.dvdx;
                        V3 dndy = si.shading.dndu * df.dudy + si.shading.dndv * df.dvdy;
                        pa.dd0[path] = make_float4(df.dpdx.x, df.dpdx.y, df.dpdx.z, df.dpdy.x);
                        pa.dd1[path] = make_float4(df.dpdy.y, df.dpdy.z, dndx.x, dndx.y);
                        pa.dd2[path] = make_float4(dndx.z, dndy.x, dndy.y, dndy.z);
                    }
                    push_class = bxdf_class_of(b.kind);
                }
                // terminate_secondary may have changed the pdfs (material.rs:609-619): written back only then
                if (lambda.pdf[1] != pdf_in.y || lambda.pdf[2] != pdf_in.z || lambda.pdf[3] != pdf_in.w || lambda.pdf[0] != pdf_in.x)
                    pa.lambda_pdf[path] = make_float4(lambda.pdf[0], lambda.pdf[1], lambda.pdf[2], lambda.pdf[3]);
            }
        }
        // stage the class queues of this chunk in LDS (wave-aggregated LDS atomics), material-sorted by construction
#pragma unroll
        for (int c = 0; c < N_VERTEX_QUEUES; ++c) {
            if (c == N_BXDF_CLASSES && !divert) break;
            const bool mine = push_class == c;
            uint32_t slot = queue_push_slot(&s_cnt[c], mine);
            if (mine) s_q[c][slot] = path;
        }
      }
      __syncthreads();
      if (threadIdx.x < N_BXDF_CLASSES) s_base[threadIdx.x] = s_cnt[threadIdx.x] ? atomicAdd(&qs->n_scatter[threadIdx.x], s_cnt[threadIdx.x]) : 0u;
      if (threadIdx.x == N_BXDF_CLASSES) s_base[N_BXDF_CLASSES] = s_cnt[N_BXDF_CLASSES] ? atomicAdd(&qs->n_lean, s_cnt[N_BXDF_CLASSES]) : 0u;
      __syncthreads();
#pragma unroll
      for (int c = 0; c < N_VERTEX_QUEUES; ++c) {
          if (c == N_BXDF_CLASSES && !divert) break;
          for (uint32_t j = threadIdx.x; j < s_cnt[c]; j += SHADE2_BLOCK) q_out[c][s_base[c] + j] = s_q[c][j];
      }
      __syncthreads();
    }
}

template <bool TRI_ONLY, bool HAS_TEX, bool SORT = false>
__global__ void __launch_bounds__(SHADE2_BLOCK) K_SHADE_ATTR k_vertex(SceneView sv_global, PathArrays pa, const uint32_t* __restrict__ q_cur, uint32_t* q_s0, uint32_t* q_s1,
                                                                     uint32_t* q_s2, uint32_t* q_s3, QueueState* qs, int cur, ShmRenderParams params, uint32_t* q_lean,
                                                                     LdsTables lds_tables, const uint32_t* n_in) {
    __shared__ uint4 s_tables[LDS_TABLE_BUDGET / 16];  // the small scene tables, staged once per workgroup (wavefront.h, stage_scene_tables)
    __shared__ uint4 s_view[HAS_TEX ? SCENE_VIEW_UINT4S : 1];  // ... and, with textures, the view itself: what the texture evaluators that are real calls read the scene through
    const SceneView sv = stage_scene_tables_tex<HAS_TEX>(sv_global, lds_tables, s_tables, s_view);
    vertex_body<TRI_ONLY, HAS_TEX, SORT>(sv, pa, q_cur, q_s0, q_s1, q_s2, q_s3, qs, cur, params, q_lean, n_in);
}
// three waves per SIMD (<= 168 VGPRs): the triangle-only instantiation needs 159 and is bound by the latency of its gathers
template <bool TRI_ONLY, bool HAS_TEX, bool SORT = false>
__global__ void __launch_bounds__(SHADE2_BLOCK) __attribute__((amdgpu_waves_per_eu(3, 3))) k_vertex_w3(SceneView sv_global, PathArrays pa, const uint32_t* __restrict__ q_cur,
                                                                                                        uint32_t* q_s0, uint32_t* q_s1, uint32_t* q_s2, uint32_t* q_s3,
                                                                                                        QueueState* qs, int cur, ShmRenderParams params, uint32_t* q_lean,
                                                                                                        LdsTables lds_tables, const uint32_t* n_in) {
    // (three workgroups of this kernel share a CU's 160 KB with 41-51 KB each of sort bins and queues: 1.5 KB are left for tables — the material table of most scenes)
    __shared__ uint4 s_tables[LDS_TABLE_BUDGET_SMALL / 16];
    __shared__ uint4 s_view[HAS_TEX ? SCENE_VIEW_UINT4S : 1];
    const SceneView sv = stage_scene_tables_tex<HAS_TEX>(sv_global, lds_tables, s_tables, s_view);
    vertex_body<TRI_ONLY, HAS_TEX, SORT>(sv, pa, q_cur, q_s0, q_s1, q_s2, q_s3, qs, cur, params, q_lean, n_in);
}

}  // namespace

#define WF_VERTEX_LAUNCH_W3(TRI, TEX, SORT)                                                                                                    \
    do {                                                                                                                                       \
        hipLaunchKernelGGL((k_vertex_w3<TRI, TEX, SORT>), dim3(a.blocks * 3 / 2), dim3(SHADE2_BLOCK), 0, a.stream, s->dsv, s->pa, a.q_in ? a.q_in : s->d_q_active[a.cur], \
                           s->d_q_scatter[0], s->d_q_scatter[1], s->d_q_scatter[2], s->d_q_scatter[3], s->d_qs, a.cur, a.params,               \
                           (a.params.force_diffuse == 0 && s->lean_divert && !a.q_in) ? s->d_q_lean : (uint32_t*)nullptr, s->lds_tables_small, a.n_in);           \
        LAUNCH_TRY("k_vertex_w3");                                                                                                             \
    } while (0)
#define WF_VERTEX_LAUNCH(TRI, TEX, SORT)                                                                                                       \
    do {                                                                                                                                       \
        hipLaunchKernelGGL((k_vertex<TRI, TEX, SORT>), dim3(a.blocks), dim3(SHADE2_BLOCK), 0, a.stream, s->dsv, s->pa, a.q_in ? a.q_in : s->d_q_active[a.cur],          \
                           s->d_q_scatter[0], s->d_q_scatter[1], s->d_q_scatter[2], s->d_q_scatter[3], s->d_qs, a.cur, a.params,               \
                           (a.params.force_diffuse == 0 && s->lean_divert && !a.q_in) ? s->d_q_lean : (uint32_t*)nullptr, s->lds_tables, a.n_in);                                 \
        LAUNCH_TRY("k_vertex");                                                                                                                \
    } while (0)
