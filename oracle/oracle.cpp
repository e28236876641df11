// oracle/oracle.cpp — CPU restatement of Shimmer's tile-parallel path-integrator loop.
//
// *** TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// *** build, load or call this.  The product (libshimmer_hip.so) never links or calls anything here.
//
// What is restated here, literally, in the reference's own control structure (scalar, one path at a time):
//   integrator.rs:226-322   ImageTileIntegrator::render  — spp waves 1,1,2,4,..,64 over 8x8 tiles; worker threads
//                           pull tiles from a shared queue (the rayon par_iter shape); per-pixel ordered f64 film sums
//   integrator.rs:326-396   evaluate_pixel_sample
//   integrator.rs:748-895   PathIntegrator::li
//   integrator.rs:897-963   PathIntegrator::sample_ld
//   integrator.rs:586-733   SimplePathIntegrator::li
//   integrator.rs:491-563   RandomWalkIntegrator::li_random_walk (recursive, as there)
//   interaction.rs:187-278  get_bsdf's frame: compute_differentials first, force_diffuse last (the body is shm/path.h get_bsdf)
//   primitive.rs:136-176    TransformedPrimitive::{intersect, intersect_predicate}: the nested traversal of an instanced aggregate
//   integrator.rs:100-116   IntegratorBase::{intersect, intersect_predicate, unoccluded}, SHADOW_EPSILON
//   aggregate.rs:71-203     BvhAggregate::{intersect, intersect_predicate} (64-entry stack, near child first)
//   film.rs:548-574         RgbFilm::add_sample
// Leaf arithmetic (vector math, intervals, triangle/sphere intersection, BxDFs, light sampling, camera,
// sensor) is single-source with the HIP kernels: the headers under shimmer_amd/csrc/shm/ are compiled here
// by the host compiler with -ffp-contract=off so that CPU and GPU evaluate bit-identical IEEE operations
// (one ulp of difference flips Russian-roulette / hit decisions and would void a per-pixel tolerance).
// Those headers are pinned independently: tests/test_oracle_golden.py checks them against the reference's
// in-source known answers (aggregate.rs:575-702, shape/shape.rs:299-342, bxdf.rs:1839-1903, float.rs:172-211,
// sampling.rs:801-836, interval.rs:534-554, square_matrix.rs:623-650, transform.rs:801-943, spectra/spectrum.rs:655-785,
// vecmath/vector.rs:1601-1755, vecmath/normal.rs:903-996, bounding_box.rs:1038-1046, math.rs:549-570) and
// against float32 numpy re-evaluations of the cited formulas
// (tests/golden/gen_golden.py).  The reference itself cannot be built here (Rust nightly + crates.io;
// no toolchain, no network), so there is no oracle/_ref.
//
// PARITY PINNING: pinned at function level by the vectors above; the whole-image result is NOT pinned
// against the Rust binary, because the reference's sample stream is not reproducible (sampler.rs:117-121,
// integrator.rs:252-255): the deterministic per-pixel stream of shm/sampling.h is used instead.
// Parity UNPINNED boundaries (no reference value exists; the choice is defined in the shared headers and documented in
// DESIGN.md §2 / §4b): the inner random walks of LayeredBxDF and the MixMaterial choice (the reference seeds both from OS
// entropy), the FMA order of fast_polynomial::poly in RgbSigmoidPolynomial (crate not vendored), BilinearPatch as a whole
// (no in-source known answers: checked against float64 evaluations of the cited formulas, tests/test_bilinear_patch.py),
// and image textures as a whole (mipmap.rs / texture.rs / camera.rs differentials carry no tests in the reference; rgb2spec's
// `fetch` and f32::log2 are not vendored): float64 re-evaluations and construction properties in tests/test_textures.py.
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#define SHM_ORACLE_REFERENCE_STREAM 1  // shm/sampling.h: a sampler may carry the reference's own SmallRng stream (orc_render_reference_stream, below)
#include "../shimmer_amd/csrc/host/flatten.h"
#include "../shimmer_amd/csrc/shm/path.h"

using namespace shm;

namespace {

struct Counters {
    uint64_t rays_closest = 0, rays_any = 0, nodes_closest = 0, tris_closest = 0, nodes_any = 0, tris_any = 0, paths = 0;
    void add(const Counters& o) {
        rays_closest += o.rays_closest; rays_any += o.rays_any; nodes_closest += o.nodes_closest;
        tris_closest += o.tris_closest; nodes_any += o.nodes_any; tris_any += o.tris_any; paths += o.paths;
    }
};

// aggregate.rs:71-139
bool bvh_intersect_from(const SceneView& sv, uint32_t root, V3 ro, V3 rd, Float t_max, Hit& hit, Counters& c);
bool bvh_intersect(const SceneView& sv, V3 ro, V3 rd, Float t_max, Hit& hit, Counters& c) {
    c.rays_closest++;
    return bvh_intersect_from(sv, 0, ro, rd, t_max, hit, c);
}
// BvhAggregate::intersect of the tree rooted at `root` (the top-level aggregate, or the aggregate of an instanced object)
bool bvh_intersect_from(const SceneView& sv, uint32_t root, V3 ro, V3 rd, Float t_max, Hit& hit, Counters& c) {
    hit.prim = -1;
    hit.inst = -1;
    if (sv.n_nodes == 0) return false;
    V3 inv_dir = v3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
    int dir_is_neg[3] = {inv_dir.x < 0.0f, inv_dir.y < 0.0f, inv_dir.z < 0.0f};
    bool found = false;
    int to_visit_offset = 0;
    uint32_t current = root;
    uint32_t nodes_to_visit[64];
    for (;;) {
        const ShmBvhNode& node = sv.nodes[current];
        c.nodes_closest++;
        if (intersect_p_cached(node.bmin, node.bmax, ro, t_max, inv_dir, dir_is_neg)) {
            if (node.n_prims > 0) {
                for (uint32_t i = 0; i < node.n_prims; ++i) {
                    c.tris_closest++;
                    const uint32_t slot = node.offset + i;
                    if (sv.prim_recs[slot].kind_index & PRIM_INSTANCE_BIT) {
                        // TransformedPrimitive::intersect, primitive.rs:158-171: the ray goes into the instance's space with
                        // apply_ray_inverse (t_max shrinks by the origin's error step), the instanced aggregate is traversed, and
                        // the hit keeps the instance-space parameters; hit_interaction maps the interaction back.
                        const ShmInstance& in = sv.instances[sv.prim_recs[slot].kind_index & PRIM_INDEX_MASK];
                        Float t_inst = t_max;
                        Ray r = xf_ray_inverse(in.primitive_from_render, ro, rd, t_inst);
                        Hit h2;
                        if (bvh_intersect_from(sv, in.root_node, r.o, r.d, t_inst, h2, c)) {
                            hit = h2;
                            hit.inst = (int32_t)slot;
                            t_max = hit.t;
                            found = true;
                        }
                        continue;
                    }
                    Hit h1;
                    if (prim_intersect(sv, slot, ro, rd, t_max, h1)) {
                        hit = h1;
                        t_max = hit.t;
                        found = true;
                    }
                }
                if (to_visit_offset == 0) break;
                current = nodes_to_visit[--to_visit_offset];
            } else {
                if (dir_is_neg[node.axis]) {
                    nodes_to_visit[to_visit_offset++] = current + 1;
                    current = node.offset;
                } else {
                    nodes_to_visit[to_visit_offset++] = node.offset;
                    current = current + 1;
                }
            }
        } else {
            if (to_visit_offset == 0) break;
            current = nodes_to_visit[--to_visit_offset];
        }
    }
    return found;
}

// aggregate.rs:141-203
bool bvh_intersect_predicate_from(const SceneView& sv, uint32_t root, V3 ro, V3 rd, Float t_max, Counters& c);
bool bvh_intersect_predicate(const SceneView& sv, V3 ro, V3 rd, Float t_max, Counters& c) {
    c.rays_any++;
    return bvh_intersect_predicate_from(sv, 0, ro, rd, t_max, c);
}
bool bvh_intersect_predicate_from(const SceneView& sv, uint32_t root, V3 ro, V3 rd, Float t_max, Counters& c) {
    if (sv.n_nodes == 0) return false;
    V3 inv_dir = v3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
    int dir_is_neg[3] = {inv_dir.x < 0.0f, inv_dir.y < 0.0f, inv_dir.z < 0.0f};
    int to_visit_offset = 0;
    uint32_t current = root;
    uint32_t nodes_to_visit[64];
    for (;;) {
        const ShmBvhNode& node = sv.nodes[current];
        c.nodes_any++;
        if (intersect_p_cached(node.bmin, node.bmax, ro, t_max, inv_dir, dir_is_neg)) {
            if (node.n_prims > 0) {
                for (uint32_t i = 0; i < node.n_prims; ++i) {
                    c.tris_any++;
                    const uint32_t slot = node.offset + i;
                    if (sv.prim_recs[slot].kind_index & PRIM_INSTANCE_BIT) {
                        // TransformedPrimitive::intersect_predicate, primitive.rs:173-176: the FORWARD apply_ray, as written there
                        const ShmInstance& in = sv.instances[sv.prim_recs[slot].kind_index & PRIM_INDEX_MASK];
                        if (sv.quirks_off) {  // PBRT-v4: TransformedPrimitive::IntersectP maps the ray as Intersect does (ApplyInverse, t_max with it)
                            Float t_inst = t_max;
                            Ray ri = xf_ray_inverse(in.primitive_from_render, ro, rd, t_inst);
                            if (bvh_intersect_predicate_from(sv, in.root_node, ri.o, ri.d, t_inst, c)) return true;
                            continue;
                        }
                        Ray w;
                        w.o = ro; w.d = rd;
                        Ray r = xf_ray(in.render_from_primitive, w);  // the origin is exact: no error step, t_max unchanged
                        if (bvh_intersect_predicate_from(sv, in.root_node, r.o, r.d, t_max, c)) return true;
                        continue;
                    }
                    Hit h;
                    if (prim_intersect(sv, slot, ro, rd, t_max, h)) return true;
                }
                if (to_visit_offset == 0) break;
                current = nodes_to_visit[--to_visit_offset];
            } else {
                if (dir_is_neg[node.axis]) {
                    nodes_to_visit[to_visit_offset++] = current + 1;
                    current = node.offset;
                } else {
                    nodes_to_visit[to_visit_offset++] = node.offset;
                    current = current + 1;
                }
            }
        } else {
            if (to_visit_offset == 0) break;
            current = nodes_to_visit[--to_visit_offset];
        }
    }
    return false;
}

const Float SHADOW_EPSILON = 0.0001f;  // integrator.rs:66

// integrator.rs:897-963
Spec sample_ld(const SceneView& sv, const SurfaceInteraction& intr, const BSDF& bsdf, const Wavelengths& lambda,
               Rng& rng, Counters& c) {
    LightSampleContext ctx = light_ctx_from(intr);
    uint32_t flags = bsdf_flags(bsdf);
    if (flags_is_reflective(flags) && !flags_is_transmissive(flags)) ctx.pi = p3i_exact(offset_ray_origin(intr.pi, intr.n, intr.wo));
    else if (flags_is_transmissive(flags) && !flags_is_reflective(flags)) ctx.pi = p3i_exact(offset_ray_origin(intr.pi, intr.n, -intr.wo));
    Float u = sampler_get_1d(rng);
    Float p_sel = 0.0f;
    int li = light_sampler_sample(sv, u, p_sel);
    V2 u_light = sampler_get_2d(rng);
    if (li < 0) return spec_const(0.0f);
    const ShmLight& light = sv.lights[li];
    LightLiSample ls;
    if (!light_sample_li<false, true>(sv, light, ctx, u_light, lambda, ls)) return spec_const(0.0f);
    if (is_zero(ls.l) || ls.pdf == 0.0f) return spec_const(0.0f);
    V3 wo = intr.wo;
    V3 wi = ls.wi;
    Spec f = bsdf_f(bsdf, wo, wi) * abs_dot(wi, intr.shading.n);
    if (is_zero(f)) return spec_const(0.0f);
    // unoccluded(): !intersect_predicate(p0.spawn_ray_to_interaction(p1), 1 - SHADOW_EPSILON), integrator.rs:114-116
    Ray sr = spawn_ray_to_both_offset(intr.pi, intr.n, ls.p_light_pi, ls.p_light_n);
    if (bvh_intersect_predicate(sv, sr.o, sr.d, 1.0f - SHADOW_EPSILON, c)) return spec_const(0.0f);
    Float p_l = p_sel * ls.pdf;
    if (light_is_delta(light)) return ls.l * f / p_l;
    Float p_b = bsdf_pdf(bsdf, wo, wi, REFLTRANS_ALL);
    Float w_l = power_heuristic(1, p_l, 1, p_b);
#ifdef ORC_TRACE
    if (!(f.v[0] == f.v[0]) || !(p_b == p_b) || !(w_l == w_l) || !(p_l == p_l))
        fprintf(stderr, "[trace] sample_ld: f %g %g %g %g  p_b %g  p_l %g  w_l %g  ls.l %g  ls.pdf %g  wo.ns %g wi.ns %g\n", f.v[0], f.v[1], f.v[2], f.v[3], p_b, p_l, w_l, ls.l.v[0],
                ls.pdf, dot(wo, intr.shading.n), dot(wi, intr.shading.n));
#endif
    return w_l * ls.l * f / p_l;
}

// integrator.rs:748-895
// What the reference's get_bsdf needs beyond the interaction when image textures are bound: compute_differentials reads the ray's
// auxiliary rays, the camera, the sampler's spp and options.disable_pixel_jitter (interaction.rs:187-197).
struct TexParams {
    bool on;  // the scene binds image textures; off: differentials are dead values and not computed
    int spp;
    bool disable_pixel_jitter;
    bool force_diffuse;  // options.force_diffuse (interaction.rs:256-275)
    bool disable_texture_filtering;
};
BSDF get_bsdf_at(const SceneView& sv, SurfaceInteraction& si, const ShmMaterial& m, Wavelengths& lambda, const TexParams& tp, const AuxRays& aux,
                 Differentials& df, Rng& rng) {
    BSDF bsdf;
    if (!tp.on) {
        bsdf = get_bsdf(sv, si, m, lambda);
    } else {
        df = compute_differentials(sv, si, aux, tp.spp, tp.disable_pixel_jitter, tp.disable_texture_filtering);
        bsdf = get_bsdf<true>(sv, si, m, lambda, &df);
    }
    if (tp.force_diffuse) {  // rho_hd(wo, &[sampler.get_1d()], &[sampler.get_2d()])
        Float uc = sampler_get_1d(rng);
        V2 u2 = sampler_get_2d(rng);
        bsdf_force_diffuse(bsdf, si.wo, uc, u2);
    }
    return bsdf;
}

Spec li(const SceneView& sv, Ray ray, AuxRays aux, const TexParams& tp, Wavelengths& lambda, Rng& rng, int max_depth, bool regularize, Counters& c) {
    Spec l = spec_const(0.0f);
    Spec beta = spec_const(1.0f);
    int depth = 0;
    Float p_b = 1.0f;
    Float eta_scale = 1.0f;
    bool specular_bounce = false;
    bool any_non_specular_bounces = false;
    LightSampleContext prev_intr_ctx;
    prev_intr_ctx.pi = p3i_exact(v3s(0.0f));
    prev_intr_ctx.n = v3s(0.0f);
    prev_intr_ctx.ns = v3s(0.0f);
    for (;;) {
        Hit hit;
        bool found = bvh_intersect(sv, ray.o, ray.d, infinity(), hit, c);
        if (!found) {
            for (uint32_t k = 0; k < sv.n_infinite_lights; ++k) {
                const ShmLight& light = sv.lights[sv.infinite_lights[k]];
                Spec le = infinite_light_le<true>(sv, light, ray.d, lambda);  // light.rs:795-797 / 900-904
                if (depth == 0 || specular_bounce) {
                    l = l + beta * le;
                } else {
                    Float p_l = light_sampler_pmf(sv) * light_pdf_li<false, true>(sv, light, prev_intr_ctx, ray.d);
                    Float w_b = power_heuristic(1, p_b, 1, p_l);
                    l = l + beta * w_b * le;
                }
            }
            break;
        }
        SurfaceInteraction si = hit_interaction(sv, hit, -ray.d);
        const ShmPrimitive& prim = sv.primitives[hit.prim];
        // si.intr.le(-ray.d, lambda), interaction.rs:369-377
        if (prim.area_light >= 0) {
            const ShmLight& light = sv.lights[prim.area_light];
            Spec le = area_light_l(sv, light, si.n, -ray.d, lambda);
            if (!is_zero(le)) {
                if (depth == 0 || specular_bounce) {
                    l = l + beta * le;
                } else {
                    Float p_l = light_sampler_pmf(sv) * light_pdf_li<false, true>(sv, light, prev_intr_ctx, ray.d);
                    Float w_l = power_heuristic(1, p_b, 1, p_l);
                    l = l + beta * w_l * le;
                }
            }
        }
        Differentials df;
        BSDF bsdf = get_bsdf_at(sv, si, sv.materials[prim.material], lambda, tp, aux, df, rng);
        if (regularize && any_non_specular_bounces) bxdf_regularize(bsdf.bxdf);
        if (depth == max_depth) break;
        depth += 1;
        if (flags_is_non_specular(bsdf_flags(bsdf))) {
            Spec ld = sample_ld(sv, si, bsdf, lambda, rng, c);
            l = l + beta * ld;
#ifdef ORC_TRACE
            if (!(ld.v[0] == ld.v[0]) || !(beta.v[0] == beta.v[0])) fprintf(stderr, "[trace] depth %d: ld %g %g %g %g beta %g kind %d\n", depth, ld.v[0], ld.v[1], ld.v[2], ld.v[3], beta.v[0], bsdf.bxdf.kind);
#endif
        }
        V3 wo = -ray.d;
        Float u = sampler_get_1d(rng);
        V2 u2 = sampler_get_2d(rng);
        BSDFSample bs;
        if (!bsdf_sample_f(bsdf, wo, u, u2, REFLTRANS_ALL, bs)) break;
#ifdef ORC_TRACE
        if (!(bs.f.v[0] == bs.f.v[0]) || !(bs.pdf == bs.pdf) || bs.pdf == 0.0f || !(bs.wi.x == bs.wi.x))
            fprintf(stderr, "[trace] depth %d: sample_f f %g %g %g %g pdf %g wi %g %g %g flags %u kind %d wo.n %g\n", depth, bs.f.v[0], bs.f.v[1], bs.f.v[2], bs.f.v[3], bs.pdf,
                    bs.wi.x, bs.wi.y, bs.wi.z, bs.flags, bsdf.bxdf.kind, dot(wo, si.shading.n));
#endif
        beta = beta * (bs.f * abs_dot(bs.wi, si.shading.n) / bs.pdf);
        p_b = bs.pdf_is_proportional ? bsdf_pdf(bsdf, wo, bs.wi, REFLTRANS_ALL) : bs.pdf;
        specular_bounce = flags_is_specular(bs.flags);
        any_non_specular_bounces |= !specular_bounce;
        if (flags_is_transmissive(bs.flags)) eta_scale *= sqr(bs.eta);
        prev_intr_ctx = light_ctx_from(si);
        // spawn_ray_with_differentials: interaction.spawn_ray(wi) (interaction.rs:68-75, 441) + the specular differentials
        if (tp.on) aux = spawn_ray_differentials(si, df, aux, bs.wi, bs.flags, bs.eta);
        ray.o = offset_ray_origin(si.pi, si.n, bs.wi);
        ray.d = bs.wi;
        if (is_finite(eta_scale)) {
            Spec rr_beta = beta * eta_scale;
            if (max_component_value(rr_beta) < 1.0f && depth > 1) {
                Float q = max(0.0f, 1.0f - max_component_value(rr_beta));
                if (sampler_get_1d(rng) < q) break;
                beta = beta / (1.0f - q);
            }
        }
    }
    return l;
}

// SimplePathIntegrator::li, integrator.rs:586-733: no MIS, no Russian roulette; direct lighting by light sampling
// (sample_lights) or only by hitting emitters; directions from the BSDF (sample_bsdf) or uniform over the (hemi)sphere.
Spec li_simple_path(const SceneView& sv, Ray ray, AuxRays aux, const TexParams& tp, Wavelengths& lambda, Rng& rng, int max_depth, bool sample_lights,
                    bool sample_bsdf, Counters& c) {
    Spec l = spec_const(0.0f);
    bool specular_bounce = true;
    Spec beta = spec_const(1.0f);
    int depth = 0;
    while (!is_zero(beta)) {
        Hit hit;
        if (!bvh_intersect(sv, ray.o, ray.d, infinity(), hit, c)) {
            if (!sample_lights || specular_bounce)
                for (uint32_t k = 0; k < sv.n_infinite_lights; ++k) {
                    const ShmLight& light = sv.lights[sv.infinite_lights[k]];
                    l = l + beta * infinite_light_le<true>(sv, light, ray.d, lambda);
                }
            break;
        }
        SurfaceInteraction si = hit_interaction(sv, hit, -ray.d);
        const ShmPrimitive& prim = sv.primitives[hit.prim];
        if (!sample_lights || specular_bounce) {
            if (prim.area_light >= 0) l = l + beta * area_light_l(sv, sv.lights[prim.area_light], si.n, -ray.d, lambda);
            else l = l + beta * spec_const(0.0f);  // isect.le() of a non-emitter is a zero spectrum that is still added (interaction.rs:369-377)
        }
        if (depth == max_depth) break;
        depth += 1;
        Differentials df;
        BSDF bsdf = get_bsdf_at(sv, si, sv.materials[prim.material], lambda, tp, aux, df, rng);
        aux = aux_none();  // every later ray is interaction.spawn_ray(wi): no auxiliary rays (integrator.rs:686, 716)
        V3 wo = -ray.d;
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
                        if (!bvh_intersect_predicate(sv, sr.o, sr.d, 1.0f - SHADOW_EPSILON, c)) l = l + beta * f * ls.l / (p_sel * ls.pdf);
                    }
                }
            }
        }
        if (sample_bsdf) {
            Float u = sampler_get_1d(rng);
            V2 u2 = sampler_get_2d(rng);
            BSDFSample bs;
            if (!bsdf_sample_f(bsdf, wo, u, u2, REFLTRANS_ALL, bs)) break;
            beta = beta * (bs.f * abs_dot(bs.wi, si.shading.n) / bs.pdf);
            specular_bounce = flags_is_specular(bs.flags);
            ray.o = offset_ray_origin(si.pi, si.n, bs.wi);
            ray.d = bs.wi;
        } else {
            uint32_t flags = bsdf_flags(bsdf);
            Float pdf;
            V3 wi;
            if (flags_is_reflective(flags) && flags_is_transmissive(flags)) {
                wi = sample_uniform_sphere(sampler_get_2d(rng));
                pdf = uniform_sphere_pdf();
            } else {
                wi = sample_uniform_hemisphere(sampler_get_2d(rng));
                pdf = uniform_hemisphere_pdf(sv.quirks_off != 0);
                if ((flags_is_reflective(flags) && dot(wo, si.n) * dot(wi, si.n) < 0.0f) ||
                    (flags_is_transmissive(flags) && dot(wo, si.n) * dot(wi, si.n) > 0.0f))
                    wi = -wi;
            }
            beta = beta * (bsdf_f(bsdf, wo, wi) * abs_dot(wi, si.shading.n) / pdf);
            specular_bounce = false;
            ray.o = offset_ray_origin(si.pi, si.n, wi);
            ray.d = wi;
        }
    }
    return l;
}

// RandomWalkIntegrator::li_random_walk, integrator.rs:491-563 (recursive, as there: the innermost term is evaluated first)
Spec li_random_walk(const SceneView& sv, Ray ray, const AuxRays& aux, const TexParams& tp, Wavelengths& lambda, Rng& rng, int depth, int max_depth,
                    Counters& c) {
    Hit hit;
    if (!bvh_intersect(sv, ray.o, ray.d, infinity(), hit, c)) {
        Spec le = spec_const(0.0f);
        for (uint32_t k = 0; k < sv.n_infinite_lights; ++k) {
            const ShmLight& light = sv.lights[sv.infinite_lights[k]];
            le = le + infinite_light_le<true>(sv, light, ray.d, lambda);
        }
        return le;
    }
    SurfaceInteraction si = hit_interaction(sv, hit, -ray.d);
    const ShmPrimitive& prim = sv.primitives[hit.prim];
    V3 wo = -ray.d;
    Spec le = (prim.area_light >= 0) ? area_light_l(sv, sv.lights[prim.area_light], si.n, wo, lambda) : spec_const(0.0f);
    if (depth == max_depth) return le;
    Differentials df;
    BSDF bsdf = get_bsdf_at(sv, si, sv.materials[prim.material], lambda, tp, aux, df, rng);
    V2 u = sampler_get_2d(rng);
    V3 wp = sample_uniform_sphere(u);
    Spec f = bsdf_f(bsdf, wo, wp);
    if (is_zero(f)) return le;
    Spec fcos = f * abs_dot(wp, si.shading.n);
    Ray next;
    next.o = offset_ray_origin(si.pi, si.n, wp);
    next.d = wp;
    return le + fcos * li_random_walk(sv, next, aux_none(), tp, lambda, rng, depth + 1, max_depth, c) / (1.0f / (4.0f * PI_F));
}

struct Oracle {
    shm_host::FlatScene flat;
    SceneView sv;
};

thread_local std::string g_err;

}  // namespace

#pragma GCC visibility push(default)
extern "C" {

struct OrcScene;

const char* orc_last_error() { return g_err.c_str(); }

int orc_scene_create(const ShmSceneDesc* desc, OrcScene** out) {
    if (!out) return SHM_ERR_INVALID_ARGUMENT;
    Oracle* o = new Oracle();
    int rc = shm_host::flatten_scene(desc, o->flat, g_err);
    if (rc != SHM_OK) { delete o; return rc; }
    o->sv = o->flat.view();
    *out = reinterpret_cast<OrcScene*>(o);
    return SHM_OK;
}
void orc_scene_destroy(OrcScene* s) { delete reinterpret_cast<Oracle*>(s); }

int orc_trace_closest(OrcScene* s, const ShmRay* rays, uint32_t n, ShmHit* hits, ShmStats* stats) {
    Oracle* o = reinterpret_cast<Oracle*>(s);
    Counters c;
    for (uint32_t i = 0; i < n; ++i) {
        Hit h;
        bvh_intersect(o->sv, v3(rays[i].o[0], rays[i].o[1], rays[i].o[2]), v3(rays[i].d[0], rays[i].d[1], rays[i].d[2]), rays[i].t_max, h, c);
        memset(&hits[i], 0, sizeof(ShmHit));
        hits[i].prim = h.prim;
        if (h.prim >= 0) { hits[i].t = h.t; hits[i].b0 = h.b0; hits[i].b1 = h.b1; hits[i].b2 = h.b2; hits[i].phi = h.phi; hits[i].instance = (uint32_t)(h.inst + 1); }
    }
    if (stats) { memset(stats, 0, sizeof(*stats)); stats->rays_closest = c.rays_closest; stats->nodes_closest = c.nodes_closest; stats->tris_closest = c.tris_closest; }
    return SHM_OK;
}
int orc_trace_any(OrcScene* s, const ShmRay* rays, uint32_t n, uint8_t* occluded, ShmStats* stats) {
    Oracle* o = reinterpret_cast<Oracle*>(s);
    Counters c;
    for (uint32_t i = 0; i < n; ++i)
        occluded[i] = bvh_intersect_predicate(o->sv, v3(rays[i].o[0], rays[i].o[1], rays[i].o[2]), v3(rays[i].d[0], rays[i].d[1], rays[i].d[2]), rays[i].t_max, c) ? 1 : 0;
    if (stats) { memset(stats, 0, sizeof(*stats)); stats->rays_any = c.rays_any; stats->nodes_any = c.nodes_any; stats->tris_any = c.tris_any; }
    return SHM_OK;
}

// One spp-wave over the given tiles, n_threads workers pulling tiles from a shared counter
// (integrator.rs:242-304).  film: pixel_bounds-sized, += semantics.
int orc_render_wave(OrcScene* s, const ShmRenderParams* params, const ShmTile* tiles, uint32_t n_tiles,
                    int32_t sample_begin, int32_t sample_end, int n_threads, ShmFilmPixel* film, ShmStats* stats) {
    Oracle* o = reinterpret_cast<Oracle*>(s);
    if (!params || !tiles || !film) { g_err = "invalid render arguments"; return SHM_ERR_INVALID_ARGUMENT; }
    SceneView sv = o->sv;
    sv.quirks_off = params->disable_reference_quirks ? 1u : 0u;  // SHM_REFERENCE_QUIRKS: 0 = reference-exact
    const int width = sv.pixel_bounds[2] - sv.pixel_bounds[0];
    const TexParams tp{o->flat.has_textures, params->samples_per_pixel, params->disable_pixel_jitter != 0, params->force_diffuse != 0, params->disable_texture_filtering != 0};
    if (n_threads < 1) n_threads = 1;
    std::atomic<uint32_t> next(0);
    std::vector<Counters> counters(n_threads);
    auto t0 = std::chrono::steady_clock::now();
    auto worker = [&](int tid) {
        Counters c;
        for (;;) {
            uint32_t ti = next.fetch_add(1);
            if (ti >= n_tiles) break;
            const ShmTile& tile = tiles[ti];
            for (int x = tile.x0; x < tile.x1; ++x) {
                for (int y = tile.y0; y < tile.y1; ++y) {
                    for (int si = sample_begin; si < sample_end; ++si) {
                        // evaluate_pixel_sample, integrator.rs:326-396
                        Rng rng = sampler_start_pixel_sample(x, y, si, params->seed);
                        Wavelengths lambda;
                        Float weight;
                        AuxRays aux = aux_none();
                        Ray ray = generate_camera_ray(sv, x, y, rng, params->disable_wavelength_jitter != 0,
                                                      params->disable_pixel_jitter != 0, lambda, weight, tp.on ? &aux : nullptr, params->samples_per_pixel);
                        Spec L = spec_const(1.0f) * ((params->integrator == SHM_INTEGRATOR_RANDOM_WALK)
                                                         ? li_random_walk(sv, ray, aux, tp, lambda, rng, 0, params->max_depth, c)
                                                         : (params->integrator == SHM_INTEGRATOR_SIMPLE_PATH)
                                                         ? li_simple_path(sv, ray, aux, tp, lambda, rng, params->max_depth, params->sample_lights != 0, params->sample_bsdf != 0, c)
                                                         : li(sv, ray, aux, tp, lambda, rng, params->max_depth, params->regularize != 0, c));  // camera_ray.weight * li
                        c.paths++;
                        // integrator.rs:377-382 holds two TODOs where PBRT-v4 drops a sample with a NaN or an infinite value; only with the
                        // reference's quirks switched off is that done here
                        if (sv.quirks_off && !spec_is_finite(L)) L = spec_const(0.0f);
                        // RgbFilm::add_sample, film.rs:548-574
                        V3 rgb = film_sample_rgb(sv, L, lambda);
                        ShmFilmPixel& px = film[(size_t)(y - sv.pixel_bounds[1]) * width + (x - sv.pixel_bounds[0])];
                        px.rgb_sum[0] += (double)(weight * rgb.x);
                        px.rgb_sum[1] += (double)(weight * rgb.y);
                        px.rgb_sum[2] += (double)(weight * rgb.z);
                        px.weight_sum += (double)weight;
                    }
                }
            }
        }
        counters[tid] = c;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; ++t) th.emplace_back(worker, t);
    worker(0);
    for (auto& t : th) t.join();
    auto t1 = std::chrono::steady_clock::now();
    if (stats) {
        Counters c;
        for (auto& k : counters) c.add(k);
        stats->paths += c.paths; stats->rays_closest += c.rays_closest; stats->rays_any += c.rays_any;
        stats->nodes_closest += c.nodes_closest; stats->tris_closest += c.tris_closest;
        stats->nodes_any += c.nodes_any; stats->tris_any += c.tris_any;
        stats->ms_total += std::chrono::duration<double, std::milli>(t1 - t0).count();
    }
    return SHM_OK;
}

// ImageTileIntegrator::render: all waves (integrator.rs:231-233, 306-308).
int orc_render(OrcScene* s, const ShmRenderParams* params, const ShmTile* tiles, uint32_t n_tiles, int n_threads,
               ShmFilmPixel* film, ShmStats* stats) {
    if (!params) return SHM_ERR_INVALID_ARGUMENT;
    if (stats) memset(stats, 0, sizeof(*stats));
    int spp = params->samples_per_pixel;
    int wave_start = 0, wave_end = 1, next_wave_size = 1;
    while (wave_start < spp) {
        int rc = orc_render_wave(s, params, tiles, n_tiles, wave_start, wave_end, n_threads, film, stats);
        if (rc != SHM_OK) return rc;
        wave_start = wave_end;
        wave_end = std::min(spp, wave_end + next_wave_size);
        next_wave_size = std::min(2 * next_wave_size, 64);
    }
    return SHM_OK;
}

// ImageTileIntegrator::render as `RAYON_NUM_THREADS=1 shimmer scene.pbrt --seed S` draws it (ORACLE ONLY — the product keeps the defined per-pixel stream: a sequential
// stream cannot be replayed by 268 M paths in flight). With one rayon worker the thread-local sampler of integrator.rs:252-253 is ONE clone of the prototype
// (IndependentSampler::new: SmallRng::seed_from_u64(seed), sampler.rs:103-109) that lives across tiles AND across the spp-waves (sampler_tl is made outside the wave loop,
// integrator.rs:238), start_pixel_sample is a no-op (sampler.rs:117-121), and the tiles come in index order (par_iter's halves are joined left first on one worker):
// one Xoshiro256++ stream consumed by waves -> tiles -> x (OUTER, integrator.rs:257) -> y -> sample, every dimension in evaluate_pixel_sample's / li's call order.
// Valid for scenes without LayeredBxDF / MixMaterial (those draw from SmallRng::from_entropy(), integrator.rs:255, bxdf.rs:1011: not reproducible by anybody).
// UNVERIFIED HERE: no Rust toolchain in this image — the generator is checked against the published xoshiro256++ / SplitMix64 vectors and the mode for determinism
// (tests/test_reference_stream.py); INTEGRATION.md holds the recipe for a maintainer with cargo.
int orc_render_reference_stream(OrcScene* s, const ShmRenderParams* params, const ShmTile* tiles, uint32_t n_tiles, ShmFilmPixel* film, ShmStats* stats, uint64_t* draws_out) {
    Oracle* o = reinterpret_cast<Oracle*>(s);
    if (!params || !tiles || !film) { g_err = "invalid render arguments"; return SHM_ERR_INVALID_ARGUMENT; }
    if (params->integrator != SHM_INTEGRATOR_PATH && params->integrator != SHM_INTEGRATOR_SIMPLE_PATH && params->integrator != SHM_INTEGRATOR_RANDOM_WALK) return SHM_ERR_INVALID_ARGUMENT;
    bool entropy = o->flat.has_class[3];  // (class 3: the coated materials)
    for (const ShmMaterial& m : o->flat.materials) entropy = entropy || m.kind == SHM_MATERIAL_MIX;
    if (entropy) { g_err = "reference stream: the scene holds a LayeredBxDF or a MixMaterial (the reference seeds those from OS entropy)"; return SHM_ERR_UNSUPPORTED; }
    SceneView sv = o->sv;
    sv.quirks_off = params->disable_reference_quirks ? 1u : 0u;
    const int width = sv.pixel_bounds[2] - sv.pixel_bounds[0];
    const TexParams tp{o->flat.has_textures, params->samples_per_pixel, params->disable_pixel_jitter != 0, params->force_diffuse != 0, params->disable_texture_filtering != 0};
    if (stats) memset(stats, 0, sizeof(*stats));
    RefStream stream = ref_stream_seed_from_u64(params->seed);
    const RefStream first = stream;
    uint64_t draws = 0;
    Counters c;
    const int spp = params->samples_per_pixel;
    int wave_start = 0, wave_end = 1, next_wave_size = 1;
    while (wave_start < spp) {
        for (uint32_t ti = 0; ti < n_tiles; ++ti) {
            const ShmTile& tile = tiles[ti];
            for (int x = tile.x0; x < tile.x1; ++x)
                for (int y = tile.y0; y < tile.y1; ++y)
                    for (int si = wave_start; si < wave_end; ++si) {
                        Rng rng{0u, 1u, &stream};
                        Wavelengths lambda;
                        Float weight;
                        AuxRays aux = aux_none();
                        Ray ray = generate_camera_ray(sv, x, y, rng, params->disable_wavelength_jitter != 0, params->disable_pixel_jitter != 0, lambda, weight, tp.on ? &aux : nullptr, params->samples_per_pixel);
                        Spec L = spec_const(1.0f) * ((params->integrator == SHM_INTEGRATOR_RANDOM_WALK) ? li_random_walk(sv, ray, aux, tp, lambda, rng, 0, params->max_depth, c)
                                                     : (params->integrator == SHM_INTEGRATOR_SIMPLE_PATH) ? li_simple_path(sv, ray, aux, tp, lambda, rng, params->max_depth, params->sample_lights != 0, params->sample_bsdf != 0, c)
                                                     : li(sv, ray, aux, tp, lambda, rng, params->max_depth, params->regularize != 0, c));
                        c.paths++;
                        if (sv.quirks_off && !spec_is_finite(L)) L = spec_const(0.0f);
                        V3 rgb = film_sample_rgb(sv, L, lambda);
                        ShmFilmPixel& px = film[(size_t)(y - sv.pixel_bounds[1]) * width + (x - sv.pixel_bounds[0])];
                        px.rgb_sum[0] += (double)(weight * rgb.x);
                        px.rgb_sum[1] += (double)(weight * rgb.y);
                        px.rgb_sum[2] += (double)(weight * rgb.z);
                        px.weight_sum += (double)weight;
                    }
        }
        wave_start = wave_end;
        wave_end = std::min(spp, wave_end + next_wave_size);
        next_wave_size = std::min(2 * next_wave_size, 64);
    }
    if (draws_out) {  // how far the stream went: replayed from the seed until the state recurs (the period is 2^256 - 1: the first match is the end of this render)
        RefStream r = first;
        while (memcmp(&r, &stream, sizeof(r)) != 0) { ref_stream_next_u64(r); ++draws; }
        *draws_out = draws;
    }
    if (stats) { stats->paths = c.paths; stats->rays_closest = c.rays_closest; stats->rays_any = c.rays_any; stats->nodes_closest = c.nodes_closest; stats->tris_closest = c.tris_closest; stats->nodes_any = c.nodes_any; stats->tris_any = c.tris_any; }
    return SHM_OK;
}
// the generator alone (tests/test_reference_stream.py: published vectors)
void orc_fn_xoshiro256pp(const uint64_t* state4, int n, uint64_t* out) { RefStream r; memcpy(r.s, state4, sizeof(r.s)); for (int i = 0; i < n; ++i) out[i] = ref_stream_next_u64(r); }
void orc_fn_splitmix64(uint64_t seed, int n, uint64_t* out) { for (int i = 0; i < n; ++i) out[i] = ref_splitmix64_next(seed); }
void orc_fn_reference_stream_f32(uint64_t seed, int n, float* out, uint64_t* state4_out) { RefStream r = ref_stream_seed_from_u64(seed); if (state4_out) memcpy(state4_out, r.s, sizeof(r.s)); for (int i = 0; i < n; ++i) out[i] = ref_stream_next_f32(r); }

// ---- unit-function entry points for the golden-vector tests (tests/test_oracle_golden.py) ----
float orc_fn_next_float_up(float v) { return next_float_up(v); }
float orc_fn_next_float_down(float v) { return next_float_down(v); }
float orc_fn_gamma(int n) { return gamma(n); }
float orc_fn_difference_of_products(float a, float b, float c, float d) { return difference_of_products(a, b, c, d); }
float orc_fn_lerp(float t, float a, float b) { return shm::lerp(t, a, b); }   // math.rs:246-248
// fast_polynomial::poly(x, [c0, c1, c2]) = c0 + c1 x + c2 x^2 as the path evaluates it: rgb_sigmoid's polynomial is poly(lambda, [c2, c1, c0])
// (color.rs:359), i.e. the sigmoid's argument; recovered here through the sigmoid's inverse-free form x / (2 sqrt(1 + x^2)) is not needed:
// the polynomial itself is exposed
float orc_fn_poly3(float x, float c0, float c1, float c2) { return shm::poly3(x, c0, c1, c2); }
float orc_fn_sin(float x) { return shm::sin(x); }
float orc_fn_cos(float x) { return shm::cos(x); }
float orc_fn_asin(float x) { return shm::asin(x); }
float orc_fn_acos(float x) { return shm::acos(x); }
float orc_fn_atan2(float y, float x) { return shm::atan2(y, x); }
float orc_fn_exp(float x) { return shm::exp(x); }
float orc_fn_log(float x) { return shm::log(x); }
float orc_fn_atanh(float x) { return shm::atanh(x); }
float orc_fn_cosh(float x) { return shm::cosh(x); }
float orc_fn_hypot(float x, float y) { return shm::hypot(x, y); }
float orc_fn_round(float x) { return shm::round(x); }
float orc_fn_sample_visible_wavelengths(float u) { return sample_visible_wavelengths(u); }
float orc_fn_visible_wavelengths_pdf(float l) { return visible_wavelengths_pdf(l); }
float orc_fn_dot(const float* a, const float* b) { return dot(ld3(a), ld3(b)); }
void orc_fn_cross(const float* a, const float* b, float* out) { V3 r = cross(ld3(a), ld3(b)); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void orc_fn_coordinate_system(const float* v, float* out6) {
    V3 a, b;
    coordinate_system(ld3(v), a, b);
    out6[0] = a.x; out6[1] = a.y; out6[2] = a.z; out6[3] = b.x; out6[4] = b.y; out6[5] = b.z;
}
int orc_fn_intersect_p_cached(const float* bmin, const float* bmax, const float* o, const float* d, float t_max) {
    V3 rd = ld3(d);
    V3 inv = v3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
    int neg[3] = {inv.x < 0.0f, inv.y < 0.0f, inv.z < 0.0f};
    return intersect_p_cached(bmin, bmax, ld3(o), t_max, inv, neg) ? 1 : 0;
}
int orc_fn_intersect_triangle(const float* o, const float* d, float t_max, const float* p0, const float* p1, const float* p2, float* out4) {
    TriangleIntersection ti;
    if (!intersect_triangle(ld3(o), ld3(d), t_max, ld3(p0), ld3(p1), ld3(p2), ti)) return 0;
    out4[0] = ti.b0; out4[1] = ti.b1; out4[2] = ti.b2; out4[3] = ti.t;
    return 1;
}
float orc_fn_tr_d(float ax, float ay, const float* wm) { return trowbridge_reitz_new(ax, ay).d(ld3(wm)); }
float orc_fn_tr_g(float ax, float ay, const float* wo, const float* wi) { return trowbridge_reitz_new(ax, ay).g(ld3(wo), ld3(wi)); }
float orc_fn_tr_lambda(float ax, float ay, const float* w) { return trowbridge_reitz_new(ax, ay).lambda(ld3(w)); }
void orc_fn_tr_sample_wm(float ax, float ay, const float* w, const float* u, float* out) {
    V3 r = trowbridge_reitz_new(ax, ay).sample_wm(ld3(w), v2(u[0], u[1]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
float orc_fn_fresnel_dielectric(float c, float eta) { return fresnel_dielectric(c, eta); }
float orc_fn_fresnel_complex(float c, float eta, float k) { return fresnel_complex(c, cx(eta, k)); }
// BxDF::sample_f in the local frame. out: f[4], wi[3], pdf, flags, eta (10 floats). Returns 1 if Some.
int orc_fn_bxdf_sample_f(int kind, const float* r4, const float* k4, float eta, float ax, float ay, const float* wo,
                         float uc, const float* u, float* out10) {
    BxDF b;
    b.strict = 0;
    b.kind = (uint32_t)kind;
    for (int i = 0; i < 4; ++i) { b.r.v[i] = r4[i]; b.k.v[i] = k4[i]; }
    b.eta = eta;
    b.mf = trowbridge_reitz_new(ax, ay);
    BSDFSample bs;
    if (!bxdf_sample_f(b, ld3(wo), uc, v2(u[0], u[1]), REFLTRANS_ALL, bs)) return 0;
    for (int i = 0; i < 4; ++i) out10[i] = bs.f.v[i];
    out10[4] = bs.wi.x; out10[5] = bs.wi.y; out10[6] = bs.wi.z; out10[7] = bs.pdf; out10[8] = (float)bs.flags; out10[9] = bs.eta;
    return 1;
}
void orc_fn_bxdf_f_pdf(int kind, const float* r4, const float* k4, float eta, float ax, float ay, const float* wo,
                       const float* wi, float* out5) {
    BxDF b;
    b.strict = 0;
    b.kind = (uint32_t)kind;
    for (int i = 0; i < 4; ++i) { b.r.v[i] = r4[i]; b.k.v[i] = k4[i]; }
    b.eta = eta;
    b.mf = trowbridge_reitz_new(ax, ay);
    Spec f = bxdf_f(b, ld3(wo), ld3(wi));
    for (int i = 0; i < 4; ++i) out5[i] = f.v[i];
    out5[4] = bxdf_pdf(b, ld3(wo), ld3(wi), REFLTRANS_ALL);
}
// LayeredBxDF (CoatedDiffuse kind 4 / CoatedConductor kind 5) in the local frame. p: r[4], k[4], albedo[4], eta, ax, ay,
// ax2, ay2, thickness, g (19 floats); ip: max_depth, n_samples.
static BxDF make_layered(int kind, const float* p, const int* ip) {
    BxDF b;
    b.strict = ip[1] < 0 ? 1 : 0;  // test hook: a negative n_samples asks for the PBRT-v4 guard (ShmRenderParams::disable_reference_quirks)
    b.kind = (uint32_t)kind;
    for (int i = 0; i < 4; ++i) { b.r.v[i] = p[i]; b.k.v[i] = p[4 + i]; b.albedo.v[i] = p[8 + i]; }
    b.eta = p[12];
    b.mf = trowbridge_reitz_new(p[13], p[14]);
    b.mf2 = trowbridge_reitz_new(p[15], p[16]);
    b.thickness = p[17];
    b.g = p[18];
    b.max_depth = ip[0];
    b.n_samples = ip[1] < 0 ? -ip[1] : ip[1];
    return b;
}
void orc_fn_layered_f_pdf(int kind, const float* p, const int* ip, const float* wo, const float* wi, float* out6) {
    BxDF b = make_layered(kind, p, ip);
    Spec f = bxdf_f(b, ld3(wo), ld3(wi));
    for (int i = 0; i < 4; ++i) out6[i] = f.v[i];
    out6[4] = bxdf_pdf(b, ld3(wo), ld3(wi), REFLTRANS_ALL);
    out6[5] = (float)bxdf_flags(b);
}
// LayeredBxDF::f with the whole walk evaluated (no opposite-hemisphere early-out): what the shortcut of shm/bxdf.h must equal
void orc_fn_layered_f_full(int kind, const float* p, const int* ip, const float* wo, const float* wi, float* out4) {
    BxDF b = make_layered(kind, p, ip);
    Spec f = layered_f<false>(b, ld3(wo), ld3(wi), MODE_RADIANCE);
    for (int i = 0; i < 4; ++i) out4[i] = f.v[i];
}
int orc_fn_layered_sample_f(int kind, const float* p, const int* ip, const float* wo, float uc, const float* u, float* out10) {
    BxDF b = make_layered(kind, p, ip);
    BSDFSample bs;
    if (!bxdf_sample_f(b, ld3(wo), uc, v2(u[0], u[1]), REFLTRANS_ALL, bs)) return 0;
    for (int i = 0; i < 4; ++i) out10[i] = bs.f.v[i];
    out10[4] = bs.wi.x; out10[5] = bs.wi.y; out10[6] = bs.wi.z; out10[7] = bs.pdf; out10[8] = (float)bs.flags;
    out10[9] = bs.pdf_is_proportional ? 1.0f : 0.0f;
    return 1;
}
// LayeredBxDF::sample_f through the resumable form the staged layered kernel runs (layered_sample_begin + layered_sample_step, shm/bxdf.h), `steps_per_pass`
// steps at a time with the walk's state copied between passes as the kernel's job buffer does; returns the number of steps taken in *n_steps
int orc_fn_layered_sample_f_steps(int kind, const float* p, const int* ip, const float* wo, float uc, const float* u, int steps_per_pass, float* out10, int* n_steps) {
    BxDF b = make_layered(kind, p, ip);
    BSDFSample bs;
    LayeredWalk k;
    *n_steps = 0;
    int status = layered_sample_begin(b, ld3(wo), uc, v2(u[0], u[1]), MODE_RADIANCE, bs, k);
    while (status == WALK_CONTINUES) {
        LayeredWalk resumed = k;  // (a deposit and a reload)
        for (int s = 0; s < steps_per_pass && status == WALK_CONTINUES; ++s) { status = layered_sample_step(b, MODE_RADIANCE, resumed, bs); ++*n_steps; }
        k = resumed;
    }
    if (status == WALK_FAILED) return 0;
    for (int i = 0; i < 4; ++i) out10[i] = bs.f.v[i];
    out10[4] = bs.wi.x; out10[5] = bs.wi.y; out10[6] = bs.wi.z; out10[7] = bs.pdf; out10[8] = (float)bs.flags;
    out10[9] = bs.pdf_is_proportional ? 1.0f : 0.0f;
    return 1;
}
float orc_fn_henyey_greenstein(float cos_theta, float g) { return henyey_greenstein(cos_theta, g); }
void orc_fn_sample_henyey_greenstein(const float* wo, float g, const float* u, float* out4) {
    Float pdf;
    V3 wi = sample_henyey_greenstein(ld3(wo), g, v2(u[0], u[1]), pdf);
    out4[0] = wi.x; out4[1] = wi.y; out4[2] = wi.z; out4[3] = pdf;
}
float orc_fn_sample_exponential(float x, float a) { return sample_exponential(x, a); }
void orc_fn_sample_cosine_hemisphere(const float* u, float* out3) { V3 r = sample_cosine_hemisphere(v2(u[0], u[1])); out3[0] = r.x; out3[1] = r.y; out3[2] = r.z; }
float orc_fn_power_heuristic(float f, float g) { return power_heuristic(1, f, 1, g); }
int orc_fn_sample_discrete(const float* weights, int n, float u, float* pmf, float* u_remapped) { return sample_discrete(weights, n, u, pmf, u_remapped); }
float orc_fn_sampler_stream(int px, int py, int sample_index, uint64_t seed, int n, float* out) {
    Rng r = sampler_start_pixel_sample(px, py, sample_index, seed);
    for (int i = 0; i < n; ++i) out[i] = sampler_get_1d(r);
    return n > 0 ? out[0] : 0.0f;
}
void orc_fn_offset_ray_origin(const float* p, const float* err, const float* n, const float* w, float* out3) {
    V3 r = offset_ray_origin(p3i_from_value_and_error(ld3(p), ld3(err)), ld3(n), ld3(w));
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
// Triangle light sampling from a reference point: out = p[3], n[3], pdf ; returns 1 if Some
int orc_fn_triangle_sample_with_context(const float* p0, const float* p1, const float* p2, const float* ctx_p, const float* ctx_n,
                                        const float* ctx_ns, const float* u, float* out7) {
    TriangleData tr;
    memset(&tr, 0, sizeof(tr));
    tr.p0 = ld3(p0); tr.p1 = ld3(p1); tr.p2 = ld3(p2);
    ShapeSampleContext c;
    c.pi = p3i_exact(ld3(ctx_p)); c.n = ld3(ctx_n); c.ns = ld3(ctx_ns);
    ShapeSample ss;
    if (!triangle_sample_with_context(tr, c, v2(u[0], u[1]), ss)) return 0;
    V3 p = ss.pi.mid();
    out7[0] = p.x; out7[1] = p.y; out7[2] = p.z; out7[3] = ss.n.x; out7[4] = ss.n.y; out7[5] = ss.n.z; out7[6] = ss.pdf;
    return 1;
}
// ... the same with ShmRenderParams::disable_reference_quirks' value (strict = 1: PBRT-v4's forms of the two sampling behaviours, shm/shapes.h)
int orc_fn_triangle_sample_with_context_strict(const float* p0, const float* p1, const float* p2, const float* ctx_p, const float* ctx_n,
                                               const float* ctx_ns, const float* u, int strict, float* out7) {
    TriangleData tr;
    memset(&tr, 0, sizeof(tr));
    tr.p0 = ld3(p0); tr.p1 = ld3(p1); tr.p2 = ld3(p2);
    ShapeSampleContext c;
    c.pi = p3i_exact(ld3(ctx_p)); c.n = ld3(ctx_n); c.ns = ld3(ctx_ns);
    ShapeSample ss;
    if (!triangle_sample_with_context(tr, c, v2(u[0], u[1]), ss, strict != 0)) return 0;
    V3 p = ss.pi.mid();
    out7[0] = p.x; out7[1] = p.y; out7[2] = p.z; out7[3] = ss.n.x; out7[4] = ss.n.y; out7[5] = ss.n.z; out7[6] = ss.pdf;
    return 1;
}
float orc_fn_triangle_pdf_with_context(const float* p0, const float* p1, const float* p2, const float* ctx_p, const float* ctx_n,
                                       const float* ctx_ns, const float* wi) {
    TriangleData tr;
    memset(&tr, 0, sizeof(tr));
    tr.p0 = ld3(p0); tr.p1 = ld3(p1); tr.p2 = ld3(p2);
    ShapeSampleContext c;
    c.pi = p3i_exact(ld3(ctx_p)); c.n = ld3(ctx_n); c.ns = ld3(ctx_ns);
    return triangle_pdf_with_context(tr, c, ld3(wi));
}
// ---- BilinearPatch (shape/bilinear_patch.rs) unit entry points. pts = p00, p10, p01, p11 (12 floats) ----
static PatchData make_patch(const float* pts, int flip) {
    PatchData pd;
    pd.p00 = ld3(pts); pd.p10 = ld3(pts + 3); pd.p01 = ld3(pts + 6); pd.p11 = ld3(pts + 9);
    pd.flip = flip != 0;
    pd.is_rect = blp_is_rectangle(pd.p00, pd.p10, pd.p01, pd.p11);
    pd.area = blp_area(pd.p00, pd.p10, pd.p01, pd.p11, pd.is_rect);
    pd.has_n = pd.has_uv = false;
    pd.n00 = pd.n10 = pd.n01 = pd.n11 = v3s(0.0f);
    pd.uv00 = pd.uv10 = pd.uv01 = pd.uv11 = v2(0.0f, 0.0f);
    return pd;
}
// with per-vertex attributes: normals (12 floats, p00 p10 p01 p11 order) and / or uv (8 floats); NULL = absent
static PatchData make_patch_attr(const float* pts, int flip, const float* n12, const float* uv8) {
    PatchData pd = make_patch(pts, flip);
    if (n12) { pd.has_n = true; pd.n00 = ld3(n12); pd.n10 = ld3(n12 + 3); pd.n01 = ld3(n12 + 6); pd.n11 = ld3(n12 + 9); }
    if (uv8) { pd.has_uv = true; pd.uv00 = v2(uv8[0], uv8[1]); pd.uv10 = v2(uv8[2], uv8[3]); pd.uv01 = v2(uv8[4], uv8[5]); pd.uv11 = v2(uv8[6], uv8[7]); }
    return pd;
}
// out = p[3], n[3], dpdu[3], dpdv[3], ns[3], dpdu_s[3], dpdv_s[3], uv[2] (23 floats)
void orc_fn_blp_interaction_attr(const float* pts, int flip, const float* n12, const float* uv8, float u, float v, const float* wo, float* out23) {
    SurfaceInteraction si = blp_interaction(make_patch_attr(pts, flip, n12, uv8), u, v, ld3(wo));
    const V3 vs[7] = {si.p(), si.n, si.dpdu, si.dpdv, si.shading.n, si.shading.dpdu, si.shading.dpdv};
    for (int i = 0; i < 7; ++i) { out23[3 * i] = vs[i].x; out23[3 * i + 1] = vs[i].y; out23[3 * i + 2] = vs[i].z; }
    out23[21] = si.uv.x; out23[22] = si.uv.y;
}
void orc_fn_invert_bilinear(const float* p2, const float* v8, float* out2) {
    V2 r = invert_bilinear(v2(p2[0], p2[1]), v2(v8[0], v8[1]), v2(v8[2], v8[3]), v2(v8[4], v8[5]), v2(v8[6], v8[7]));
    out2[0] = r.x; out2[1] = r.y;
}
void orc_fn_rotate_from_to(const float* from, const float* to, const float* v, float* out3) {
    V3 r = rot3_apply(rotate_from_to(ld3(from), ld3(to)), ld3(v));
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
// Interval arithmetic (interval.rs:366-414): op 0 = a * b, 1 = a / b, 2 = a + b, 3 = a - b, 4 = sqr(a), 5 = sqrt(a); out = low, high
void orc_fn_interval_op(int op, float alo, float ahi, float blo, float bhi, float* out2) {
    Interval a = iv_new(alo, ahi), b = iv_new(blo, bhi), r = a;
    switch (op) {
        case 0: r = a * b; break;
        case 1: r = a / b; break;
        case 2: r = a + b; break;
        case 3: r = a - b; break;
        case 4: r = iv_sqr(a); break;
        default: r = iv_sqrt(a); break;
    }
    out2[0] = r.low; out2[1] = r.high;
}
// SquareMatrix<3>::determinant (square_matrix.rs:281-292) as the bilinear patch uses it; m = 9 floats, row-major
float orc_fn_det3(const float* m) { return det3(ld3(m), ld3(m + 3), ld3(m + 6)); }
void orc_fn_blp_info(const float* pts, float* out2) {
    PatchData pd = make_patch(pts, 0);
    out2[0] = pd.is_rect ? 1.0f : 0.0f;
    out2[1] = pd.area;
}
// out = u, v, t ; returns 1 on a hit
int orc_fn_blp_intersect(const float* pts, const float* o, const float* d, float t_max, float* out3) {
    BilinearIntersection bi;
    if (!blp_intersect(ld3(o), ld3(d), t_max, ld3(pts), ld3(pts + 3), ld3(pts + 6), ld3(pts + 9), bi)) return 0;
    out3[0] = bi.u; out3[1] = bi.v; out3[2] = bi.t;
    return 1;
}
// out = p[3], n[3], dpdu[3], dpdv[3], dndu[3], dndv[3], p_error[3] (21 floats)
void orc_fn_blp_interaction(const float* pts, int flip, float u, float v, const float* wo, float* out21) {
    SurfaceInteraction si = blp_interaction(make_patch(pts, flip), u, v, ld3(wo));
    V3 p = si.p();
    V3 err = v3(0.5f * (si.pi.x.high - si.pi.x.low), 0.5f * (si.pi.y.high - si.pi.y.low), 0.5f * (si.pi.z.high - si.pi.z.low));
    const V3 vs[7] = {p, si.n, si.dpdu, si.dpdv, si.dndu, si.dndv, err};
    for (int i = 0; i < 7; ++i) { out21[3 * i] = vs[i].x; out21[3 * i + 1] = vs[i].y; out21[3 * i + 2] = vs[i].z; }
}
int orc_fn_blp_sample_with_context(const float* pts, int flip, const float* ctx_p, const float* ctx_n, const float* ctx_ns,
                                   const float* u, float* out7) {
    ShapeSampleContext c;
    c.pi = p3i_exact(ld3(ctx_p)); c.n = ld3(ctx_n); c.ns = ld3(ctx_ns);
    ShapeSample ss;
    if (!blp_sample_with_context(make_patch(pts, flip), c, v2(u[0], u[1]), ss)) return 0;
    V3 p = ss.pi.mid();
    out7[0] = p.x; out7[1] = p.y; out7[2] = p.z; out7[3] = ss.n.x; out7[4] = ss.n.y; out7[5] = ss.n.z; out7[6] = ss.pdf;
    return 1;
}
float orc_fn_blp_pdf_with_context(const float* pts, int flip, const float* ctx_p, const float* ctx_n, const float* ctx_ns, const float* wi) {
    ShapeSampleContext c;
    c.pi = p3i_exact(ld3(ctx_p)); c.n = ld3(ctx_n); c.ns = ld3(ctx_ns);
    return blp_pdf_with_context(make_patch(pts, flip), c, ld3(wi));
}
float orc_fn_spherical_quad_area(const float* a, const float* b, const float* c, const float* d) {
    return spherical_quad_area(ld3(a), ld3(b), ld3(c), ld3(d));
}
// out = p[3], pdf
void orc_fn_sample_spherical_rectangle(const float* p_ref, const float* s, const float* ex, const float* ey, const float* u, float* out4) {
    Float pdf = 0.0f;
    V3 p = sample_spherical_rectangle(ld3(p_ref), ld3(s), ld3(ex), ld3(ey), v2(u[0], u[1]), pdf);
    out4[0] = p.x; out4[1] = p.y; out4[2] = p.z; out4[3] = pdf;
}
void orc_fn_invert_spherical_rectangle_sample(const float* p_ref, const float* s, const float* ex, const float* ey, const float* p_rect,
                                              float* out2) {
    V2 u = invert_spherical_rectangle_sample(ld3(p_ref), ld3(s), ld3(ex), ld3(ey), ld3(p_rect));
    out2[0] = u.x; out2[1] = u.y;
}
int orc_fn_quadratic(float a, float b, float c, float* out2) {
    Float t0, t1;
    if (!quadratic(a, b, c, t0, t1)) return 0;
    out2[0] = t0; out2[1] = t1;
    return 1;
}
// Full interaction at a hit of the scene: out = p[3], n[3], ns[3], dpdu_s[3] (12 floats)
int orc_fn_hit_interaction(OrcScene* s, const ShmRay* ray, float* out12, ShmHit* hit_out) {
    Oracle* o = reinterpret_cast<Oracle*>(s);
    Counters c;
    Hit h;
    V3 ro = v3(ray->o[0], ray->o[1], ray->o[2]), rd = v3(ray->d[0], ray->d[1], ray->d[2]);
    if (!bvh_intersect(o->sv, ro, rd, ray->t_max, h, c)) return 0;
    SurfaceInteraction si = hit_interaction(o->sv, h, -rd);
    V3 p = si.p();
    out12[0] = p.x; out12[1] = p.y; out12[2] = p.z;
    out12[3] = si.n.x; out12[4] = si.n.y; out12[5] = si.n.z;
    out12[6] = si.shading.n.x; out12[7] = si.shading.n.y; out12[8] = si.shading.n.z;
    out12[9] = si.shading.dpdu.x; out12[10] = si.shading.dpdu.y; out12[11] = si.shading.dpdu.z;
    if (hit_out) { memset(hit_out, 0, sizeof(*hit_out)); hit_out->prim = h.prim; hit_out->t = h.t; hit_out->b0 = h.b0; hit_out->b1 = h.b1; hit_out->b2 = h.b2; hit_out->phi = h.phi; }
    return 1;
}
void orc_fn_camera_ray(OrcScene* s, int px, int py, int sample_index, uint64_t seed, float* out14) {
    Oracle* o = reinterpret_cast<Oracle*>(s);
    Rng rng = sampler_start_pixel_sample(px, py, sample_index, seed);
    Wavelengths lambda;
    Float w;
    Ray r = generate_camera_ray(o->sv, px, py, rng, false, false, lambda, w);
    out14[0] = r.o.x; out14[1] = r.o.y; out14[2] = r.o.z; out14[3] = r.d.x; out14[4] = r.d.y; out14[5] = r.d.z;
    for (int i = 0; i < 4; ++i) { out14[6 + i] = lambda.lambda[i]; out14[10 + i] = lambda.pdf[i]; }
}
void orc_fn_film_sample_rgb(OrcScene* s, const float* L4, const float* lambda4, const float* pdf4, float* out3) {
    Oracle* o = reinterpret_cast<Oracle*>(s);
    Spec L;
    Wavelengths w;
    for (int i = 0; i < 4; ++i) { L.v[i] = L4[i]; w.lambda[i] = lambda4[i]; w.pdf[i] = pdf4[i]; }
    V3 r = film_sample_rgb(o->sv, L, w);
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
float orc_fn_spectrum_get(OrcScene* s, const ShmSpectrum* sp, float lambda) {
    Oracle* o = reinterpret_cast<Oracle*>(s);
    return spectrum_get(*sp, o->sv.spectrum_data, lambda);
}
// Spectrum::sample at four wavelengths
void orc_fn_spectrum_sample(OrcScene* s, const ShmSpectrum* sp, const float* lambda4, float* out4) {
    Oracle* o = reinterpret_cast<Oracle*>(s);
    Wavelengths w;
    for (int i = 0; i < 4; ++i) { w.lambda[i] = lambda4[i]; w.pdf[i] = 1.0f; }
    Spec r = spectrum_sample(*sp, o->sv.spectrum_data, w);
    for (int i = 0; i < 4; ++i) out4[i] = r.v[i];
}


// Transform::apply / apply_inverse for a point (kind 0), vector (1) or normal (2) given the matrix m and its inverse m_inv
// (transform.rs:363-383, 606-629): the inverse applies swap the two matrices.
void orc_fn_transform_apply(int kind, int inverse, const float* m, const float* m_inv, const float* v, float* out3) {
    const float* fwd = inverse ? m_inv : m;
    const float* bwd = inverse ? m : m_inv;
    V3 r = (kind == 0) ? xf_point(fwd, ld3(v)) : ((kind == 1) ? xf_vector(fwd, ld3(v)) : xf_normal(bwd, ld3(v)));
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}

// vecmath: out = length(a), length_squared(a), angle_between(normalize(a), normalize(b)), normalize(a)[3], gram_schmidt(b, normalize(a))[3]
void orc_fn_vecmath(const float* a, const float* b, float* out9) {
    V3 va = ld3(a), vb = ld3(b);
    out9[0] = length(va);
    out9[1] = length_squared(va);
    out9[2] = angle_between(normalize(va), normalize(vb));
    V3 n = normalize(va);
    out9[3] = n.x; out9[4] = n.y; out9[5] = n.z;
    V3 g = gram_schmidt(vb, n);
    out9[6] = g.x; out9[7] = g.y; out9[8] = g.z;
}

// Light::preprocess: the radius of Bounds3::bounding_sphere of the scene bounds (bounding_box.rs:460-468), as flatten.h computes it
float orc_fn_scene_radius(OrcScene* s) { return reinterpret_cast<Oracle*>(s)->flat.scene_radius; }

// ---- image textures (shm/texture.h) ----
static TextureEvalContext make_tex_ctx(const float* c18) {
    TextureEvalContext c;
    c.p = ld3(c18); c.dpdx = ld3(c18 + 3); c.dpdy = ld3(c18 + 6); c.n = ld3(c18 + 9);
    c.uv = v2(c18[12], c18[13]);
    c.dudx = c18[14]; c.dudy = c18[15]; c.dvdx = c18[16]; c.dvdy = c18[17];
    return c;
}
float orc_fn_log2(float x) { return shm::log2(x); }
// TextureMapping2D::map: out = s, t, dsdx, dsdy, dtdx, dtdy
void orc_fn_texture_map(OrcScene* s, uint32_t tex, const float* ctx18, float* out6) {
    Oracle* o = reinterpret_cast<Oracle*>(s);
    TexCoord2D c = texture_map(o->sv.image_textures[tex], make_tex_ctx(ctx18));
    out6[0] = c.st.x; out6[1] = c.st.y; out6[2] = c.dsdx; out6[3] = c.dsdy; out6[4] = c.dtdx; out6[5] = c.dtdy;
}
// MIPMap::filter::<RGB>(st, dst0, dst1)
void orc_fn_texture_filter(OrcScene* s, uint32_t tex, const float* st, const float* dst0, const float* dst1, float* out3) {
    Oracle* o = reinterpret_cast<Oracle*>(s);
    TextureView tv;
    tv.t = &o->sv.image_textures[tex];
    tv.levels = o->sv.image_levels + tv.t->first_level;
    tv.texels = o->sv.texel_data;
    RGB3 r = tex_filter<RGB3>(tv, o->sv.ewa_lut, v2(st[0], st[1]), v2(dst0[0], dst0[1]), v2(dst1[0], dst1[1]));
    out3[0] = r.r; out3[1] = r.g; out3[2] = r.b;
}
void orc_fn_rgb2spec_fetch(OrcScene* s, const float* rgb, float* out3) {
    Oracle* o = reinterpret_cast<Oracle*>(s);
    rgb2spec_fetch(o->sv, rgb3(rgb[0], rgb[1], rgb[2]), out3);
}
// SpectrumImageTexture::evaluate at four wavelengths
void orc_fn_image_texture_evaluate(OrcScene* s, uint32_t tex, const float* ctx18, const float* lambda4, float* out4) {
    Oracle* o = reinterpret_cast<Oracle*>(s);
    Wavelengths w;
    for (int i = 0; i < 4; ++i) { w.lambda[i] = lambda4[i]; w.pdf[i] = 1.0f; }
    Spec r = image_texture_evaluate(o->sv, tex, make_tex_ctx(ctx18), w);
    for (int i = 0; i < 4; ++i) out4[i] = r.v[i];
}
void orc_fn_approximate_dp_dxy(OrcScene* s, const float* p, const float* n, int spp, int disable_pixel_jitter, float* out6) {
    Oracle* o = reinterpret_cast<Oracle*>(s);
    V3 dpdx, dpdy;
    approximate_dp_dxy(o->sv.camera, ld3(p), ld3(n), spp, disable_pixel_jitter != 0, dpdx, dpdy);
    out6[0] = dpdx.x; out6[1] = dpdx.y; out6[2] = dpdx.z; out6[3] = dpdy.x; out6[4] = dpdy.y; out6[5] = dpdy.z;
}
// The camera ray of a pixel sample with its (scaled) auxiliary rays, intersected with the scene; at the hit the differentials of
// compute_differentials, with the ray's auxiliary rays (use_aux) or through approximate_dp_dxy. Returns 0 on a miss.
// out[0..18): ray o, d, rx_o, rx_d, ry_o, ry_d; out[18..44): p, n, uv, dpdu, dpdv, dpdx, dpdy, dudx, dvdx, dudy, dvdy
int orc_fn_camera_hit_differentials(OrcScene* s, int px, int py, int sample_index, uint64_t seed, int spp, int disable_pixel_jitter, int use_aux,
                                    float* out) {
    Oracle* o = reinterpret_cast<Oracle*>(s);
    Rng rng = sampler_start_pixel_sample(px, py, sample_index, seed);
    Wavelengths lambda;
    Float w;
    AuxRays aux = aux_none();
    Ray r = generate_camera_ray(o->sv, px, py, rng, false, disable_pixel_jitter != 0, lambda, w, &aux, spp);
    const V3 rv[6] = {r.o, r.d, aux.rx_o, aux.rx_d, aux.ry_o, aux.ry_d};
    for (int i = 0; i < 6; ++i) { out[3 * i] = rv[i].x; out[3 * i + 1] = rv[i].y; out[3 * i + 2] = rv[i].z; }
    Hit h;
    Counters c;
    if (!bvh_intersect(o->sv, r.o, r.d, infinity(), h, c)) return 0;
    SurfaceInteraction si = hit_interaction(o->sv, h, -r.d);
    Differentials df = compute_differentials(o->sv, si, use_aux ? aux : aux_none(), spp, disable_pixel_jitter != 0);
    V3 p = si.p();
    const V3 sv3[6] = {p, si.n, v3(si.uv.x, si.uv.y, 0.0f), si.dpdu, si.dpdv, df.dpdx};
    float* q = out + 18;
    q[0] = p.x; q[1] = p.y; q[2] = p.z; q[3] = si.n.x; q[4] = si.n.y; q[5] = si.n.z; q[6] = si.uv.x; q[7] = si.uv.y;
    q[8] = si.dpdu.x; q[9] = si.dpdu.y; q[10] = si.dpdu.z; q[11] = si.dpdv.x; q[12] = si.dpdv.y; q[13] = si.dpdv.z;
    q[14] = df.dpdx.x; q[15] = df.dpdx.y; q[16] = df.dpdx.z; q[17] = df.dpdy.x; q[18] = df.dpdy.y; q[19] = df.dpdy.z;
    q[20] = df.dudx; q[21] = df.dvdx; q[22] = df.dudy; q[23] = df.dvdy;
    (void)sv3;
    return 1;
}
// spawn_ray_with_differentials for a planar interaction (p, n = ns, dndu = dndv = 0 unless given): out12 = rx_o, rx_d, ry_o, ry_d; returns has
int orc_fn_spawn_ray_differentials(const float* p, const float* n, const float* wo, const float* dpdx, const float* dpdy, const float* dndx_uv,
                                   const float* aux12, const float* wi, uint32_t flags, float eta, float* out12) {
    SurfaceInteraction si;
    memset(&si, 0, sizeof(si));
    si.pi = p3i_exact(ld3(p));
    si.n = ld3(n); si.wo = ld3(wo);
    si.shading.n = ld3(n);
    // dndx = dndu * dudx + dndv * dvdx: pass dndu = dndx_uv[0..3), dndv = dndx_uv[3..6) with (dudx, dvdx, dudy, dvdy) = (1, 0, 0, 1)
    si.shading.dndu = ld3(dndx_uv); si.shading.dndv = ld3(dndx_uv + 3);
    Differentials df = differentials_zero();
    df.dpdx = ld3(dpdx); df.dpdy = ld3(dpdy);
    df.dudx = 1.0f; df.dvdy = 1.0f;
    AuxRays a;
    a.has = true;
    a.rx_o = ld3(aux12); a.rx_d = ld3(aux12 + 3); a.ry_o = ld3(aux12 + 6); a.ry_d = ld3(aux12 + 9);
    AuxRays r = spawn_ray_differentials(si, df, a, ld3(wi), flags, eta);
    const V3 rv[4] = {r.rx_o, r.rx_d, r.ry_o, r.ry_d};
    for (int i = 0; i < 4; ++i) { out12[3 * i] = rv[i].x; out12[3 * i + 1] = rv[i].y; out12[3 * i + 2] = rv[i].z; }
    return r.has ? 1 : 0;
}


// FloatTexture::evaluate of node `index`
float orc_fn_float_texture_evaluate(OrcScene* s, uint32_t index, const float* ctx18) {
    Oracle* o = reinterpret_cast<Oracle*>(s);
    return float_texture_evaluate(o->sv, index, make_tex_ctx(ctx18));
}
// bump_map (which = 0, tex = float texture node) / normal_map (which = 1, tex = image texture) on a hand-made interaction:
// geo = p, n (geometric = shading), dpdu, dpdv, dndu, dndv (18 floats), uv, duv = dudx, dudy, dvdx, dvdy; out6 = dpdu', dpdv'
void orc_fn_bump_or_normal_map(OrcScene* s, int which, uint32_t tex, const float* geo18, const float* uv, const float* duv, float* out6) {
    Oracle* o = reinterpret_cast<Oracle*>(s);
    SurfaceInteraction si;
    memset(&si, 0, sizeof(si));
    si.pi = p3i_exact(ld3(geo18));
    si.n = ld3(geo18 + 3);
    si.uv = v2(uv[0], uv[1]);
    si.dpdu = ld3(geo18 + 6); si.dpdv = ld3(geo18 + 9); si.dndu = ld3(geo18 + 12); si.dndv = ld3(geo18 + 15);
    si.shading.n = si.n; si.shading.dpdu = si.dpdu; si.shading.dpdv = si.dpdv; si.shading.dndu = si.dndu; si.shading.dndv = si.dndv;
    Differentials df = differentials_zero();
    df.dudx = duv[0]; df.dudy = duv[1]; df.dvdx = duv[2]; df.dvdy = duv[3];
    V3 a, b;
    if (which == 0) bump_map_texture(o->sv, tex, si, df, a, b);
    else normal_map_texture(o->sv, tex, si, a, b);
    out6[0] = a.x; out6[1] = a.y; out6[2] = a.z; out6[3] = b.x; out6[4] = b.y; out6[5] = b.z;
}

// SpectrumTexture::evaluate of a material slot value (spectrum, image texture or composite node)
void orc_fn_spectrum_texture_evaluate(OrcScene* s, const ShmSpectrum* sp, const float* ctx18, const float* lambda4, float* out4) {
    Oracle* o = reinterpret_cast<Oracle*>(s);
    Wavelengths w;
    for (int i = 0; i < 4; ++i) { w.lambda[i] = lambda4[i]; w.pdf[i] = 1.0f; }
    TextureEvalContext ctx = make_tex_ctx(ctx18);
    Spec r = spectrum_texture_evaluate<true>(o->sv, *sp, &ctx, w);
    for (int i = 0; i < 4; ++i) out4[i] = r.v[i];
}

// ---- ImageInfinitelight ----
void orc_fn_equal_area_square_to_sphere(const float* uv, float* out3) {
    V3 d = equal_area_square_to_sphere(v2(uv[0], uv[1]));
    out3[0] = d.x; out3[1] = d.y; out3[2] = d.z;
}
void orc_fn_equal_area_sphere_to_square(const float* d, float* out2) {
    V2 p = equal_area_sphere_to_square(ld3(d));
    out2[0] = p.x; out2[1] = p.y;
}
// the value of ShmRenderParams::disable_reference_quirks for the scene-level leaf entries below (a render sets it from its own parameters)
void orc_set_quirks_off(OrcScene* s, int off) { reinterpret_cast<Oracle*>(s)->sv.quirks_off = off ? 1u : 0u; }
// Light::sample_li of light `li` from a point at the origin: out = wi[3], pdf, L[4]; returns 1 if Some
int orc_fn_light_sample_li(OrcScene* s, uint32_t li, const float* u, int allow_incomplete_pdf, const float* lambda4, float* out8) {
    Oracle* o = reinterpret_cast<Oracle*>(s);
    Wavelengths w;
    for (int i = 0; i < 4; ++i) { w.lambda[i] = lambda4[i]; w.pdf[i] = 1.0f; }
    LightSampleContext ctx;
    ctx.pi = p3i_exact(v3s(0.0f)); ctx.n = v3s(0.0f); ctx.ns = v3s(0.0f);
    LightLiSample ls;
    if (!light_sample_li<false, true>(o->sv, o->sv.lights[li], ctx, v2(u[0], u[1]), w, ls, allow_incomplete_pdf != 0)) return 0;
    out8[0] = ls.wi.x; out8[1] = ls.wi.y; out8[2] = ls.wi.z; out8[3] = ls.pdf;
    for (int i = 0; i < 4; ++i) out8[4 + i] = ls.l.v[i];
    return 1;
}
// ImageInfinitelight::pdf_li with either distribution
float orc_fn_image_light_pdf(OrcScene* s, uint32_t li, const float* wi, int allow_incomplete_pdf) {
    Oracle* o = reinterpret_cast<Oracle*>(s);
    const ShmLight& light = o->sv.lights[li];
    if (allow_incomplete_pdf) {
        LightSampleContext ctx;
        ctx.pi = p3i_exact(v3s(0.0f)); ctx.n = v3s(0.0f); ctx.ns = v3s(0.0f);
        return light_pdf_li<false, true>(o->sv, light, ctx, ld3(wi));
    }
    const ImageLightRec& il = o->sv.image_lights[light.primitive];
    return pc2d_pdf(o->sv.dist_data, il.distribution, (int)il.n, equal_area_sphere_to_square(xf_vector(il.light_from_render, ld3(wi)))) / (4.0f * PI_F);
}
void orc_fn_infinite_light_le(OrcScene* s, uint32_t li, const float* d, const float* lambda4, float* out4) {
    Oracle* o = reinterpret_cast<Oracle*>(s);
    Wavelengths w;
    for (int i = 0; i < 4; ++i) { w.lambda[i] = lambda4[i]; w.pdf[i] = 1.0f; }
    Spec r = infinite_light_le<true>(o->sv, o->sv.lights[li], ld3(d), w);
    for (int i = 0; i < 4; ++i) out4[i] = r.v[i];
}
// The flattened PiecewiseConstant2D of an image light: which = 0 distribution, 1 compensated. Copies func (n*n) and the marginal
// cdf (n+1); returns n.
int orc_fn_image_light_distribution(OrcScene* s, uint32_t li, int which, float* func_out, float* marginal_cdf_out, float* integral_out) {
    Oracle* o = reinterpret_cast<Oracle*>(s);
    const ImageLightRec& il = o->sv.image_lights[o->sv.lights[li].primitive];
    const Dist2DRec& d = which ? il.compensated : il.distribution;
    const int n = (int)il.n;
    if (func_out) memcpy(func_out, o->sv.dist_data + d.func, sizeof(float) * n * n);
    if (marginal_cdf_out) memcpy(marginal_cdf_out, o->sv.dist_data + d.marginal_cdf, sizeof(float) * (n + 1));
    if (integral_out) *integral_out = d.marginal_int;
    return n;
}

// ---- unit entry points for the leaf fixtures of tests/golden/golden_leaves.json (tests/test_leaf_golden.py) ----
// Triangle::interaction_from_intersection (triangle.rs:305-504). p9 = p0 p1 p2; n9 / s9 / uv6 may be NULL (mesh without them).
// out35 = pi.low[3], pi.high[3], uv[2], n[3], dpdu[3], dpdv[3], shading n[3], shading dpdu[3], shading dpdv[3], dndu[3], dndv[3],
// shading dndu[3], shading dndv[3] (dndu/dndv of the interaction stay zero: SurfaceInteraction::new receives Normal3f::ZERO)
void orc_fn_triangle_interaction(const float* p9, const float* n9, const float* s9, const float* uv6, int flip, const float* b3,
                                 const float* wo, float* out) {
    TriangleData tr;
    memset(&tr, 0, sizeof(tr));
    tr.p0 = ld3(p9); tr.p1 = ld3(p9 + 3); tr.p2 = ld3(p9 + 6);
    tr.flip = flip != 0;
    if (n9) { tr.has_n = true; tr.n0 = ld3(n9); tr.n1 = ld3(n9 + 3); tr.n2 = ld3(n9 + 6); }
    if (s9) { tr.has_s = true; tr.s0 = ld3(s9); tr.s1 = ld3(s9 + 3); tr.s2 = ld3(s9 + 6); }
    if (uv6) { tr.has_uv = true; tr.uv0 = v2(uv6[0], uv6[1]); tr.uv1 = v2(uv6[2], uv6[3]); tr.uv2 = v2(uv6[4], uv6[5]); }
    TriangleIntersection ti;
    ti.b0 = b3[0]; ti.b1 = b3[1]; ti.b2 = b3[2]; ti.t = 1.0f;
    SurfaceInteraction si = triangle_interaction(tr, ti, ld3(wo));
    float* o = out;
    auto put3 = [&](V3 v) { *o++ = v.x; *o++ = v.y; *o++ = v.z; };
    put3(v3(si.pi.x.low, si.pi.y.low, si.pi.z.low));
    put3(v3(si.pi.x.high, si.pi.y.high, si.pi.z.high));
    *o++ = si.uv.x; *o++ = si.uv.y;
    put3(si.n); put3(si.dpdu); put3(si.dpdv);
    put3(si.shading.n); put3(si.shading.dpdu); put3(si.shading.dpdv);
    put3(si.dndu); put3(si.dndv); put3(si.shading.dndu); put3(si.shading.dndv);
}
// Sphere::sample_with_context / pdf_with_context (sphere.rs:339-457). out7 = p[3] (interval midpoint), n[3], pdf
int orc_fn_sphere_sample_with_context(const ShmSphere* sp, const float* ctx_p, const float* ctx_n, const float* ctx_ns, const float* u, float* out7) {
    ShapeSampleContext c;
    c.pi = p3i_exact(ld3(ctx_p)); c.n = ld3(ctx_n); c.ns = ld3(ctx_ns);
    ShapeSample ss;
    if (!sphere_sample_with_context(*sp, c, v2(u[0], u[1]), ss)) return 0;
    V3 p = ss.pi.mid();
    out7[0] = p.x; out7[1] = p.y; out7[2] = p.z; out7[3] = ss.n.x; out7[4] = ss.n.y; out7[5] = ss.n.z; out7[6] = ss.pdf;
    return 1;
}
float orc_fn_sphere_pdf_with_context(const ShmSphere* sp, const float* ctx_p, const float* ctx_n, const float* ctx_ns, const float* wi) {
    ShapeSampleContext c;
    c.pi = p3i_exact(ld3(ctx_p)); c.n = ld3(ctx_n); c.ns = ld3(ctx_ns);
    return sphere_pdf_with_context(*sp, c, ld3(wi));
}
// DiffuseAreaLight::l (light.rs:668-684) with a DenselySampledSpectrum l_emit given as a table from lambda_min in 1-nm steps
void orc_fn_area_light_l(int two_sided, float scale, const float* table, int n_table, int lambda_min, const float* n, const float* w,
                         const float* lambda4, float* out4) {
    SceneView sv;
    memset(&sv, 0, sizeof(sv));
    sv.spectrum_data = table;
    ShmLight light;
    memset(&light, 0, sizeof(light));
    light.kind = SHM_LIGHT_DIFFUSE_AREA;
    light.two_sided = two_sided ? 1u : 0u;
    light.scale = scale;
    light.spectrum.kind = SHM_SPECTRUM_DENSE;
    light.spectrum.offset = 0; light.spectrum.n = (uint32_t)n_table; light.spectrum.lambda_min = lambda_min;
    Wavelengths wl;
    for (int i = 0; i < 4; ++i) { wl.lambda[i] = lambda4[i]; wl.pdf[i] = 1.0f; }
    Spec r = area_light_l(sv, light, ld3(n), ld3(w), wl);
    for (int i = 0; i < 4; ++i) out4[i] = r.v[i];
}
// Light::pdf_li of light `li` of a scene (DiffuseAreaLight::pdf_li = shape.pdf_with_context, light.rs:663-666)
float orc_fn_light_pdf_li(OrcScene* s, uint32_t li, const float* ctx_p, const float* ctx_n, const float* ctx_ns, const float* wi) {
    Oracle* o = reinterpret_cast<Oracle*>(s);
    LightSampleContext ctx;
    ctx.pi = p3i_exact(ld3(ctx_p)); ctx.n = ld3(ctx_n); ctx.ns = ld3(ctx_ns);
    return light_pdf_li<false, true>(o->sv, o->sv.lights[li], ctx, ld3(wi));
}
// PixelSensor::to_sensor_rgb + RgbFilm::add_sample (film.rs:548-574, 907-914) on one pixel {rgb_sum[3], weight_sum} (doubles), with the
// sensor given as three 360..=830 nm tables
void orc_fn_film_add_sample(const float* r_bar, const float* g_bar, const float* b_bar, float imaging_ratio, float max_component_value,
                            const float* L4, const float* lambda4, const float* pdf4, float weight, double* pixel4, float* rgb_out3) {
    SceneView sv;
    memset(&sv, 0, sizeof(sv));
    sv.sensor_r_bar = r_bar; sv.sensor_g_bar = g_bar; sv.sensor_b_bar = b_bar;
    sv.imaging_ratio = imaging_ratio; sv.max_component_value = max_component_value;
    Spec L;
    Wavelengths w;
    for (int i = 0; i < 4; ++i) { L.v[i] = L4[i]; w.lambda[i] = lambda4[i]; w.pdf[i] = pdf4[i]; }
    V3 rgb = film_sample_rgb(sv, L, w);
    if (rgb_out3) { rgb_out3[0] = rgb.x; rgb_out3[1] = rgb.y; rgb_out3[2] = rgb.z; }
    pixel4[0] += (double)(weight * rgb.x);
    pixel4[1] += (double)(weight * rgb.y);
    pixel4[2] += (double)(weight * rgb.z);
    pixel4[3] += (double)weight;
}
// PerspectiveCamera::generate_ray_differential (camera.rs:1003-1079) for an explicit CameraSample. out18 = o[3], d[3], rx_o[3],
// rx_d[3], ry_o[3], ry_d[3]
void orc_fn_camera_ray_differential(const ShmCamera* cam, const float* p_film, const float* p_lens, float* out18) {
    AuxRays aux = aux_none();
    Ray r = camera_generate_ray_differential(*cam, v2(p_film[0], p_film[1]), v2(p_lens[0], p_lens[1]), &aux);
    float* o = out18;
    auto put3 = [&](V3 v) { *o++ = v.x; *o++ = v.y; *o++ = v.z; };
    put3(r.o); put3(r.d); put3(aux.rx_o); put3(aux.rx_d); put3(aux.ry_o); put3(aux.ry_d);
}

}  // extern "C"
#pragma GCC visibility pop
