// shm/probe.h — one leaf function of the shared arithmetic evaluated on flat arguments: TEST INFRASTRUCTURE of the product library (shm_debug_eval_leaf, k_leaf_probe.hip).
// The `-m gpu` suite replays the committed golden vectors — the reference's own in-source known answers and the independent float32 / float64 re-evaluations of
// tests/golden/*.json — through the DEVICE code of these functions and compares with the committed expected values directly (tests/test_gpu_leaf_replay.py): GPU == vectors,
// not GPU == oracle == vectors. Arguments and results are 32-bit words (floats by their bits; small integers as integers; POD structs as their bytes; doubles as two words).
// Each case cites the function it evaluates; the wrappers take the same arguments as the oracle's test entry points of the same name (oracle/oracle.cpp: orc_fn_*), which
// is what lets the GPU tests call the CPU tests' bodies with a device-backed object in the oracle library's place.
#pragma once
#include "path.h"

namespace shm {

enum : int {
    PROBE_NEXT_FLOAT_UP = 1, PROBE_NEXT_FLOAT_DOWN, PROBE_GAMMA, PROBE_DIFFERENCE_OF_PRODUCTS, PROBE_DOT, PROBE_CROSS, PROBE_COORDINATE_SYSTEM, PROBE_HYPOT, PROBE_ROUND,
    PROBE_ATAN2, PROBE_INTERSECT_P_CACHED, PROBE_INTERSECT_TRIANGLE, PROBE_TR_D, PROBE_TR_G, PROBE_TR_LAMBDA, PROBE_TR_SAMPLE_WM, PROBE_FRESNEL_DIELECTRIC,
    PROBE_FRESNEL_COMPLEX, PROBE_BXDF_SAMPLE_F, PROBE_BXDF_F_PDF, PROBE_LAYERED_F_PDF, PROBE_LAYERED_SAMPLE_F, PROBE_OFFSET_RAY_ORIGIN, PROBE_TRIANGLE_SAMPLE_WITH_CONTEXT,
    PROBE_TRIANGLE_PDF_WITH_CONTEXT, PROBE_TRIANGLE_INTERACTION, PROBE_SPHERE_SAMPLE_WITH_CONTEXT, PROBE_SPHERE_PDF_WITH_CONTEXT, PROBE_AREA_LIGHT_L, PROBE_FILM_ADD_SAMPLE,
    PROBE_CAMERA_RAY_DIFFERENTIAL, PROBE_INTERVAL_OP, PROBE_DET3, PROBE_ROTATE_FROM_TO, PROBE_SAMPLE_DISCRETE, PROBE_SAMPLER_STREAM, PROBE_SAMPLE_VISIBLE_WAVELENGTHS,
    PROBE_VISIBLE_WAVELENGTHS_PDF, PROBE_VECMATH, PROBE_TRANSFORM_APPLY, PROBE_BLP_INTERSECT, PROBE_BLP_SAMPLE_WITH_CONTEXT, PROBE_BLP_PDF_WITH_CONTEXT, PROBE_SPHERE_INTERSECT,
    PROBE_UNARY, PROBE_EQUAL_AREA_SQUARE_TO_SPHERE, PROBE_EQUAL_AREA_SPHERE_TO_SQUARE, PROBE_N_OPS
};

namespace probe_detail {
SHM_HD Float f(const uint32_t* in, int i) { return bits_to_float(in[i]); }
SHM_HD V3 f3(const uint32_t* in, int i) { return v3(bits_to_float(in[i]), bits_to_float(in[i + 1]), bits_to_float(in[i + 2])); }
SHM_HD void put(uint32_t* out, int i, Float v) { out[i] = float_to_bits(v); }
SHM_HD void put3(uint32_t* out, int i, V3 v) { put(out, i, v.x); put(out, i + 1, v.y); put(out, i + 2, v.z); }
SHM_HD BxDF plain_bxdf(const uint32_t* in) {  // kind, r[4], k[4], eta, ax, ay: 12 words
    BxDF b;
    b.strict = 0;
    b.kind = in[0];
    for (int i = 0; i < 4; ++i) { b.r.v[i] = f(in, 1 + i); b.k.v[i] = f(in, 5 + i); }
    b.eta = f(in, 9);
    b.mf = trowbridge_reitz_new(f(in, 10), f(in, 11));
    return b;
}
SHM_HD BxDF layered_bxdf(const uint32_t* in) {  // kind, p[19], max_depth, n_samples: 22 words (oracle.cpp: make_layered)
    BxDF b;
    const int ns = (int)in[21];
    b.strict = ns < 0 ? 1 : 0;
    b.kind = in[0];
    for (int i = 0; i < 4; ++i) { b.r.v[i] = f(in, 1 + i); b.k.v[i] = f(in, 5 + i); b.albedo.v[i] = f(in, 9 + i); }
    b.eta = f(in, 13);
    b.mf = trowbridge_reitz_new(f(in, 14), f(in, 15));
    b.mf2 = trowbridge_reitz_new(f(in, 16), f(in, 17));
    b.thickness = f(in, 18);
    b.g = f(in, 19);
    b.max_depth = (int)in[20];
    b.n_samples = ns < 0 ? -ns : ns;
    return b;
}
SHM_HD ShapeSampleContext shape_ctx(const uint32_t* in, int i) {  // ctx_p, ctx_n, ctx_ns: 9 words
    ShapeSampleContext c;
    c.pi = p3i_exact(f3(in, i)); c.n = f3(in, i + 3); c.ns = f3(in, i + 6);
    return c;
}
SHM_HD void put_shape_sample(uint32_t* out, const ShapeSample& ss) {
    put3(out, 0, ss.pi.mid()); put3(out, 3, ss.n); put(out, 6, ss.pdf);
}
SHM_HD PatchData patch_of(const uint32_t* in, int i, int flip) {  // oracle.cpp: make_patch
    PatchData pd;
    pd.p00 = f3(in, i); pd.p10 = f3(in, i + 3); pd.p01 = f3(in, i + 6); pd.p11 = f3(in, i + 9);
    pd.flip = flip != 0;
    pd.is_rect = blp_is_rectangle(pd.p00, pd.p10, pd.p01, pd.p11);
    pd.area = blp_area(pd.p00, pd.p10, pd.p01, pd.p11, pd.is_rect);
    pd.has_n = pd.has_uv = false;
    pd.n00 = pd.n10 = pd.n01 = pd.n11 = v3s(0.0f);
    pd.uv00 = pd.uv10 = pd.uv01 = pd.uv11 = v2(0.0f, 0.0f);
    return pd;
}
}  // namespace probe_detail

// Returns the wrapped function's integer result (1 / 0 for the Option-returning ones, 0 otherwise); results go to `out`.
SHM_HD int leaf_probe(int op, const uint32_t* in, uint32_t* out) {
    using namespace probe_detail;
    switch (op) {
        case PROBE_NEXT_FLOAT_UP: put(out, 0, next_float_up(f(in, 0))); return 0;                       // float.rs:53-70
        case PROBE_NEXT_FLOAT_DOWN: put(out, 0, next_float_down(f(in, 0))); return 0;                   // float.rs:72-90
        case PROBE_GAMMA: put(out, 0, gamma((int)in[0])); return 0;                                      // float.rs:41-43
        case PROBE_DIFFERENCE_OF_PRODUCTS: put(out, 0, difference_of_products(f(in, 0), f(in, 1), f(in, 2), f(in, 3))); return 0;  // math.rs:170-176
        case PROBE_DOT: put(out, 0, dot(f3(in, 0), f3(in, 3))); return 0;
        case PROBE_CROSS: put3(out, 0, cross(f3(in, 0), f3(in, 3))); return 0;
        case PROBE_COORDINATE_SYSTEM: { V3 a, b; coordinate_system(f3(in, 0), a, b); put3(out, 0, a); put3(out, 3, b); return 0; }
        case PROBE_HYPOT: put(out, 0, shm::hypot(f(in, 0), f(in, 1))); return 0;
        case PROBE_ROUND: put(out, 0, shm::round(f(in, 0))); return 0;
        case PROBE_ATAN2: put(out, 0, shm::atan2(f(in, 0), f(in, 1))); return 0;
        case PROBE_INTERSECT_P_CACHED: {  // bounding_box.rs:520-563; bmin[3], bmax[3], o[3], d[3], t_max
            const V3 rd = f3(in, 9);
            const V3 inv = v3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
            const int neg[3] = {inv.x < 0.0f, inv.y < 0.0f, inv.z < 0.0f};
            const Float bmin[3] = {f(in, 0), f(in, 1), f(in, 2)}, bmax[3] = {f(in, 3), f(in, 4), f(in, 5)};
            return intersect_p_cached(bmin, bmax, f3(in, 6), f(in, 12), inv, neg) ? 1 : 0;
        }
        case PROBE_INTERSECT_TRIANGLE: {  // triangle.rs:173-302; o, d, t_max, p0, p1, p2
            TriangleIntersection ti;
            if (!intersect_triangle(f3(in, 0), f3(in, 3), f(in, 6), f3(in, 7), f3(in, 10), f3(in, 13), ti)) return 0;
            put(out, 0, ti.b0); put(out, 1, ti.b1); put(out, 2, ti.b2); put(out, 3, ti.t);
            return 1;
        }
        case PROBE_TR_D: put(out, 0, trowbridge_reitz_new(f(in, 0), f(in, 1)).d(f3(in, 2))); return 0;               // scattering.rs:128-147
        case PROBE_TR_G: put(out, 0, trowbridge_reitz_new(f(in, 0), f(in, 1)).g(f3(in, 2), f3(in, 5))); return 0;
        case PROBE_TR_LAMBDA: put(out, 0, trowbridge_reitz_new(f(in, 0), f(in, 1)).lambda(f3(in, 2))); return 0;
        case PROBE_TR_SAMPLE_WM: put3(out, 0, trowbridge_reitz_new(f(in, 0), f(in, 1)).sample_wm(f3(in, 2), v2(f(in, 5), f(in, 6)))); return 0;
        case PROBE_FRESNEL_DIELECTRIC: put(out, 0, fresnel_dielectric(f(in, 0), f(in, 1))); return 0;                 // scattering.rs:49-76
        case PROBE_FRESNEL_COMPLEX: put(out, 0, fresnel_complex(f(in, 0), cx(f(in, 1), f(in, 2)))); return 0;         // scattering.rs:78-89
        case PROBE_BXDF_SAMPLE_F: {  // bxdf.rs: Diffuse / Conductor / Dielectric sample_f; bxdf[12], wo[3], uc, u[2]
            const BxDF b = plain_bxdf(in);
            BSDFSample bs;
            if (!bxdf_sample_f(b, f3(in, 12), f(in, 15), v2(f(in, 16), f(in, 17)), REFLTRANS_ALL, bs)) return 0;
            for (int i = 0; i < 4; ++i) put(out, i, bs.f.v[i]);
            put3(out, 4, bs.wi); put(out, 7, bs.pdf); put(out, 8, (Float)bs.flags); put(out, 9, bs.eta);
            return 1;
        }
        case PROBE_BXDF_F_PDF: {  // bxdf[12], wo[3], wi[3]
            const BxDF b = plain_bxdf(in);
            const Spec r = bxdf_f(b, f3(in, 12), f3(in, 15));
            for (int i = 0; i < 4; ++i) put(out, i, r.v[i]);
            put(out, 4, bxdf_pdf(b, f3(in, 12), f3(in, 15), REFLTRANS_ALL));
            return 0;
        }
        case PROBE_LAYERED_F_PDF: {  // bxdf.rs:883-1620; layered[22], wo[3], wi[3]
            const BxDF b = layered_bxdf(in);
            const Spec r = bxdf_f(b, f3(in, 22), f3(in, 25));
            for (int i = 0; i < 4; ++i) put(out, i, r.v[i]);
            put(out, 4, bxdf_pdf(b, f3(in, 22), f3(in, 25), REFLTRANS_ALL));
            put(out, 5, (Float)bxdf_flags(b));
            return 0;
        }
        case PROBE_LAYERED_SAMPLE_F: {  // layered[22], wo[3], uc, u[2]
            const BxDF b = layered_bxdf(in);
            BSDFSample bs;
            if (!bxdf_sample_f(b, f3(in, 22), f(in, 25), v2(f(in, 26), f(in, 27)), REFLTRANS_ALL, bs)) return 0;
            for (int i = 0; i < 4; ++i) put(out, i, bs.f.v[i]);
            put3(out, 4, bs.wi); put(out, 7, bs.pdf); put(out, 8, (Float)bs.flags); put(out, 9, bs.pdf_is_proportional ? 1.0f : 0.0f);
            return 1;
        }
        case PROBE_OFFSET_RAY_ORIGIN:  // interaction.rs:68-75 -> ray.rs:53-71; p, err, n, w
            put3(out, 0, offset_ray_origin(p3i_from_value_and_error(f3(in, 0), f3(in, 3)), f3(in, 6), f3(in, 9)));
            return 0;
        case PROBE_TRIANGLE_SAMPLE_WITH_CONTEXT: {  // triangle.rs:595-694; p0, p1, p2, ctx[9], u[2]
            TriangleData tr;
            memset(&tr, 0, sizeof(tr));
            tr.p0 = f3(in, 0); tr.p1 = f3(in, 3); tr.p2 = f3(in, 6);
            ShapeSample ss;
            if (!triangle_sample_with_context(tr, shape_ctx(in, 9), v2(f(in, 18), f(in, 19)), ss)) return 0;
            put_shape_sample(out, ss);
            return 1;
        }
        case PROBE_TRIANGLE_PDF_WITH_CONTEXT: {  // triangle.rs:696-745; p0, p1, p2, ctx[9], wi
            TriangleData tr;
            memset(&tr, 0, sizeof(tr));
            tr.p0 = f3(in, 0); tr.p1 = f3(in, 3); tr.p2 = f3(in, 6);
            put(out, 0, triangle_pdf_with_context(tr, shape_ctx(in, 9), f3(in, 18)));
            return 0;
        }
        case PROBE_TRIANGLE_INTERACTION: {  // triangle.rs:305-504; p9, has_n, n9, has_s, s9, has_uv, uv6, flip, b3, wo
            TriangleData tr;
            memset(&tr, 0, sizeof(tr));
            tr.p0 = f3(in, 0); tr.p1 = f3(in, 3); tr.p2 = f3(in, 6);
            if (in[9]) { tr.has_n = true; tr.n0 = f3(in, 10); tr.n1 = f3(in, 13); tr.n2 = f3(in, 16); }
            if (in[19]) { tr.has_s = true; tr.s0 = f3(in, 20); tr.s1 = f3(in, 23); tr.s2 = f3(in, 26); }
            if (in[29]) { tr.has_uv = true; tr.uv0 = v2(f(in, 30), f(in, 31)); tr.uv1 = v2(f(in, 32), f(in, 33)); tr.uv2 = v2(f(in, 34), f(in, 35)); }
            tr.flip = in[36] != 0;
            TriangleIntersection ti;
            ti.b0 = f(in, 37); ti.b1 = f(in, 38); ti.b2 = f(in, 39); ti.t = 1.0f;
            const SurfaceInteraction si = triangle_interaction(tr, ti, f3(in, 40));
            put3(out, 0, v3(si.pi.x.low, si.pi.y.low, si.pi.z.low));
            put3(out, 3, v3(si.pi.x.high, si.pi.y.high, si.pi.z.high));
            put(out, 6, si.uv.x); put(out, 7, si.uv.y);
            put3(out, 8, si.n); put3(out, 11, si.dpdu); put3(out, 14, si.dpdv);
            put3(out, 17, si.shading.n); put3(out, 20, si.shading.dpdu); put3(out, 23, si.shading.dpdv);
            put3(out, 26, si.dndu); put3(out, 29, si.dndv); put3(out, 32, si.shading.dndu); put3(out, 35, si.shading.dndv);
            return 0;
        }
        case PROBE_SPHERE_SAMPLE_WITH_CONTEXT: {  // sphere.rs:339-420; ctx[9], u[2], then the ShmSphere's bytes
            ShapeSample ss;
            if (!sphere_sample_with_context(*reinterpret_cast<const ShmSphere*>(in + 11), shape_ctx(in, 0), v2(f(in, 9), f(in, 10)), ss)) return 0;
            put_shape_sample(out, ss);
            return 1;
        }
        case PROBE_SPHERE_PDF_WITH_CONTEXT:  // sphere.rs:422-457; ctx[9], wi[3], sphere
            put(out, 0, sphere_pdf_with_context(*reinterpret_cast<const ShmSphere*>(in + 12), shape_ctx(in, 0), f3(in, 9)));
            return 0;
        case PROBE_AREA_LIGHT_L: {  // light.rs:668-684; two_sided, scale, n_table, lambda_min, n[3], w[3], lambda[4], table[n_table]
            SceneView sv;
            memset(&sv, 0, sizeof(sv));
            sv.spectrum_data = reinterpret_cast<const Float*>(in + 14);
            ShmLight light;
            memset(&light, 0, sizeof(light));
            light.kind = SHM_LIGHT_DIFFUSE_AREA;
            light.two_sided = in[0] ? 1u : 0u;
            light.scale = f(in, 1);
            light.spectrum.kind = SHM_SPECTRUM_DENSE;
            light.spectrum.offset = 0; light.spectrum.n = in[2]; light.spectrum.lambda_min = (int)in[3];
            Wavelengths wl;
            for (int i = 0; i < 4; ++i) { wl.lambda[i] = f(in, 10 + i); wl.pdf[i] = 1.0f; }
            const Spec r = area_light_l(sv, light, f3(in, 4), f3(in, 7), wl);
            for (int i = 0; i < 4; ++i) put(out, i, r.v[i]);
            return 0;
        }
        case PROBE_FILM_ADD_SAMPLE: {  // film.rs:548-574, 907-914; imaging_ratio, max_component_value, L[4], lambda[4], pdf[4], weight, pixel (4 doubles), r_bar / g_bar / b_bar [471 each]
            SceneView sv;
            memset(&sv, 0, sizeof(sv));
            sv.sensor_r_bar = reinterpret_cast<const Float*>(in + 23);
            sv.sensor_g_bar = sv.sensor_r_bar + 471;
            sv.sensor_b_bar = sv.sensor_g_bar + 471;
            sv.imaging_ratio = f(in, 0); sv.max_component_value = f(in, 1);
            Spec L;
            Wavelengths w;
            for (int i = 0; i < 4; ++i) { L.v[i] = f(in, 2 + i); w.lambda[i] = f(in, 6 + i); w.pdf[i] = f(in, 10 + i); }
            const Float weight = f(in, 14);
            double px[4];
            memcpy(px, in + 15, sizeof(px));
            const V3 rgb = film_sample_rgb(sv, L, w);
            px[0] += (double)(weight * rgb.x); px[1] += (double)(weight * rgb.y); px[2] += (double)(weight * rgb.z); px[3] += (double)weight;
            memcpy(out, px, sizeof(px));
            put3(out, 8, rgb);
            return 0;
        }
        case PROBE_CAMERA_RAY_DIFFERENTIAL: {  // camera.rs:1003-1079; p_film[2], p_lens[2], then the ShmCamera's bytes
            AuxRays aux = aux_none();
            const Ray r = camera_generate_ray_differential(*reinterpret_cast<const ShmCamera*>(in + 4), v2(f(in, 0), f(in, 1)), v2(f(in, 2), f(in, 3)), &aux);
            put3(out, 0, r.o); put3(out, 3, r.d); put3(out, 6, aux.rx_o); put3(out, 9, aux.rx_d); put3(out, 12, aux.ry_o); put3(out, 15, aux.ry_d);
            return 0;
        }
        case PROBE_INTERVAL_OP: {  // interval.rs:366-414; op, a.low, a.high, b.low, b.high
            const Interval a = iv_new(f(in, 1), f(in, 2)), b = iv_new(f(in, 3), f(in, 4));
            Interval r = a;
            switch ((int)in[0]) {
                case 0: r = a * b; break;
                case 1: r = a / b; break;
                case 2: r = a + b; break;
                case 3: r = a - b; break;
                case 4: r = iv_sqr(a); break;
                default: r = iv_sqrt(a); break;
            }
            put(out, 0, r.low); put(out, 1, r.high);
            return 0;
        }
        case PROBE_DET3: put(out, 0, det3(f3(in, 0), f3(in, 3), f3(in, 6))); return 0;  // square_matrix.rs:281-292
        case PROBE_ROTATE_FROM_TO: put3(out, 0, rot3_apply(rotate_from_to(f3(in, 0), f3(in, 3)), f3(in, 6))); return 0;  // transform.rs:305-361
        case PROBE_SAMPLE_DISCRETE: {  // sampling.rs:196-240; n, u, weights[n]
            Float pmf = 0.0f, ur = 0.0f;
            const int r = sample_discrete(reinterpret_cast<const Float*>(in + 2), (int)in[0], f(in, 1), &pmf, &ur);
            put(out, 0, pmf); put(out, 1, ur);
            return r;
        }
        case PROBE_SAMPLER_STREAM: {  // the defined per-pixel stream (shm/sampling.h); px, py, sample_index, seed lo, seed hi, n
            Rng r = sampler_start_pixel_sample((int)in[0], (int)in[1], (int)in[2], (uint64_t)in[3] | ((uint64_t)in[4] << 32));
            for (int i = 0; i < (int)in[5]; ++i) put(out, i, sampler_get_1d(r));
            return 0;
        }
        case PROBE_SAMPLE_VISIBLE_WAVELENGTHS: put(out, 0, sample_visible_wavelengths(f(in, 0))); return 0;  // sampling.rs:347-371
        case PROBE_VISIBLE_WAVELENGTHS_PDF: put(out, 0, visible_wavelengths_pdf(f(in, 0))); return 0;
        case PROBE_VECMATH: {  // oracle.cpp: orc_fn_vecmath
            const V3 va = f3(in, 0), vb = f3(in, 3);
            put(out, 0, length(va)); put(out, 1, length_squared(va));
            put(out, 2, angle_between(normalize(va), normalize(vb)));
            const V3 n = normalize(va);
            put3(out, 3, n);
            put3(out, 6, gram_schmidt(vb, n));
            return 0;
        }
        case PROBE_TRANSFORM_APPLY: {  // transform.rs:363-383, 606-629; kind, inverse, m[16], m_inv[16], v[3]
            const Float* m = reinterpret_cast<const Float*>(in + 2);
            const Float* m_inv = m + 16;
            const Float* fwd = in[1] ? m_inv : m;
            const Float* bwd = in[1] ? m : m_inv;
            const V3 v = f3(in, 34);
            put3(out, 0, in[0] == 0 ? xf_point(fwd, v) : (in[0] == 1 ? xf_vector(fwd, v) : xf_normal(bwd, v)));
            return 0;
        }
        case PROBE_BLP_INTERSECT: {  // bilinear_patch.rs:144-236; pts[12], o, d, t_max
            BilinearIntersection bi;
            if (!blp_intersect(f3(in, 12), f3(in, 15), f(in, 18), f3(in, 0), f3(in, 3), f3(in, 6), f3(in, 9), bi)) return 0;
            put(out, 0, bi.u); put(out, 1, bi.v); put(out, 2, bi.t);
            return 1;
        }
        case PROBE_BLP_SAMPLE_WITH_CONTEXT: {  // bilinear_patch.rs:440-560; pts[12], flip, ctx[9], u[2]
            ShapeSample ss;
            if (!blp_sample_with_context(patch_of(in, 0, (int)in[12]), shape_ctx(in, 13), v2(f(in, 22), f(in, 23)), ss)) return 0;
            put_shape_sample(out, ss);
            return 1;
        }
        case PROBE_BLP_PDF_WITH_CONTEXT:  // pts[12], flip, ctx[9], wi
            put(out, 0, blp_pdf_with_context(patch_of(in, 0, (int)in[12]), shape_ctx(in, 13), f3(in, 22)));
            return 0;
        case PROBE_SPHERE_INTERSECT: {  // sphere.rs:95-196; o, d, t_max, sphere
            QuadricIntersection qi;
            if (!sphere_basic_intersect(*reinterpret_cast<const ShmSphere*>(in + 7), f3(in, 0), f3(in, 3), f(in, 6), qi)) return 0;
            put(out, 0, qi.t_hit); put3(out, 1, qi.p_obj); put(out, 4, qi.phi);
            return 1;
        }
        case PROBE_UNARY: {  // fp.h's own transcendentals (the parity contract trusts hardware with + - * / sqrt fma only); which, x
            const Float x = f(in, 1);
            Float r = 0.0f;
            switch ((int)in[0]) {
                case 0: r = shm::sin(x); break;
                case 1: r = shm::cos(x); break;
                case 2: r = shm::asin(x); break;
                case 3: r = shm::acos(x); break;
                case 4: r = shm::exp(x); break;
                case 5: r = shm::log(x); break;
                case 6: r = shm::atanh(x); break;
                case 7: r = shm::cosh(x); break;
                default: r = shm::log2(x); break;
            }
            put(out, 0, r);
            return 0;
        }
        // the equal-area octahedral mapping of ImageInfinitelight (math.rs:456-525): the direction of a map texel and back — what the ENV_LIGHT instantiations' look-up, sample
        // and pdf start from
        case PROBE_EQUAL_AREA_SQUARE_TO_SPHERE: put3(out, 0, equal_area_square_to_sphere(v2(f(in, 0), f(in, 1)))); return 0;
        case PROBE_EQUAL_AREA_SPHERE_TO_SQUARE: { const V2 p = equal_area_sphere_to_square(f3(in, 0)); put(out, 0, p.x); put(out, 1, p.y); return 0; }
        default: return -1;
    }
}

}  // namespace shm
