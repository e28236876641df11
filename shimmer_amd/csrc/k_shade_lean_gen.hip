// k_shade_lean_gen.hip — the fused shade kernel for scenes that hold spheres, bilinear patches or instances beside their triangles (round 5): every material a
// DiffuseMaterial (or: the plain-diffuse hits of a scene with other materials, diverted by k_vertex), no textures. The same template as the triangle scenes' kernel
// (k_shade.inl) with TRI_ONLY = false: the interaction of a sphere / patch / instanced hit and the light samples of sphere / patch emitters are compiled in — 168 VGPRs, three
// waves per SIMD, 9 spilled (the triangle instantiation: 158, none). Hit records are the 32-byte ShmHit (t, phi, instance); a vertex leaves its LightSampleContext
// for the next vertex's emitter MIS weight (the 16-byte hit-record forms of the triangle class need triangle hits).
#include "k_shade.inl"

int wf_launch_shade_lean_gen(ShmScene* s, const ShadeArgs& a) {
#define CTX_AS_HIT_FLAG 0
    WF_SHADE_LAUNCH((k_shade<false, false, false, true, false>));
#undef CTX_AS_HIT_FLAG
    hipLaunchKernelGGL((k_emit_jobs<false, false>), dim3(s->n_cu * 4), dim3(SHADE2_BLOCK), 0, a.stream, s->dsv, s->pa, s->d_q_emit, s->d_qs, 0);
    LAUNCH_TRY("k_emit_jobs");
    return SHM_OK;
}
int wf_launch_shade_lean_gen_diverted(ShmScene* s, const ShadeArgs& a) {
    WF_SHADE_LAUNCH_DIVERTED((k_shade<false, false, false, true, false>));
    hipLaunchKernelGGL((k_emit_jobs<false, false>), dim3(s->n_cu * 4), dim3(SHADE2_BLOCK), 0, a.stream, s->dsv, s->pa, s->d_q_emit, s->d_qs, 0);
    LAUNCH_TRY("k_emit_jobs");
    return SHM_OK;
}
