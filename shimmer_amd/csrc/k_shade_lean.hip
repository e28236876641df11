// k_shade_lean.hip — the fused shade kernel of the one scene class that keeps it: triangles only, every material a DiffuseMaterial, no
// textures (the headline scene class). One BxDF class means there is nothing to sort, and the staged pipeline's parameter block would be
// pure HBM traffic; every other scene class runs k_vertex -> k_scatter<class> (k_vertex.inl, k_scatter.inl).
#include "k_shade.inl"

int wf_launch_shade_lean(ShmScene* s, const ShadeArgs& a) {
    // the scene runs this kernel alone: a vertex leaves its hit record, not its LightSampleContext, for the next vertex's emitter MIS weight (k_shade.inl; round 4 A/B: DESIGN.md section 6)
    constexpr int ctx_as_hit = 1;
#define CTX_AS_HIT_FLAG ((ctx_as_hit << 1) | ((ctx_as_hit && a.hit_kept) ? 4 : 0))
    WF_SHADE_LAUNCH((k_shade<false, true, false, true, false>));
#undef CTX_AS_HIT_FLAG
    WF_EMIT_JOBS_LAUNCH(ctx_as_hit);  // emission of the vertices that hit an emitter, before K3 adds this bounce's shadow contributions (k_shade.inl)
    return SHM_OK;
}
// The same kernel over the queue k_vertex diverted plain-diffuse hits to (a scene that also holds other materials): for those vertices the
// fused kernel beats the staged pair — no parameter block to write and read back ("staged shading").
int wf_launch_shade_lean_diverted(ShmScene* s, const ShadeArgs& a) {
    WF_SHADE_LAUNCH_DIVERTED((k_shade<false, true, false, true, false>));
    WF_EMIT_JOBS_LAUNCH(0);
    return SHM_OK;
}
