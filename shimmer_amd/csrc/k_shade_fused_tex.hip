// k_shade_fused_tex.hip — the fused all-materials vertex kernel with material-sorted chunks (k_shade_tail_sorted.hip) for triangle scenes that bind textures:
// k_shade.inl <HAS_LAYERED = false, TRI_ONLY = true, HAS_TEX = true, DIFFUSE_ONLY = false, EMIT_INLINE = true, SORT_CHUNK = true>. Round 5 found the sorted fused
// kernel ahead of the staged pair (k_vertex + one scatter kernel per BxDF class) at EVERY bounce of the crown proxy; this is the same question for the textured class.
#ifndef K_SHADE_LEAN_WAVES
#define K_SHADE_LEAN_WAVES 2
#endif
#include "k_shade.inl"

int wf_launch_shade_fused_tex(ShmScene* s, const ShadeArgs& a) {
#define CTX_AS_HIT_FLAG 0
    WF_SHADE_LAUNCH((k_shade<false, true, true, false, true, true>));
#undef CTX_AS_HIT_FLAG
    return SHM_OK;
}
