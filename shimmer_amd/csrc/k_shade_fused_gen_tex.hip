// k_shade_fused_gen_tex.hip — the fused all-materials vertex kernel with material-sorted chunks for scenes of any shape that bind textures (image textures, image
// infinite lights), without coated materials: k_shade.inl <HAS_LAYERED = false, TRI_ONLY = false, HAS_TEX = true, DIFFUSE_ONLY = false, EMIT_INLINE = true,
// SORT_CHUNK = true>.
#ifndef K_SHADE_LEAN_WAVES
#define K_SHADE_LEAN_WAVES 2
#endif
#include "k_shade.inl"

int wf_launch_shade_fused_gen_tex(ShmScene* s, const ShadeArgs& a) {
#define CTX_AS_HIT_FLAG 0
    WF_SHADE_LAUNCH((k_shade<false, false, true, false, true, true>));
#undef CTX_AS_HIT_FLAG
    return SHM_OK;
}
