// k_shade_fused_gen.hip — the fused all-materials vertex kernel with material-sorted chunks (k_shade_tail_sorted.hip) for scenes that hold spheres, bilinear patches or
// instances and more than the diffuse BxDF class, without textures or coated materials: k_shade.inl <HAS_LAYERED = false, TRI_ONLY = false, HAS_TEX = false,
// DIFFUSE_ONLY = false, EMIT_INLINE = true, SORT_CHUNK = true> (247 VGPRs, no spill, two waves per SIMD).
#ifndef K_SHADE_LEAN_WAVES
#define K_SHADE_LEAN_WAVES 2
#endif
#include "k_shade.inl"

int wf_launch_shade_fused_gen(ShmScene* s, const ShadeArgs& a) {
#define CTX_AS_HIT_FLAG 0
    WF_SHADE_LAUNCH((k_shade<false, false, false, false, true, true>));
#undef CTX_AS_HIT_FLAG
    return SHM_OK;
}
