// k_shade_tail_sorted_env.hip — the material-sorted fused all-materials kernel (k_shade_tail_sorted.hip) for triangle scenes whose only image is an ImageInfinitelight:
// k_shade.inl <false, TRI_ONLY = true, HAS_TEX = false, false, true, SORT_CHUNK = true, ENV_LIGHT = true> — glass and metal under an environment map without the textured
// class's ray differentials and auxiliary rays (k_shade_lean_env.hip says why).
#ifndef K_SHADE_LEAN_WAVES
#define K_SHADE_LEAN_WAVES 3
#endif
#include "k_shade.inl"

int wf_launch_shade_tail_sorted_env(ShmScene* s, const ShadeArgs& a) {
#define CTX_AS_HIT_FLAG 0
    WF_SHADE_LAUNCH((k_shade<false, true, false, false, true, true, true>));
#undef CTX_AS_HIT_FLAG
    return SHM_OK;
}
