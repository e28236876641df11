// k_shade_fused_gen_env.hip — k_shade_tail_sorted_env.hip for scenes that hold spheres, bilinear patches or instances: k_shade.inl <false, TRI_ONLY = false, false, false, true,
// SORT_CHUNK = true, ENV_LIGHT = true>.
#ifndef K_SHADE_LEAN_WAVES
#define K_SHADE_LEAN_WAVES 2
#endif
#include "k_shade.inl"

int wf_launch_shade_fused_gen_env(ShmScene* s, const ShadeArgs& a) {
#define CTX_AS_HIT_FLAG 0
    WF_SHADE_LAUNCH((k_shade<false, false, false, false, true, true, true>));
#undef CTX_AS_HIT_FLAG
    return SHM_OK;
}
