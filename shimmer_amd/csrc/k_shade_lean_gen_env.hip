// k_shade_lean_gen_env.hip — k_shade_lean_env.hip for scenes that hold spheres, bilinear patches or instances: k_shade.inl <false, TRI_ONLY = false, false, true, false, false,
// ENV_LIGHT = true>.
#include "k_shade.inl"

int wf_launch_shade_lean_gen_env(ShmScene* s, const ShadeArgs& a) {
#define CTX_AS_HIT_FLAG 0
    WF_SHADE_LAUNCH((k_shade<false, false, false, true, false, false, true>));
#undef CTX_AS_HIT_FLAG
    hipLaunchKernelGGL((k_emit_jobs<false, false>), dim3(s->n_cu * 4), dim3(SHADE2_BLOCK), 0, a.stream, s->dsv, s->pa, s->d_q_emit, s->d_qs, 0);
    LAUNCH_TRY("k_emit_jobs");
    return SHM_OK;
}
int wf_launch_shade_lean_gen_env_diverted(ShmScene* s, const ShadeArgs& a) {
    WF_SHADE_LAUNCH_DIVERTED((k_shade<false, false, false, true, false, false, true>));
    hipLaunchKernelGGL((k_emit_jobs<false, false>), dim3(s->n_cu * 4), dim3(SHADE2_BLOCK), 0, a.stream, s->dsv, s->pa, s->d_q_emit, s->d_qs, 0);
    LAUNCH_TRY("k_emit_jobs");
    return SHM_OK;
}
