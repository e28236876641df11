// k_scatter_conductor_env.hip — k_scatter_conductor.hip for scenes whose only image is an ImageInfinitelight (K_ENV_LIGHT, k_scatter.inl; k_vertex_env.hip says why).
#define K_ENV_LIGHT true
#include "k_scatter.inl"

int wf_launch_scatter_conductor_env(ShmScene* s, const ShadeArgs& a, bool tri_only) {
    if (tri_only) WF_SCATTER_LAUNCH(CLASS_CONDUCTOR, true, false);
    else WF_SCATTER_LAUNCH(CLASS_CONDUCTOR, false, false);
    return SHM_OK;
}
