// k_scatter_dielectric_env.hip — k_scatter_dielectric.hip for scenes whose only image is an ImageInfinitelight (K_ENV_LIGHT, k_scatter.inl; k_vertex_env.hip says why).
// (options.force_diffuse renders of such a scene run the textured class's kernels: render.hip)
#define K_ENV_LIGHT true
#include "k_scatter.inl"

int wf_launch_scatter_dielectric_env(ShmScene* s, const ShadeArgs& a, bool tri_only) {
    if (a.params.regularize != 0) {
        if (tri_only) WF_SCATTER_LAUNCH(CLASS_DIELECTRIC, true, false);
        else WF_SCATTER_LAUNCH(CLASS_DIELECTRIC, false, false);
        return SHM_OK;
    }
    const int spec_blocks = s->n_cu * 4;
    if (tri_only) WF_SCATTER_LAUNCH_SUB(k_scatter_specular, spec_blocks, CLASS_DIELECTRIC, true, false);
    else WF_SCATTER_LAUNCH_SUB(k_scatter_specular, spec_blocks, CLASS_DIELECTRIC, false, false);
    if (s->flat.has_rough_dielectric) {
        if (tri_only) WF_SCATTER_LAUNCH_SUB(k_scatter_nonspecular, a.blocks, CLASS_DIELECTRIC, true, false);
        else WF_SCATTER_LAUNCH_SUB(k_scatter_nonspecular, a.blocks, CLASS_DIELECTRIC, false, false);
    }
    return SHM_OK;
}
