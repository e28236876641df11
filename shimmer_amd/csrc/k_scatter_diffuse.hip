// k_scatter_diffuse.hip — the scattering half of a vertex (k_scatter.inl) for the CLASS_DIFFUSE queue, in the three scene classes.
#include "k_scatter.inl"

int wf_launch_scatter_diffuse(ShmScene* s, const ShadeArgs& a, bool tri_only, bool has_tex) {
    WF_SCATTER_DISPATCH(CLASS_DIFFUSE);
    return SHM_OK;
}
