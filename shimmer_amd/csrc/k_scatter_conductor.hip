// k_scatter_conductor.hip — the scattering half of a vertex (k_scatter.inl) for the CLASS_CONDUCTOR queue, in the three scene classes.
#include "k_scatter.inl"

int wf_launch_scatter_conductor(ShmScene* s, const ShadeArgs& a, bool tri_only, bool has_tex) {
    WF_SCATTER_DISPATCH(CLASS_CONDUCTOR);
    return SHM_OK;
}
