// k_scatter_dielectric.hip — the scattering half of a vertex (k_scatter.inl) for the CLASS_DIELECTRIC queue, in the three scene classes.
#include "k_scatter.inl"

int wf_launch_scatter_dielectric(ShmScene* s, const ShadeArgs& a, bool tri_only, bool has_tex) {
    WF_SCATTER_DISPATCH(CLASS_DIELECTRIC);
    return SHM_OK;
}
