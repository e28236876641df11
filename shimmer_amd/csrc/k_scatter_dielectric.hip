// k_scatter_dielectric.hip — the scattering half of a vertex (k_scatter.inl) for the CLASS_DIELECTRIC queue, in the three scene classes.
#include "k_scatter.inl"

// Without options.force_diffuse / regularize (which change the BxDF inside the scatter half) the class queue is worked through by TWO kernels: the specular
// entries — smooth DielectricBxDF, ThinDielectricBxDF: no NEE, no microfacet code — at four waves per SIMD, and the rough ones (launched only if the
// material table holds a dielectric that can be rough) by the general kernel; each skips the other's entries.
int wf_launch_scatter_dielectric(ShmScene* s, const ShadeArgs& a, bool tri_only, bool has_tex) {
    if (a.params.force_diffuse != 0 || a.params.regularize != 0) {
        WF_SCATTER_DISPATCH(CLASS_DIELECTRIC);
        return SHM_OK;
    }
    const int spec_blocks = s->n_cu * 4;
    if (has_tex) WF_SCATTER_LAUNCH_SUB(k_scatter_specular, spec_blocks, CLASS_DIELECTRIC, false, true);
    else if (tri_only) WF_SCATTER_LAUNCH_SUB(k_scatter_specular, spec_blocks, CLASS_DIELECTRIC, true, false);
    else WF_SCATTER_LAUNCH_SUB(k_scatter_specular, spec_blocks, CLASS_DIELECTRIC, false, false);
    if (s->flat.has_rough_dielectric) {
        if (has_tex) WF_SCATTER_LAUNCH_SUB(k_scatter_nonspecular, a.blocks, CLASS_DIELECTRIC, false, true);
        else if (tri_only) WF_SCATTER_LAUNCH_SUB(k_scatter_nonspecular, a.blocks, CLASS_DIELECTRIC, true, false);
        else WF_SCATTER_LAUNCH_SUB(k_scatter_nonspecular, a.blocks, CLASS_DIELECTRIC, false, false);
    }
    return SHM_OK;
}
