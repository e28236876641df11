// k_shade_lean.hip — fused shade kernels of the scene classes without LayeredBxDF and without textures.
#include "k_shade.inl"

int wf_launch_shade_lean(ShmScene* s, const ShadeArgs& a, bool tri_only, bool diffuse_only) {
    if (tri_only && diffuse_only) WF_SHADE_LAUNCH((k_shade<false, true, false, true>));
    else if (tri_only) WF_SHADE_LAUNCH((k_shade<false, true>));
    else WF_SHADE_LAUNCH((k_shade<false, false>));
    return SHM_OK;
}
