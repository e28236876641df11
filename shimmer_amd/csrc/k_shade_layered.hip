// k_shade_layered.hip — fused shade kernels of the scene classes with LayeredBxDF (CoatedDiffuse / CoatedConductor), no textures.
#include "k_shade.inl"

int wf_launch_shade_layered(ShmScene* s, const ShadeArgs& a, bool tri_only) {
    if (tri_only) WF_SHADE_LAUNCH((k_shade<true, true>));
    else WF_SHADE_LAUNCH((k_shade<true, false>));
    return SHM_OK;
}
