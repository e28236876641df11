// k_shade_tex.hip — the general fused shade kernel without LayeredBxDF: textures, image lights, force_diffuse.
#include "k_shade.inl"

int wf_launch_shade_tex(ShmScene* s, const ShadeArgs& a) {
    WF_SHADE_LAUNCH((k_shade<false, false, true>));
    return SHM_OK;
}
