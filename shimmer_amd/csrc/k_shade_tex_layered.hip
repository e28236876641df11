// k_shade_tex_layered.hip — the general fused shade kernel: textures, image lights, force_diffuse and LayeredBxDF.
#include "k_shade.inl"

int wf_launch_shade_tex_layered(ShmScene* s, const ShadeArgs& a) {
    WF_SHADE_LAUNCH((k_shade<true, false, true>));
    return SHM_OK;
}
