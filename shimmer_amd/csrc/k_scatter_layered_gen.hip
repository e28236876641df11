// k_scatter_layered_gen.hip — the scattering half of a vertex for the LayeredBxDF queue (CoatedDiffuse / CoatedConductor): the three
// random walks per vertex (f and pdf for NEE, sample_f) run here and nowhere else. Scene class <TRI_ONLY, HAS_TEX> = <false,false>.
#include "k_scatter.inl"

int wf_launch_scatter_layered_gen(ShmScene* s, const ShadeArgs& a) {
    if (layered_two_waves()) WF_SCATTER_LAUNCH(CLASS_LAYERED, false,false);
    else WF_SCATTER_LAUNCH_W1(CLASS_LAYERED, false,false);
    return SHM_OK;
}
