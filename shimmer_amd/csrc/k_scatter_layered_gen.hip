// k_scatter_layered_gen.hip — the scattering half of a vertex for the LayeredBxDF queue (CoatedDiffuse / CoatedConductor): the three
// random walks per vertex (f and pdf for NEE, sample_f) run here and nowhere else. Scene class <TRI_ONLY, HAS_TEX> = <false,false>.
#include "shm/fp.h"
#define SHM_BASE_BXDF_CALL SHM_HD_NOINLINE  // the walks call the interface BxDFs instead of inlining them ~30 times
#include "k_scatter.inl"

int wf_launch_scatter_layered_gen(ShmScene* s, const ShadeArgs& a) {
    WF_SCATTER_LAUNCH(CLASS_LAYERED, false,false);
    return SHM_OK;
}
