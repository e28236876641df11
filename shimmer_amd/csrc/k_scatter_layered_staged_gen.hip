// k_scatter_layered_staged_gen.hip — the LayeredBxDF queue's scattering half as dense per-wave stages (k_scatter_layered.inl). Scene class <TRI_ONLY, HAS_TEX> = <false,false>.
#include "shm/fp.h"
#define SHM_BASE_BXDF_CALL SHM_HD_NOINLINE  // the walks call the interface BxDFs instead of inlining them ~30 times
#include "k_scatter_layered.inl"

int wf_launch_scatter_layered_staged_gen(ShmScene* s, const ShadeArgs& a) {
    WF_SCATTER_LAYERED_LAUNCH(false, false);
    return SHM_OK;
}
