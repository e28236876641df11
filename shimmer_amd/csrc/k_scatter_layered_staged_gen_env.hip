// k_scatter_layered_staged_gen_env.hip — k_scatter_layered_staged_gen.hip for scenes whose only image is an ImageInfinitelight (K_ENV_LIGHT; k_vertex_env.hip says why):
// the reference's showcase class — a coated object under an environment map.
#include "shm/fp.h"
#define SHM_BASE_BXDF_CALL SHM_HD_NOINLINE  // the walks call the interface BxDFs instead of inlining them ~30 times
#define K_ENV_LIGHT true
#include "k_scatter_layered.inl"

int wf_launch_scatter_layered_staged_gen_env(ShmScene* s, const ShadeArgs& a) {
    WF_SCATTER_LAYERED_LAUNCH(false, false);
    return SHM_OK;
}
