// k_scatter_layered_gen_env.hip — k_scatter_layered_gen.hip (the LayeredBxDF class in one pass per vertex) for scenes whose only image is an ImageInfinitelight (K_ENV_LIGHT).
#include "shm/fp.h"
#define SHM_BASE_BXDF_CALL SHM_HD_NOINLINE
#define K_ENV_LIGHT true
#include "k_scatter.inl"

int wf_launch_scatter_layered_gen_env(ShmScene* s, const ShadeArgs& a) {
    WF_SCATTER_LAUNCH(CLASS_LAYERED, false, false);
    return SHM_OK;
}
