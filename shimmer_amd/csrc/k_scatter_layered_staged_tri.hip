// k_scatter_layered_staged_tri.hip — the LayeredBxDF queue's scattering half as dense per-wave stages (k_scatter_layered.inl). Scene class <TRI_ONLY, HAS_TEX> = <true,false>.
#include "shm/fp.h"
#define SHM_BASE_BXDF_CALL SHM_HD_NOINLINE  // the walks call the interface BxDFs instead of inlining them ~30 times
#include "k_scatter_layered.inl"

int wf_launch_scatter_layered_staged_tri(ShmScene* s, const ShadeArgs& a) {
    WF_SCATTER_LAYERED_LAUNCH(true, false);
    return SHM_OK;
}

// development build (-DLJ_CENSUS=1): what the four stages did, printed at scene destruction
void wf_layered_census() {
#ifdef LJ_CENSUS
    unsigned long long c[24];
    if (hipMemcpyFromSymbol(c, HIP_SYMBOL(g_lj_census), sizeof(c)) != hipSuccess || !c[0]) return;
    const char* name[4] = {"A set-up + light sample", "B walk, two steps", "C epilogue (pdf, RR, ray)", "N NEE (f, pdf)"};
    double ticks = 0.0;
    for (int k = 0; k < 4; ++k) ticks += (double)c[k * 3 + 2];
    for (int k = 0; k < 4; ++k)
        fprintf(stderr, "[layered census] %-26s passes %.3e at %.1f lanes, %.0f ticks per pass, %.1f %% of the stages' ticks\n", name[k], (double)c[k * 3], (double)c[k * 3 + 1] / (double)c[k * 3],
                (double)c[k * 3 + 2] / (double)c[k * 3], 100.0 * (double)c[k * 3 + 2] / ticks);
    fprintf(stderr, "[layered census] A: vertices %.3e -> walks %.3e, reflected at the top %.3e, NEE jobs %.3e | B: jobs in %.3e -> go on %.3e, left %.3e | N: queued %.3e\n", (double)c[1], (double)c[15],
            (double)c[16], (double)c[17], (double)c[4], (double)c[12], (double)c[13], (double)c[14]);
    fprintf(stderr, "[layered census] A rounds without a single NEE job %.3e, with at most 8: %.3e (of %.3e)\n", (double)c[18], (double)c[19], (double)c[0]);
#endif
}
