// k_shade_tail.hip — the fused vertex kernel for every material of a triangle scene without textures and without coated materials (k_shade.inl,
// <HAS_LAYERED = false, TRI_ONLY = true, HAS_TEX = false, DIFFUSE_ONLY = false>): the LATE bounces of a deep render in such a scene. There the queues
// hold a percent of the paths and the staged pipeline's four or five launches per bounce (k_vertex, the diverted fused kernel, one scatter kernel
// per class) each last as long as their slowest chunk; one launch does the same arithmetic per path (both pipelines are parity-green against the
// oracle, and a path's state arrays mean the same in both: the switch is made at a bounce boundary, render.hip).
#define K_SHADE_LEAN_WAVES 2  // (227 VGPRs without a spill; at three waves 115 would be spilled)
#include "k_shade.inl"

int wf_launch_shade_tail(ShmScene* s, const ShadeArgs& a) {
#define CTX_AS_HIT_FLAG 0  // (the tail kernel follows staged bounces: the vertex context stays the LightSampleContext)
    // material-sorted chunks when the scene holds more than one material (k_shade_tail_sorted.hip: its own translation unit — two instantiations of the kernel in one
    // unit stop the inliner, 121 spilled VGPRs in BOTH; SHM_TAIL_SORT=0: A/B)
    if (s->tail_sort && s->flat.materials.size() > 1) return wf_launch_shade_tail_sorted(s, a);
    WF_SHADE_LAUNCH((k_shade<false, true, false, false>));
#undef CTX_AS_HIT_FLAG
    return SHM_OK;
}
