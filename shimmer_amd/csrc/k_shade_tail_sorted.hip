// k_shade_tail_sorted.hip — the tail kernel (k_shade_tail.hip) with material-sorted chunks: k_shade.inl <HAS_LAYERED = false, TRI_ONLY = true, HAS_TEX = false,
// DIFFUSE_ONLY = false, EMIT_INLINE = true, SORT_CHUNK = true>. The late bounces of a deep render leave queues in image order with every material class mixed in a
// wave (C4, bounces 8-32: 11.5 of 64 lanes per instruction); the workgroup counting-sorts its 2048-entry chunk by the hit primitive's material first.
#ifndef K_SHADE_LEAN_WAVES
#define K_SHADE_LEAN_WAVES 3  // (168 VGPRs + 58 spilled, 68 B of scratch — and C4's shading 118 -> 107 ms against two waves at 202 without a spill: the kernel waits 58 % of its wave cycles; four waves: 138 spilled, 144 ms)
#endif
#include "k_shade.inl"

int wf_launch_shade_tail_sorted(ShmScene* s, const ShadeArgs& a) {
#define CTX_AS_HIT_FLAG 0
    WF_SHADE_LAUNCH((k_shade<false, true, false, false, true, true>));
#undef CTX_AS_HIT_FLAG
    return SHM_OK;
}
