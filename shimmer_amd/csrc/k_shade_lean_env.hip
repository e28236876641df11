// k_shade_lean_env.hip — the lean fused kernel (k_shade_lean.hip, k_shade_lean_gen.hip) for all-diffuse scenes whose only image is an ImageInfinitelight (round 5):
// k_shade.inl <HAS_LAYERED = false, TRI_ONLY, HAS_TEX = false, DIFFUSE_ONLY = true, EMIT_INLINE = false, SORT_CHUNK = false, ENV_LIGHT = true>. Until then an environment
// map put a scene into the textured class — ray differentials, auxiliary rays, the texture evaluators' registers — although nothing in it filters a texture: the object of
// the headline scene under a map ran at 2.9 Gray/s against 5.1 in its room. The light's look-up, sample and pdf are the shared functions of shm/path.h / shm/texture.h.
#include "k_shade.inl"

int wf_launch_shade_lean_env(ShmScene* s, const ShadeArgs& a) {
    constexpr int ctx_as_hit = 1;
#define CTX_AS_HIT_FLAG ((ctx_as_hit << 1) | ((ctx_as_hit && a.hit_kept) ? 4 : 0))
    WF_SHADE_LAUNCH((k_shade<false, true, false, true, false, false, true>));
#undef CTX_AS_HIT_FLAG
    WF_EMIT_JOBS_LAUNCH(ctx_as_hit);
    return SHM_OK;
}
// ... over the queue k_vertex diverted plain-diffuse hits to (a scene under a map that also holds other materials: k_shade_lean.hip)
int wf_launch_shade_lean_env_diverted(ShmScene* s, const ShadeArgs& a) {
    WF_SHADE_LAUNCH_DIVERTED((k_shade<false, true, false, true, false, false, true>));
    WF_EMIT_JOBS_LAUNCH(0);
    return SHM_OK;
}
