// k_vertex_env.hip — k_vertex (k_vertex_tri.hip, k_vertex_gen.hip) for scenes whose only image is an ImageInfinitelight: the class without textures, with the light's
// look-up and pdf compiled in (K_ENV_LIGHT, k_vertex.inl). Round 5: the coated object of the headline scene under a map ran the textured class at 2.1 Gray/s.
#define K_ENV_LIGHT true
#include "k_vertex.inl"

int wf_launch_vertex_tri_env(ShmScene* s, const ShadeArgs& a) {
    WF_VERTEX_LAUNCH_W3(true, false, true);
    return SHM_OK;
}
int wf_launch_vertex_gen_env(ShmScene* s, const ShadeArgs& a) {
    WF_VERTEX_LAUNCH(false, false, true);
    return SHM_OK;
}
