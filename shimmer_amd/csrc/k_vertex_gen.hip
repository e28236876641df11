// k_vertex_gen.hip — k_vertex for scenes with quadrics / bilinear patches / instances, no textures.
#include "k_vertex.inl"

int wf_launch_vertex_gen(ShmScene* s, const ShadeArgs& a) {
    WF_VERTEX_LAUNCH(false, false, true);
    return SHM_OK;
}
