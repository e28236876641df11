// k_vertex_gen.hip — k_vertex for scenes with quadrics / bilinear patches / instances, no textures.
#include "k_vertex.inl"

int wf_launch_vertex_gen(ShmScene* s, const ShadeArgs& a) {
    if (wf_vertex_sort(s)) WF_VERTEX_LAUNCH(false, false, true);
    else WF_VERTEX_LAUNCH(false, false, false);
    return SHM_OK;
}
