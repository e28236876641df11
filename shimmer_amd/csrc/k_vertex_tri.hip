// k_vertex_tri.hip — k_vertex for scenes made of top-level triangles only, no textures.
#include "k_vertex.inl"

int wf_launch_vertex_tri(ShmScene* s, const ShadeArgs& a) {
    WF_VERTEX_LAUNCH(true, false);
    return SHM_OK;
}
