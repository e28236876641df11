// k_vertex_tex.hip — k_vertex for scenes that bind image textures or an image infinite light (ray differentials, MIP filtering).
#include "k_vertex.inl"

int wf_launch_vertex_tex(ShmScene* s, const ShadeArgs& a) {
    WF_VERTEX_LAUNCH(false, true, true);
    return SHM_OK;
}
