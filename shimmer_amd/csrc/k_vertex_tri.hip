// k_vertex_tri.hip — k_vertex for scenes made of top-level triangles only, no textures.
#include "k_vertex.inl"

int wf_launch_vertex_tri(ShmScene* s, const ShadeArgs& a) {
    // 159 VGPRs: three waves per SIMD hide the latency of the hit / vertex gathers (measured on C4: k_vertex 142 -> 110 ms per frame);
    // SHM_VERTEX_WAVES=2 selects the two-wave build for A/B runs. Chunks are material-sorted when the scene has more than one material.
    static int w2 = -1;
    if (w2 < 0) { const char* e = getenv("SHM_VERTEX_WAVES"); w2 = (e && atoi(e) == 2) ? 1 : 0; }
    if (w2) WF_VERTEX_LAUNCH(true, false, false);
    else if (wf_vertex_sort(s)) WF_VERTEX_LAUNCH_W3(true, false, true);
    else WF_VERTEX_LAUNCH_W3(true, false, false);
    return SHM_OK;
}
