// k_vertex_tri.hip — k_vertex for scenes made of top-level triangles only, no textures.
#include "k_vertex.inl"

int wf_launch_vertex_tri(ShmScene* s, const ShadeArgs& a) {
    // 159 VGPRs: three waves per SIMD hide the latency of the hit / vertex gathers (measured on C4 against the two-wave build: k_vertex 142 -> 110 ms per frame).
    // Chunks are always material-sorted (a one-material scene pays the counting sort's pass over its chunk: the unsorted instantiations left the library in round 6).
    WF_VERTEX_LAUNCH_W3(true, false, true);
    return SHM_OK;
}
