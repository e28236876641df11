// host/host_mirror.cpp — C++ mirror of the reference's host-side scene-construction steps that sit
// directly either side of the hot path (the reference is compiled Rust; no Rust toolchain exists here,
// so the host side above the C ABI is C++; see INTEGRATION.md for the Rust shim a maintainer would add).
//
// Restates (paths relative to /root/reference/src):
//   aggregate.rs:207-467        BvhAggregate::new / build_recursive / flatten_bvh
//   bounding_box.rs:284-415     Bounds3::union, union_point, surface_area, max_dimension, Default
//   tile.rs:21-104              Tile::tile
//   camera.rs:507-523,594-642,893-963   CameraTransform::new, ProjectiveCameraBase::new, PerspectiveCamera::new
//   transform.rs:263-316        Transform::inverse / look_at / perspective / scale / translate
//   camera.rs:356-440           CameraBase::find_minimum_differentials (what Camera::approximate_dp_dxy falls back on)
//   image.rs:1333-1377          Image::write_pfm
//   shape/mesh.rs:179-358       TriQuadMesh::read_ply (the ply-rs 0.1.3 crate parses the container; restated from the PLY format)
//
// Third-party pieces the reference pulls in here, restated from their published behaviour (unpinned):
//   itertools 0.11 `partition` (aggregate.rs:362) and pdqselect 0.1.1 `select_by` (aggregate.rs:371,386);
//   both only decide the ORDER of primitives inside a partition, i.e. tie-breaking of BVH topology.
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/shimmer_hip.h"
#include "../shm/path.h"  // camera_generate_ray_differential for CameraBase::find_minimum_differentials (host use only)

extern "C" __attribute__((visibility("hidden"))) void shm_set_last_error(const char* msg);  // shimmer_hip.hip

namespace {

struct B3 {
    float mn[3], mx[3];
};
inline B3 b3_default() {  // bounding_box.rs:568-581: min = MAX, max = MIN (f32::MIN is the most negative float)
    B3 b;
    for (int i = 0; i < 3; ++i) { b.mn[i] = 3.40282346638528859812e+38f; b.mx[i] = -3.40282346638528859812e+38f; }
    return b;
}
inline float fmin_rs(float a, float b) { return (b != b) ? a : ((a != a) ? b : (a < b ? a : b)); }
inline float fmax_rs(float a, float b) { return (b != b) ? a : ((a != a) ? b : (a > b ? a : b)); }
inline B3 b3_union(const B3& a, const B3& b) {
    B3 r;
    for (int i = 0; i < 3; ++i) { r.mn[i] = fmin_rs(a.mn[i], b.mn[i]); r.mx[i] = fmax_rs(a.mx[i], b.mx[i]); }
    return r;
}
inline B3 b3_union_point(const B3& a, const float p[3]) {
    B3 r;
    for (int i = 0; i < 3; ++i) { r.mn[i] = fmin_rs(a.mn[i], p[i]); r.mx[i] = fmax_rs(a.mx[i], p[i]); }
    return r;
}
inline float b3_surface_area(const B3& b) {  // bounding_box.rs:394-397
    float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
    return 2.0f * (dx * dy + dx * dz + dy * dz);
}
inline int b3_max_dimension(const B3& b) {  // bounding_box.rs:404-415
    float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
    if (dx > dy && dx > dz) return 0;
    return (dy > dz) ? 1 : 2;
}

struct BvhPrimitive {  // aggregate.rs:484-493
    uint32_t primitive_index;
    B3 bounds;
    float centroid(int dim) const {  // 0.5 * min + (max * 0.5)
        return 0.5f * bounds.mn[dim] + bounds.mx[dim] * 0.5f;
    }
};

struct BuildNode {  // aggregate.rs:498-510
    B3 bounds;
    int left = -1, right = -1;
    uint8_t split_axis = 0;
    uint32_t first_prim_offset = 0;
    uint32_t n_primitives = 0;
};

// build_recursive is serial in the reference (aggregate.rs:389-394: "This can be done in parallel, but let's do it sequentially for now"). Here the SAME recursion runs
// over disjoint sub-ranges on host threads: a builder that is given `tasks` stops at ranges of `grain` primitives or fewer and records them (a placeholder node each);
// every recorded range is then built by the same code into its own node vector, and shm_bvh_build splices the vectors in at the placeholders. A range's primitives end
// up in ordered[begin, end) whoever builds it (the leaves of a pre-order walk cover the primitives in order: the reference's running offset IS `begin`), every node runs
// the same partition over the same elements in the same order, and the pre-order numbering is restored by the splice: nodes and primitive order are bit-identical to the
// serial build's (tests/test_host_mirror.py builds both ways).
struct BuildTask {
    size_t begin, end;
    int top_index;  // the placeholder in the top builder's nodes
    std::vector<BuildNode> nodes;
};
constexpr uint32_t TASK_PLACEHOLDER = 0xffffffffu;  // BuildNode::n_primitives of a placeholder (first_prim_offset: the task)

struct Builder {
    BvhPrimitive* prims;
    std::vector<BuildNode> build_nodes;
    uint32_t* ordered;
    int split_method;
    std::vector<BuildTask>* tasks = nullptr;
    size_t grain = 0;

    // itertools::partition: elements for which pred is true are moved to the front (unstable, two-ended).
    template <typename Pred>
    static size_t partition(BvhPrimitive* a, size_t n, Pred pred) {
        size_t split_index = 0;
        size_t front = 0, back = n;
        while (front < back) {
            if (!pred(a[front])) {
                bool swapped = false;
                while (back > front + 1) {
                    --back;
                    if (pred(a[back])) {
                        std::swap(a[front], a[back]);
                        swapped = true;
                        break;
                    }
                }
                if (!swapped) break;
            }
            ++split_index;
            ++front;
        }
        return split_index;
    }

    // the two folds of aggregate.rs:309-313 / 340-346 over one range: the union of the primitives' bounds and of their centroids. Above the sub-range tasks (`fold_threads`
    // > 1, a large range) the range is folded in consecutive pieces on host threads and the pieces are united in order: fmin_rs / fmax_rs keep the LATER operand of two
    // equal ones, in a piece and between pieces alike, so the result — the sign of a zero included — is the serial fold's
    unsigned fold_threads = 1;
    void range_bounds(size_t begin, size_t end, B3& bounds, B3& cb) const {
        auto fold = [this](size_t b0, size_t e0, B3& bo, B3& co) {
            bo = b3_default();
            co = b3_default();
            for (size_t i = b0; i < e0; ++i) {
                bo = b3_union(bo, prims[i].bounds);
                float c[3] = {prims[i].centroid(0), prims[i].centroid(1), prims[i].centroid(2)};
                co = b3_union_point(co, c);
            }
        };
        const size_t n = end - begin;
        if (fold_threads <= 1 || n < ((size_t)1 << 17)) { fold(begin, end, bounds, cb); return; }
        const size_t pieces = std::min<size_t>(fold_threads, n >> 15);
        std::vector<B3> pb(pieces), pc(pieces);
        std::vector<std::thread> pool;
        auto piece = [&](size_t k) { fold(begin + n * k / pieces, begin + n * (k + 1) / pieces, pb[k], pc[k]); };
        size_t started = 1;
        try {
            for (; started < pieces; ++started) pool.emplace_back(piece, started);
        } catch (...) {}
        piece(0);
        for (std::thread& t : pool) t.join();
        for (size_t k = started; k < pieces; ++k) piece(k);  // (pieces no thread could be had for)
        bounds = b3_default();
        cb = b3_default();
        for (size_t k = 0; k < pieces; ++k) { bounds = b3_union(bounds, pb[k]); cb = b3_union(cb, pc[k]); }
    }

    int make_leaf(int node_idx, size_t begin, size_t end, const B3& bounds) {  // aggregate.rs:326-337, 348-358
        BuildNode& node = build_nodes[node_idx];
        const uint32_t first = (uint32_t)begin;  // (aggregate.rs:327-331's ordered_prims_offset: every primitive before `begin` is in an earlier leaf)
        for (size_t i = begin; i < end; ++i) ordered[i] = prims[i].primitive_index;
        node.bounds = bounds;
        node.first_prim_offset = first;
        node.n_primitives = (uint32_t)(end - begin);
        return node_idx;
    }

    // aggregate.rs:304-419, recursion made explicit where it is a tail (right child) to bound host stack depth.
    int build(size_t begin, size_t end) {
        int node_idx = (int)build_nodes.size();
        build_nodes.emplace_back();
        if (tasks && end - begin <= grain) {  // a sub-range for another thread: the same call on a builder of its own
            build_nodes[node_idx].n_primitives = TASK_PLACEHOLDER;
            build_nodes[node_idx].first_prim_offset = (uint32_t)tasks->size();
            tasks->push_back(BuildTask{begin, end, node_idx, {}});
            return node_idx;
        }
        B3 bounds, cb;
        range_bounds(begin, end, bounds, cb);
        if (b3_surface_area(bounds) == 0.0f || end - begin == 1) return make_leaf(node_idx, begin, end, bounds);
        int dim = b3_max_dimension(cb);
        if (cb.mx[dim] == cb.mn[dim]) return make_leaf(node_idx, begin, end, bounds);
        size_t n = end - begin;
        size_t split_index;
        auto median_split = [&]() {
            size_t mid = n / 2;
            std::nth_element(prims + begin, prims + begin + mid, prims + end,
                             [dim](const BvhPrimitive& a, const BvhPrimitive& b) { return a.centroid(dim) < b.centroid(dim); });
            return mid;
        };
        if (split_method == 0) {
            float pmid = (cb.mn[dim] + cb.mx[dim]) / 2.0f;
            split_index = partition(prims + begin, n, [dim, pmid](const BvhPrimitive& p) { return p.centroid(dim) < pmid; });
            if (split_index == 0 || split_index == n) split_index = median_split();
        } else {
            split_index = median_split();
        }
        int left = build(begin, begin + split_index);
        int right = build(begin + split_index, end);
        BuildNode& node = build_nodes[node_idx];
        node.bounds = b3_union(build_nodes[left].bounds, build_nodes[right].bounds);  // init_interior, aggregate.rs:540-557 (above placeholders: made again once they are built)
        node.left = left;
        node.right = right;
        node.split_axis = (uint8_t)dim;
        return node_idx;
    }
};

thread_local std::string g_host_err;

// ---- 4x4 helpers (row-major), host precision is free: the results are INPUTS to oracle and GPU alike ----
struct M4 {
    double m[4][4];
};
M4 m4_identity() {
    M4 r;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = (i == j) ? 1.0 : 0.0;
    return r;
}
M4 m4_mul(const M4& a, const M4& b) {
    M4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += a.m[i][k] * b.m[k][j];
            r.m[i][j] = s;
        }
    return r;
}
bool m4_inverse(const M4& a, M4& out) {  // Gauss-Jordan with partial pivoting
    double aug[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { aug[i][j] = a.m[i][j]; aug[i][j + 4] = (i == j) ? 1.0 : 0.0; }
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        for (int r = c + 1; r < 4; ++r) if (std::fabs(aug[r][c]) > std::fabs(aug[piv][c])) piv = r;
        if (aug[piv][c] == 0.0) return false;
        if (piv != c) for (int j = 0; j < 8; ++j) std::swap(aug[c][j], aug[piv][j]);
        double inv = 1.0 / aug[c][c];
        for (int j = 0; j < 8; ++j) aug[c][j] *= inv;
        for (int r = 0; r < 4; ++r) {
            if (r == c) continue;
            double f = aug[r][c];
            if (f != 0.0) for (int j = 0; j < 8; ++j) aug[r][j] -= f * aug[c][j];
        }
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) out.m[i][j] = aug[i][j + 4];
    return true;
}
M4 m4_scale(double x, double y, double z) { M4 r = m4_identity(); r.m[0][0] = x; r.m[1][1] = y; r.m[2][2] = z; return r; }
M4 m4_translate(double x, double y, double z) { M4 r = m4_identity(); r.m[0][3] = x; r.m[1][3] = y; r.m[2][3] = z; return r; }
void m4_to_f32(const M4& a, float* out) { for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) out[i * 4 + j] = (float)a.m[i][j]; }
void m4_point(const M4& a, const double p[3], double out[3]) {
    double r[4];
    for (int i = 0; i < 4; ++i) r[i] = a.m[i][0] * p[0] + a.m[i][1] * p[1] + a.m[i][2] * p[2] + a.m[i][3];
    for (int i = 0; i < 3; ++i) out[i] = (r[3] == 1.0) ? r[i] : r[i] / r[3];
}

}  // namespace

extern "C" {

// Test entry: the Bounds3f operations BvhAggregate::build_recursive is made of (bounding_box.rs: union :432-446, union_point :417-430,
// surface_area :394-397, volume :399-402, max_dimension :404-415), exactly the helpers the builder above calls, so that the reference's own
// vectors for them (bounding_box.rs:699-733, 950-995) can be replayed. out[16]: union(a, b) min, max | union_point(a, p) min, max |
// surface_area(a), volume(a), max_dimension(a), 0.
int shm_bounds3_probe(const float a[6], const float b[6], const float p[3], float out[16]) {
    if (!a || !b || !p || !out) return SHM_ERR_INVALID_ARGUMENT;
    B3 ba, bb;
    for (int k = 0; k < 3; ++k) { ba.mn[k] = a[k]; ba.mx[k] = a[3 + k]; bb.mn[k] = b[k]; bb.mx[k] = b[3 + k]; }
    const B3 u = b3_union(ba, bb), up = b3_union_point(ba, p);
    for (int k = 0; k < 3; ++k) { out[k] = u.mn[k]; out[3 + k] = u.mx[k]; out[6 + k] = up.mn[k]; out[9 + k] = up.mx[k]; }
    out[12] = b3_surface_area(ba);
    out[13] = (ba.mx[0] - ba.mn[0]) * (ba.mx[1] - ba.mn[1]) * (ba.mx[2] - ba.mn[2]);
    out[14] = (float)b3_max_dimension(ba);
    out[15] = 0.0f;
    return SHM_OK;
}

int shm_bvh_build(const float* prim_bounds, uint32_t n, int split_method, ShmBvhNode* nodes_out,
                  uint32_t* n_nodes_out, uint32_t* prim_order_out) {
    if (!prim_bounds || n == 0 || !nodes_out || !n_nodes_out || !prim_order_out || split_method < 0 || split_method > 1)
        return SHM_ERR_INVALID_ARGUMENT;
    std::vector<BvhPrimitive> prims(n);
    for (uint32_t i = 0; i < n; ++i) {
        prims[i].primitive_index = i;
        for (int k = 0; k < 3; ++k) {
            prims[i].bounds.mn[k] = prim_bounds[6ull * i + k];
            prims[i].bounds.mx[k] = prim_bounds[6ull * i + 3 + k];
            if (prim_bounds[6ull * i + k] != prim_bounds[6ull * i + k]) return SHM_ERR_INVALID_ARGUMENT;  // "Unexpected NaN"
        }
    }
    // host threads for the sub-ranges (SHM_BVH_THREADS; 1 = the serial build): the ranges of n / 64 primitives or fewer below the first levels
    unsigned n_threads = std::min(std::max(std::thread::hardware_concurrency(), 1u), 64u);
    if (const char* e = getenv("SHM_BVH_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 1024) n_threads = (unsigned)v; }
    if (n < 65536u) n_threads = 1;
    std::vector<BuildTask> tasks;
    Builder b;
    b.prims = prims.data();
    b.build_nodes.reserve(n_threads > 1 ? 1024 : 2ull * n);
    b.ordered = prim_order_out;
    b.split_method = split_method;
    if (n_threads > 1) { b.tasks = &tasks; b.grain = n / 64u; b.fold_threads = std::min(n_threads, 16u); }
    b.build(0, n);
    if (!tasks.empty()) {
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            for (size_t t = next.fetch_add(1); t < tasks.size(); t = next.fetch_add(1)) {
                Builder sub;
                sub.prims = prims.data();
                sub.ordered = prim_order_out;
                sub.split_method = split_method;
                sub.build_nodes.reserve(2 * (tasks[t].end - tasks[t].begin));
                sub.build(tasks[t].begin, tasks[t].end);
                tasks[t].nodes.swap(sub.build_nodes);
            }
        };
        std::vector<std::thread> pool;
        try {
            for (unsigned t = 1; t < std::min<size_t>(n_threads, tasks.size()); ++t) pool.emplace_back(worker);
        } catch (...) {}  // (no more threads: this one builds what is left)
        worker();
        for (std::thread& t : pool) t.join();
        // the interior nodes above the placeholders: bounds from their children, innermost first (pre-order: children have the larger indices)
        auto bounds_of = [&](int i) -> const B3& { const BuildNode& c = b.build_nodes[i]; return c.n_primitives == TASK_PLACEHOLDER ? tasks[c.first_prim_offset].nodes[0].bounds : c.bounds; };
        for (size_t i = b.build_nodes.size(); i-- > 0;) {
            BuildNode& node = b.build_nodes[i];
            if (node.n_primitives == 0) node.bounds = b3_union(bounds_of(node.left), bounds_of(node.right));
        }
    }
    // flatten_bvh, aggregate.rs:425-467: DFS, first child right after the parent. Our build order IS DFS
    // pre-order (node created before its left subtree, right subtree after it), so indices carry over — with every placeholder replaced by its sub-range's nodes.
    std::vector<uint32_t> final_index(b.build_nodes.size());
    uint64_t total64 = 0;
    for (size_t i = 0; i < b.build_nodes.size(); ++i) {
        final_index[i] = (uint32_t)total64;
        total64 += b.build_nodes[i].n_primitives == TASK_PLACEHOLDER ? tasks[b.build_nodes[i].first_prim_offset].nodes.size() : 1u;
    }
    if (total64 > 0xffffffffull) return SHM_ERR_UNSUPPORTED;
    const uint32_t total = (uint32_t)total64;
    std::atomic<int> rc_emit{SHM_OK};
    std::atomic<uint64_t> n_in_leaves{0};
    auto emit = [&](const BuildNode& bn, uint32_t base, ShmBvhNode& ln, uint64_t& leaf_prims) {
        memset(&ln, 0, sizeof(ln));
        for (int k = 0; k < 3; ++k) { ln.bmin[k] = bn.bounds.mn[k]; ln.bmax[k] = bn.bounds.mx[k]; }
        if (bn.n_primitives > 0) {
            if (bn.n_primitives >= 65536) rc_emit = SHM_ERR_UNSUPPORTED;  // aggregate.rs:441 debug_assert
            ln.offset = bn.first_prim_offset;
            ln.n_prims = (uint16_t)bn.n_primitives;
            ln.axis = 0;
            leaf_prims += bn.n_primitives;
        } else {
            ln.offset = base;  // second_child_offset (the caller's numbering)
            ln.n_prims = 0;
            ln.axis = bn.split_axis;
        }
    };
    {
        uint64_t leaf_prims = 0;
        for (size_t i = 0; i < b.build_nodes.size(); ++i) {
            const BuildNode& bn = b.build_nodes[i];
            if (bn.n_primitives == TASK_PLACEHOLDER) continue;
            emit(bn, bn.n_primitives == 0 ? final_index[bn.right] : 0u, nodes_out[final_index[i]], leaf_prims);
        }
        n_in_leaves += leaf_prims;
    }
    if (!tasks.empty()) {
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            uint64_t leaf_prims = 0;
            for (size_t t = next.fetch_add(1); t < tasks.size(); t = next.fetch_add(1)) {
                const uint32_t base = final_index[tasks[t].top_index];
                const std::vector<BuildNode>& nodes = tasks[t].nodes;
                for (size_t i = 0; i < nodes.size(); ++i) emit(nodes[i], nodes[i].n_primitives == 0 ? base + (uint32_t)nodes[i].right : 0u, nodes_out[base + i], leaf_prims);
            }
            n_in_leaves += leaf_prims;
        };
        std::vector<std::thread> pool;
        try {
            for (unsigned t = 1; t < std::min<size_t>(n_threads, tasks.size()); ++t) pool.emplace_back(worker);
        } catch (...) {}
        worker();
        for (std::thread& t : pool) t.join();
    }
    if (rc_emit != SHM_OK) return rc_emit;
    if (n_in_leaves != n) return SHM_ERR_INTERNAL;
    *n_nodes_out = total;
    return SHM_OK;
}

int shm_tile_bounds(const int32_t pb[4], int32_t tile_w, int32_t tile_h, ShmTile* tiles_out, uint32_t* n_out) {
    if (!pb || !tiles_out || !n_out || tile_w <= 0 || tile_h <= 0) return SHM_ERR_INVALID_ARGUMENT;
    // tile.rs:21-104
    int32_t image_width = pb[2] - pb[0];
    int32_t image_height = pb[3] - pb[1];
    int32_t nh = image_width / tile_w, rh = image_width % tile_w;
    int32_t nv = image_height / tile_h, rv = image_height % tile_h;
    uint32_t k = 0;
    for (int32_t ty = 0; ty < nv; ++ty) {
        for (int32_t tx = 0; tx < nh; ++tx) {
            int32_t sx = pb[0] + tx * tile_w, sy = pb[1] + ty * tile_h;
            tiles_out[k++] = ShmTile{sx, sy, sx + tile_w, sy + tile_h};
        }
        if (rh > 0) {
            int32_t sx = pb[0] + nh * tile_w, sy = pb[1] + ty * tile_h;
            tiles_out[k++] = ShmTile{sx, sy, sx + rh, sy + tile_h};
        }
    }
    if (rv > 0) {
        for (int32_t tx = 0; tx < nh; ++tx) {
            int32_t sx = pb[0] + tx * tile_w, sy = pb[1] + nv * tile_h;
            tiles_out[k++] = ShmTile{sx, sy, sx + tile_w, sy + rv};
        }
    }
    if (rh > 0 && rv > 0) {
        int32_t sx = pb[0] + nh * tile_w, sy = pb[1] + nv * tile_h;
        tiles_out[k++] = ShmTile{sx, sy, sx + rh, sy + rv};
    }
    *n_out = k;
    return SHM_OK;
}

// ProjectiveCameraBase::new for either projection (camera.rs:594-642); fov_deg < 0 selects Transform::orthographic(0, 1)
static int projective_camera(const float world_from_camera[16], float fov_deg, const int32_t full_resolution[2],
                             float lens_radius, float focal_distance, ShmCamera* out, float render_from_world_out[16],
                             uint32_t render_space = SHM_RENDER_SPACE_CAMERA_WORLD, float frame_aspect_ratio = 0.0f, const float* screen_window = nullptr) {
    if (!world_from_camera || !full_resolution || !out || full_resolution[0] <= 0 || full_resolution[1] <= 0 || render_space > SHM_RENDER_SPACE_WORLD)
        return SHM_ERR_INVALID_ARGUMENT;
    M4 wfc;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) wfc.m[i][j] = world_from_camera[i * 4 + j];
    // CameraTransform::new (camera.rs:507-523): CameraWorld (main.rs:66 default) keeps world axes and puts the camera at the origin, Camera
    // renders in camera space, World in world space
    double origin[3] = {0, 0, 0}, p_camera[3];
    m4_point(wfc, origin, p_camera);
    M4 world_from_render = render_space == SHM_RENDER_SPACE_CAMERA ? wfc
                           : (render_space == SHM_RENDER_SPACE_WORLD ? m4_identity() : m4_translate(p_camera[0], p_camera[1], p_camera[2]));
    M4 render_from_world;
    if (!m4_inverse(world_from_render, render_from_world)) return SHM_ERR_INVALID_ARGUMENT;
    M4 render_from_camera = m4_mul(render_from_world, wfc);
    // Transform::perspective(fov, 1e-2, 1000) (transform.rs:305-316)
    double n = 1e-2, f = 1000.0;
    M4 persp = m4_identity();
    persp.m[2][2] = f / (f - n);
    persp.m[2][3] = -f * n / (f - n);
    persp.m[3][2] = 1.0;
    persp.m[3][3] = 0.0;
    double inv_tan = 1.0 / std::tan((3.14159265358979323846 / 180.0) * fov_deg / 2.0);
    M4 screen_from_camera = m4_mul(m4_scale(inv_tan, inv_tan, 1.0), persp);
    const bool ortho = fov_deg < 0.0f;
    if (ortho) screen_from_camera = m4_identity();  // Transform::orthographic(0, 1) = scale(1, 1, 1 / (1 - 0)) * translate(0, 0, -0) (transform.rs:293-303)
    // screen window from the aspect ratio, "frameaspectratio" or "screenwindow" (camera.rs:848-877: the file gives x0 x1 y0 y1)
    double frame = frame_aspect_ratio > 0.0f ? (double)frame_aspect_ratio : (double)full_resolution[0] / (double)full_resolution[1];
    double sw[4];  // min.x, min.y, max.x, max.y
    if (frame > 1.0) { sw[0] = -frame; sw[1] = -1.0; sw[2] = frame; sw[3] = 1.0; }
    else { sw[0] = -1.0; sw[1] = -1.0 / frame; sw[2] = 1.0; sw[3] = 1.0 / frame; }
    if (screen_window) { sw[0] = screen_window[0]; sw[1] = screen_window[2]; sw[2] = screen_window[1]; sw[3] = screen_window[3]; }
    if (!(sw[2] > sw[0]) || !(sw[3] > sw[1])) return SHM_ERR_INVALID_ARGUMENT;
    // ProjectiveCameraBase::new (camera.rs:612-634)
    M4 ndc_from_screen = m4_mul(m4_scale(1.0 / (sw[2] - sw[0]), 1.0 / (sw[3] - sw[1]), 1.0), m4_translate(-sw[0], -sw[3], 0.0));
    M4 raster_from_ndc = m4_scale((double)full_resolution[0], -(double)full_resolution[1], 1.0);
    M4 raster_from_screen = m4_mul(raster_from_ndc, ndc_from_screen);
    M4 screen_from_raster, camera_from_screen;
    if (!m4_inverse(raster_from_screen, screen_from_raster) || !m4_inverse(screen_from_camera, camera_from_screen))
        return SHM_ERR_INVALID_ARGUMENT;
    M4 camera_from_raster = m4_mul(camera_from_screen, screen_from_raster);
    memset(out, 0, sizeof(*out));
    m4_to_f32(camera_from_raster, out->camera_from_raster);
    m4_to_f32(render_from_camera, out->render_from_camera);
    // dx_camera / dy_camera (camera.rs:907-910)
    double px[3] = {1, 0, 0}, py[3] = {0, 1, 0}, p0[3] = {0, 0, 0}, a[3], b[3], c[3];
    m4_point(camera_from_raster, px, a);
    m4_point(camera_from_raster, py, b);
    m4_point(camera_from_raster, p0, c);
    for (int i = 0; i < 3; ++i) { out->dx_camera[i] = (float)(a[i] - c[i]); out->dy_camera[i] = (float)(b[i] - c[i]); }
    if (ortho)  // camera.rs:727-728: camera_from_raster.apply(Vector3f::X / ::Y), a vector transform
        for (int i = 0; i < 3; ++i) { out->dx_camera[i] = (float)camera_from_raster.m[i][0]; out->dy_camera[i] = (float)camera_from_raster.m[i][1]; }
    out->lens_radius = lens_radius;
    out->focal_distance = focal_distance;
    out->shutter_open = 0.0f;
    out->shutter_close = 1.0f;
    out->kind = ortho ? SHM_CAMERA_ORTHOGRAPHIC : SHM_CAMERA_PERSPECTIVE;
    M4 camera_from_render;
    if (!m4_inverse(render_from_camera, camera_from_render)) return SHM_ERR_INVALID_ARGUMENT;
    m4_to_f32(camera_from_render, out->camera_from_render);
    if (ortho) {  // camera.rs:733-736
        for (int i = 0; i < 3; ++i) {
            out->min_dir_differential_x[i] = out->min_dir_differential_y[i] = 0.0f;
            out->min_pos_differential_x[i] = out->dx_camera[i];
            out->min_pos_differential_y[i] = out->dy_camera[i];
        }
    } else {
        // CameraBase::find_minimum_differentials, camera.rs:356-440: 512 samples along the film diagonal, p_lens = (0.5, 0.5)
        using namespace shm;
        const Float inf = infinity();
        V3 min_pos_x = v3s(inf), min_pos_y = v3s(inf), min_dir_x = v3s(inf), min_dir_y = v3s(inf);
        const int n = 512;
        for (int i = 0; i < n; ++i) {
            V2 p_film = v2((Float)i / (Float)(n - 1) * (Float)full_resolution[0], (Float)i / (Float)(n - 1) * (Float)full_resolution[1]);
            AuxRays aux = aux_none();
            Ray ray = camera_generate_ray_differential(*out, p_film, v2(0.5f, 0.5f), &aux);
            V3 dox = xf_vector(out->camera_from_render, aux.rx_o - ray.o);
            if (length(dox) < length(min_pos_x)) min_pos_x = dox;
            V3 doy = xf_vector(out->camera_from_render, aux.ry_o - ray.o);
            if (length(doy) < length(min_pos_y)) min_pos_y = doy;
            ray.d = normalize(ray.d);
            aux.rx_d = normalize(aux.rx_d);
            aux.ry_d = normalize(aux.ry_d);
            Frame f = frame_from_z(ray.d);
            V3 df = f.to_local(ray.d);
            V3 dxf = normalize(f.to_local(aux.rx_d));
            V3 dyf = normalize(f.to_local(aux.ry_d));
            if (length(dxf - df) < length(min_dir_x)) min_dir_x = dxf - df;
            if (length(dyf - df) < length(min_dir_y)) min_dir_y = dyf - df;
        }
        const V3* src[4] = {&min_pos_x, &min_pos_y, &min_dir_x, &min_dir_y};
        float* dst[4] = {out->min_pos_differential_x, out->min_pos_differential_y, out->min_dir_differential_x, out->min_dir_differential_y};
        for (int k = 0; k < 4; ++k) { dst[k][0] = src[k]->x; dst[k][1] = src[k]->y; dst[k][2] = src[k]->z; }
    }
    if (render_from_world_out) m4_to_f32(render_from_world, render_from_world_out);
    return SHM_OK;
}

int shm_camera_perspective(const float world_from_camera[16], float fov_deg, const int32_t full_resolution[2],
                           float lens_radius, float focal_distance, ShmCamera* out, float render_from_world_out[16]) {
    if (!(fov_deg > 0.0f)) return SHM_ERR_INVALID_ARGUMENT;
    return projective_camera(world_from_camera, fov_deg, full_resolution, lens_radius, focal_distance, out, render_from_world_out);
}
int shm_camera_orthographic(const float world_from_camera[16], const int32_t full_resolution[2], float lens_radius,
                            float focal_distance, ShmCamera* out, float render_from_world_out[16]) {
    return projective_camera(world_from_camera, -1.0f, full_resolution, lens_radius, focal_distance, out, render_from_world_out);
}

int shm_camera_create(const ShmCameraParams* p, ShmCamera* out, float render_from_world_out[16]) {
    if (!p || (p->kind != SHM_CAMERA_PERSPECTIVE && p->kind != SHM_CAMERA_ORTHOGRAPHIC)) return SHM_ERR_INVALID_ARGUMENT;
    if (p->kind == SHM_CAMERA_PERSPECTIVE && !(p->fov_deg > 0.0f)) return SHM_ERR_INVALID_ARGUMENT;
    return projective_camera(p->world_from_camera, p->kind == SHM_CAMERA_PERSPECTIVE ? p->fov_deg : -1.0f, p->full_resolution, p->lens_radius, p->focal_distance, out,
                             render_from_world_out, p->render_space, p->frame_aspect_ratio, p->has_screen_window ? p->screen_window : nullptr);
}

int shm_film_get_image(const ShmFilmPixel* film, uint64_t n_pixels, const float m[9], int write_fp16, float* rgb_out) {
    if (!film || !m || !rgb_out) return SHM_ERR_INVALID_ARGUMENT;
    const float max_f16 = 65504.0f;
    for (uint64_t i = 0; i < n_pixels; ++i) {
        // get_pixel_rgb, film.rs:720-738
        float rgb[3] = {(float)film[i].rgb_sum[0], (float)film[i].rgb_sum[1], (float)film[i].rgb_sum[2]};
        const double weight_sum = film[i].weight_sum;
        if (weight_sum != 0.0) {
            const float w = (float)weight_sum;
            rgb[0] /= w; rgb[1] /= w; rgb[2] /= w;
        }
        for (int c = 0; c < 3; ++c) rgb[c] += 0.0f;  // splat_scale * rgb_splat / filter_integral: nothing splats on this path
        float out[3];
        for (int r = 0; r < 3; ++r) {  // mul_mat_vec, square_matrix.rs:474-491: out[i] starts at 0 and accumulates in j order
            float acc = 0.0f;
            for (int c = 0; c < 3; ++c) acc += m[3 * r + c] * rgb[c];
            out[r] = acc;
        }
        if (write_fp16) {  // film.rs:676-690, as written there
            float mx = -INFINITY;
            for (int c = 0; c < 3; ++c) mx = (out[c] > mx || mx != mx) ? out[c] : mx;
            if (mx > max_f16) {
                if (out[0] > max_f16) out[0] = max_f16;
                if (out[1] > max_f16) out[0] = max_f16;  // (sic) the reference assigns r here
                if (out[2] > max_f16) out[2] = max_f16;
            }
        }
        rgb_out[3 * i] = out[0]; rgb_out[3 * i + 1] = out[1]; rgb_out[3 * i + 2] = out[2];
    }
    return SHM_OK;
}

int shm_write_pfm(const char* path, const float* rgb, int32_t width, int32_t height) {
    if (!path || !rgb || width <= 0 || height <= 0) return SHM_ERR_INVALID_ARGUMENT;
    FILE* fp = fopen(path, "wb");
    if (!fp) return SHM_ERR_INVALID_ARGUMENT;
    // image.rs:1333-1377: "PF\n<w> <h>\n-1.0\n" then rows bottom-to-top, little-endian f32 RGB
    fprintf(fp, "PF\n%d %d\n-1.0\n", width, height);
    for (int y = height - 1; y >= 0; --y) {
        if (fwrite(rgb + 3ull * (size_t)y * (size_t)width, sizeof(float), 3ull * (size_t)width, fp) != 3ull * (size_t)width) {
            fclose(fp);
            return SHM_ERR_INTERNAL;
        }
    }
    fclose(fp);
    return SHM_OK;
}


// ---- TriQuadMesh::read_ply, shape/mesh.rs:194-287 ---------------------------------------------------------------
namespace {
struct PlyProp { std::string name; int type; int count_type; bool is_list; };  // type: index into kPlyTypes
struct PlyElem { std::string name; size_t count; std::vector<PlyProp> props; };
const char* const kPlyTypes[][2] = {{"char", "int8"}, {"uchar", "uint8"}, {"short", "int16"}, {"ushort", "uint16"},
                                    {"int", "int32"}, {"uint", "uint32"}, {"float", "float32"}, {"double", "float64"}};
const int kPlySize[] = {1, 1, 2, 2, 4, 4, 4, 8};
int ply_type(const std::string& t) {
    for (int i = 0; i < 8; ++i) if (t == kPlyTypes[i][0] || t == kPlyTypes[i][1]) return i;
    return -1;
}
struct PlyReader {
    FILE* fp;
    int format;  // 0 ascii, 1 binary little endian, 2 binary big endian
    bool fail = false;
    double read(int type) {  // one scalar of the given PLY type, as a double
        if (format == 0) {
            char tok[64];
            if (fscanf(fp, "%63s", tok) != 1) { fail = true; return 0.0; }
            return atof(tok);
        }
        unsigned char b[8];
        const int n = kPlySize[type];
        if (fread(b, 1, (size_t)n, fp) != (size_t)n) { fail = true; return 0.0; }
        if (format == 2) std::reverse(b, b + n);
        switch (type) {
            case 0: { int8_t v; memcpy(&v, b, 1); return v; }
            case 1: { uint8_t v; memcpy(&v, b, 1); return v; }
            case 2: { int16_t v; memcpy(&v, b, 2); return v; }
            case 3: { uint16_t v; memcpy(&v, b, 2); return v; }
            case 4: { int32_t v; memcpy(&v, b, 4); return v; }
            case 5: { uint32_t v; memcpy(&v, b, 4); return v; }
            case 6: { float v; memcpy(&v, b, 4); return v; }
            default: { double v; memcpy(&v, b, 8); return v; }
        }
    }
};
int ply_error(FILE* fp, const std::string& msg) {
    if (fp) fclose(fp);
    shm_set_last_error(msg.c_str());
    return SHM_ERR_INVALID_ARGUMENT;
}
}  // namespace

static int ply_read_impl(FILE* fp, ShmPlyMesh* out, bool& fp_closed);

int shm_ply_read(const char* filename, ShmPlyMesh* out) {
    if (!filename || !out) return SHM_ERR_INVALID_ARGUMENT;
    memset(out, 0, sizeof(*out));
    FILE* fp = fopen(filename, "rb");
    if (!fp) return ply_error(nullptr, "Unable to read PLY file");  // mesh.rs:200
    // the element counts come from an untrusted header: nothing may unwind across the ABI
    bool fp_closed = false;
    try {
        return ply_read_impl(fp, out, fp_closed);
    } catch (const std::bad_alloc&) {
        if (!fp_closed) fclose(fp);
        shm_ply_free(out);
        shm_set_last_error("PLY: out of memory");
        return SHM_ERR_OUT_OF_MEMORY;
    } catch (const std::exception& e) {
        if (!fp_closed) fclose(fp);
        shm_ply_free(out);
        shm_set_last_error((std::string("PLY: ") + e.what()).c_str());
        return SHM_ERR_INVALID_ARGUMENT;
    }
}

// ply_error closes the file on every error return; fp_closed tells the wrapper whether an exception found it open
static int ply_read_impl(FILE* fp, ShmPlyMesh* out, bool& fp_closed) {
    long file_size = 0;
    if (fseek(fp, 0, SEEK_END) == 0) { file_size = ftell(fp); fseek(fp, 0, SEEK_SET); }
    // header
    char line[1024];
    if (!fgets(line, sizeof line, fp) || strncmp(line, "ply", 3) != 0) return fp_closed = true, ply_error(fp, "not a PLY file");
    int format = -1;
    std::vector<PlyElem> elems;
    bool ended = false;
    while (fgets(line, sizeof line, fp)) {
        char a[256] = "", b[256] = "", c[256] = "", d[256] = "", e[256] = "";
        int n = sscanf(line, "%255s %255s %255s %255s %255s", a, b, c, d, e);
        if (n <= 0) continue;
        std::string key = a;
        if (key == "end_header") { ended = true; break; }
        if (key == "comment" || key == "obj_info") continue;
        if (key == "format") {
            std::string f = b;
            format = f == "ascii" ? 0 : (f == "binary_little_endian" ? 1 : (f == "binary_big_endian" ? 2 : -1));
        } else if (key == "element" && n >= 3) {
            elems.push_back(PlyElem{b, (size_t)strtoull(c, nullptr, 10), {}});
        } else if (key == "property" && !elems.empty()) {
            PlyProp p{};
            if (std::string(b) == "list" && n >= 5) { p.is_list = true; p.count_type = ply_type(c); p.type = ply_type(d); p.name = e; }
            else if (n >= 3) { p.is_list = false; p.count_type = 0; p.type = ply_type(b); p.name = c; }
            if (p.type < 0 || p.count_type < 0) return fp_closed = true, ply_error(fp, "PLY header: unknown property type");
            elems.back().props.push_back(p);
        } else {
            return fp_closed = true, ply_error(fp, "PLY header: unexpected line");
        }
    }
    if (!ended || format < 0) return fp_closed = true, ply_error(fp, "PLY header: missing format or end_header");
    PlyReader rd{fp, format};
    std::vector<float> p, nn, uv;
    std::vector<int32_t> tri, quad, face;
    for (const PlyElem& el : elems) {
        // an element record takes at least one byte per property (ascii: a digit; binary: >= 1 byte): a count the file cannot hold is
        // a corrupt or hostile header, rejected before anything is sized from it
        if (el.count > (size_t)std::max(file_size, 0L) || el.count * std::max<size_t>(el.props.size(), 1) > (size_t)std::max(file_size, 0L) || el.count > 0x7fffffffull / 4)
            return fp_closed = true, ply_error(fp, "PLY header: element count exceeds what the file can hold");
        if (el.name == "vertex") {
            p.assign(3 * el.count, 0.0f); nn.assign(3 * el.count, 0.0f); uv.assign(2 * el.count, 0.0f);  // PlyVertex::new, mesh.rs:303-314
            for (size_t i = 0; i < el.count; ++i)
                for (const PlyProp& pr : el.props) {
                    // mesh.rs:316-333: every vertex property must be one of these names AND a Float
                    if (pr.is_list || pr.type != 6) return fp_closed = true, ply_error(fp, "Vertex: Unexpected key/value combination: key: " + pr.name);
                    float v = (float)rd.read(pr.type);
                    const std::string& k = pr.name;
                    if (k == "x") p[3 * i] = v; else if (k == "y") p[3 * i + 1] = v; else if (k == "z") p[3 * i + 2] = v;
                    else if (k == "nx") nn[3 * i] = v; else if (k == "ny") nn[3 * i + 1] = v; else if (k == "nz") nn[3 * i + 2] = v;
                    else if (k == "u" || k == "s" || k == "texture_u" || k == "texture_s") uv[2 * i] = v;
                    else if (k == "v" || k == "t" || k == "texture_v" || k == "texture_t") uv[2 * i + 1] = v;
                    else return fp_closed = true, ply_error(fp, "Vertex: Unexpected key/value combination: key: " + k);
                }
        } else if (el.name == "face") {
            for (size_t i = 0; i < el.count; ++i) {
                std::vector<int32_t> vi, fi;
                for (const PlyProp& pr : el.props) {
                    const std::string& k = pr.name;
                    const bool known = k == "vertex_indices" || k == "vertex_index" || k == "face_indices";
                    // mesh.rs:349-356: only ListInt payloads are accepted (ply-rs: a list of `int` / `int32`)
                    if (!known || !pr.is_list || pr.type != 4) return fp_closed = true, ply_error(fp, "Face: Unexpected key/value combination: key: " + k);
                    const long cnt = (long)rd.read(pr.count_type);
                    if (rd.fail || cnt < 0 || cnt > 1024) return fp_closed = true, ply_error(fp, "PLY: bad list length");
                    std::vector<int32_t>& dst = (k == "face_indices") ? fi : vi;
                    dst.resize((size_t)cnt);
                    for (long q = 0; q < cnt; ++q) dst[(size_t)q] = (int32_t)rd.read(pr.type);
                }
                // mesh.rs:249-276
                if (!((fi.size() > 0 && vi.size() == 0) || (fi.size() == 0 && vi.size() > 0))) return fp_closed = true, ply_error(fp, "PLY face: expected either vertex indices or face indices");
                if (vi.size() == 3) { tri.insert(tri.end(), vi.begin(), vi.end()); }
                else if (vi.size() == 4) { quad.push_back(vi[0]); quad.push_back(vi[1]); quad.push_back(vi[3]); quad.push_back(vi[2]); }
                else if (fi.empty()) return fp_closed = true, ply_error(fp, "Only tris and quads are supported");
                if (!fi.empty()) face.push_back(fi[0]);
            }
        } else {
            return fp_closed = true, ply_error(fp, "Unexpected element: " + el.name);  // mesh.rs:222
        }
        if (rd.fail) return fp_closed = true, ply_error(fp, "PLY: unexpected end of file");
    }
    fclose(fp);
    fp_closed = true;
    const int32_t nv = (int32_t)(p.size() / 3);
    for (int32_t idx : tri) if (idx < 0 || idx >= nv) return ply_error(nullptr, "PLY: triangle vertex index out of range");   // mesh.rs:278-282
    for (int32_t idx : quad) if (idx < 0 || idx >= nv) return ply_error(nullptr, "PLY: quad vertex index out of range");      // mesh.rs:283-287
    auto dup_f = [](const std::vector<float>& v) { float* r = (float*)malloc(std::max<size_t>(v.size(), 1) * sizeof(float)); if (r && !v.empty()) memcpy(r, v.data(), v.size() * sizeof(float)); return r; };
    auto dup_i = [](const std::vector<int32_t>& v) { int32_t* r = (int32_t*)malloc(std::max<size_t>(v.size(), 1) * sizeof(int32_t)); if (r && !v.empty()) memcpy(r, v.data(), v.size() * sizeof(int32_t)); return r; };
    out->n_vertices = (uint32_t)nv;
    out->n_tri_indices = (uint32_t)tri.size(); out->n_quad_indices = (uint32_t)quad.size(); out->n_face_indices = (uint32_t)face.size();
    out->p = dup_f(p); out->n = dup_f(nn); out->uv = dup_f(uv);
    out->tri_indices = dup_i(tri); out->quad_indices = dup_i(quad); out->face_indices = dup_i(face);
    if (!out->p || !out->n || !out->uv || !out->tri_indices || !out->quad_indices || !out->face_indices) { shm_ply_free(out); return SHM_ERR_OUT_OF_MEMORY; }
    return SHM_OK;
}

void shm_ply_free(ShmPlyMesh* m) {
    if (!m) return;
    free(m->p); free(m->n); free(m->uv); free(m->tri_indices); free(m->quad_indices); free(m->face_indices);
    memset(m, 0, sizeof(*m));
}

}  // extern "C"
