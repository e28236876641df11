// tools/sim/pair_step_sim.cpp — development tool (CPU): the both-children traversal step of k_trace5 simulated scalar, beside the oracle's
// reference-shaped loop, to check that hits and visit counters are equal and to size the kernel (steps per ray, stack depth, culled pops).
// Test infrastructure like oracle/: built by `make -C tools/sim`, never linked into the product.
#include "../../oracle/oracle.cpp"

struct PairStats {
    uint64_t rays, visits, pair_steps, pushes, pops_pass, pops_culled, direct_far, leaf_visits, root_leaf;
    uint64_t depth_hist[66];      // max stack depth per ray
    uint64_t push_depth_hist[66]; // depth at which each push lands
    uint64_t mismatches;
};

static inline bool slab(const ShmBvhNode& n, V3 ro, V3 inv_dir, const int* neg, Float& t0o, Float& t1o) {
    // Bounds3f::intersect_p_cached without the t_max test: returns the t_max-independent part and t0
    const Float* b[2] = {n.bmin, n.bmax};
    Float t_min = (b[neg[0]][0] - ro.x) * inv_dir.x;
    Float t_max = (b[1 - neg[0]][0] - ro.x) * inv_dir.x;
    Float ty_min = (b[neg[1]][1] - ro.y) * inv_dir.y;
    Float ty_max = (b[1 - neg[1]][1] - ro.y) * inv_dir.y;
    const Float g = 1.0f + 2.0f * gamma(3);
    t_max *= g; ty_max *= g;
    bool ok = !(t_min > ty_max || ty_min > t_max);
    if (ty_min > t_min) t_min = ty_min;
    if (ty_max < t_max) t_max = ty_max;
    Float tz_min = (b[neg[2]][2] - ro.z) * inv_dir.z;
    Float tz_max = (b[1 - neg[2]][2] - ro.z) * inv_dir.z;
    tz_max *= g;
    ok = ok && !(t_min > tz_max || tz_min > t_max);
    if (tz_min > t_min) t_min = tz_min;
    if (tz_max < t_max) t_max = tz_max;
    t0o = t_min; t1o = t_max;
    return ok && (t_max > 0.0f);
}

extern "C" __attribute__((visibility("default"))) int orc_sim_pair(OrcScene* s, const ShmRay* rays, uint32_t n, PairStats* st) {
    Oracle* o = reinterpret_cast<Oracle*>(s);
    const SceneView& sv = o->sv;
    memset(st, 0, sizeof(*st));
    for (uint32_t i = 0; i < n; ++i) {
        V3 ro = v3(rays[i].o[0], rays[i].o[1], rays[i].o[2]), rd = v3(rays[i].d[0], rays[i].d[1], rays[i].d[2]);
        Float t_max = rays[i].t_max;
        Counters cref; Hit href;
        bvh_intersect(sv, ro, rd, t_max, href, cref);
        // ---- pair-step traversal ----
        V3 inv_dir = v3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
        int neg[3] = {inv_dir.x < 0.0f, inv_dir.y < 0.0f, inv_dir.z < 0.0f};
        uint64_t visits = 0, prims = 0;
        Hit hit; hit.prim = -1;
        struct E { uint32_t node; Float t0; } stack[64];
        int sp = 0, maxd = 0;
        st->rays++;
        uint32_t cur = 0;  // a node known to be hit
        Float t0, t1;
        visits++;
        bool have = slab(sv.nodes[0], ro, inv_dir, neg, t0, t1) && t0 < t_max;
        for (;;) {
            if (have) {
                const ShmBvhNode& nd = sv.nodes[cur];
                if (nd.n_prims > 0) {
                    st->leaf_visits++;
                    for (uint32_t k = 0; k < nd.n_prims; ++k) {
                        prims++;
                        Hit h1;
                        if (prim_intersect(sv, nd.offset + k, ro, rd, t_max, h1)) { hit = h1; t_max = h1.t; }
                    }
                    have = false;
                } else {
                    st->pair_steps++;
                    const uint32_t c0 = cur + 1, c1 = nd.offset;
                    const uint32_t nr = neg[nd.axis] ? c1 : c0, fr = neg[nd.axis] ? c0 : c1;
                    Float n0, n1, f0, f1;
                    visits += 2;
                    const bool hn = slab(sv.nodes[nr], ro, inv_dir, neg, n0, n1) && n0 < t_max;
                    const bool pf = slab(sv.nodes[fr], ro, inv_dir, neg, f0, f1);
                    if (hn) {
                        cur = nr;
                        if (pf) { st->push_depth_hist[sp]++; stack[sp].node = fr; stack[sp].t0 = f0; sp++; st->pushes++; if (sp > maxd) maxd = sp; }
                    } else if (pf && f0 < t_max) { cur = fr; st->direct_far++; }
                    else have = false;
                    continue;
                }
            }
            // pop
            if (sp == 0) break;
            --sp;
            if (stack[sp].t0 < t_max) { cur = stack[sp].node; have = true; st->pops_pass++; }
            else st->pops_culled++;
        }
        st->visits += visits;
        st->depth_hist[maxd]++;
        if (visits != cref.nodes_closest || prims != cref.tris_closest || hit.prim != href.prim || (hit.prim >= 0 && (hit.t != href.t || hit.b0 != href.b0 || hit.b1 != href.b1)))
            st->mismatches++;
    }
    return 0;
}
