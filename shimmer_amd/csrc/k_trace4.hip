// k_trace4.hip — K2 / K3 for scenes made of triangles: BVH traversal with a WAVE-LEVEL RAY POOL in LDS.
//   aggregate.rs:71-139    BvhAggregate::intersect           -> k_trace4<false>
//   aggregate.rs:141-203   BvhAggregate::intersect_predicate -> k_trace4<true>
//
// k_trace3 (k_trace.hip) ties a ray to a lane: while some lanes wait with a postponed leaf, or sit between rays, the others take node
// steps, and the profile of round 1 shows what that costs — 29.8 of 64 lanes active per VALU instruction at 72-80 % issue utilisation
// (profiles/r01_v5_valu_lanes_spp256.txt): the kernel is bound by VALU issue at half-empty waves, not by memory.
// Here a wave owns a pool of P4_POOL ray slots whose whole state lives in LDS (origin, 1/d, t_max, current node, stack pointer, state,
// and the per-slot traversal stack [level][slot]); lanes are not tied to rays. Every iteration the wave picks ONE kind of work —
//   node step   for up to 64 slots that have a node to test (pop if requested -> fetch the 32-B record -> slab test -> push far / enter
//               near, or mark the leaf pending, or request a pop): the reference's step, unchanged, per ray;
//   leaf phase  once `leaf_min` slots have a pending leaf (or nothing else can run): the watertight triangle test for up to 64 of them;
// — compacts the ready slots into a list with two ballots and hands slot list[lane] to each lane. With more slots than lanes a node step
// runs with (nearly) all 64 lanes as long as the queue lasts. Per ray the algorithm is the reference's node for node: node and primitive
// visit counts, hit records and occlusion flags are identical to k_trace3's and to the oracle's.
// One wave per workgroup (the pool is wave-private: no barriers beyond LDS ordering); ~14 KB of LDS per wave -> 11 waves per CU.
#include "wavefront.h"

namespace {

constexpr int P4_POOL = 96;                    // ray slots per wave (a multiple of 32)
constexpr int P4_SPL = (P4_POOL + 63) / 64;    // slots a lane scans: lane, lane + 64, ...
constexpr int P4_LDS_N = K3_LDS_N;             // stack levels held in LDS per slot; deeper levels spill to HBM
enum : uint32_t { S4_IDLE = 0, S4_NODE = 1, S4_POP = 2, S4_LEAF = 3 };
// slot meta word: state [2:0] | has_hit [3] | sp [15:8] | leaf_n [31:16]
#ifndef K4_CHUNK_MAX
#define K4_CHUNK_MAX 1024
#endif

template <bool ANY>
__global__ void __launch_bounds__(WAVE) k_trace4(SceneView sv, const uint32_t* __restrict__ queue, const uint32_t* __restrict__ n_ptr, uint32_t n_direct,
                                                uint32_t* head, const ShmRay* __restrict__ rays, ShmHit* __restrict__ hits,
                                                uint8_t* __restrict__ occluded_out, float4* __restrict__ L, const float4* __restrict__ contrib,
                                                DeviceCounters* counters, uint32_t* __restrict__ spill, int spill_levels, int refill_min, int leaf_min,
                                                int queue_parts) {
    __shared__ float s_o[3][P4_POOL], s_i[3][P4_POOL], s_t[P4_POOL];
    __shared__ uint32_t s_path[P4_POOL], s_cur[P4_POOL], s_meta[P4_POOL];
    __shared__ uint32_t s_stack[P4_LDS_N][P4_POOL];
    __shared__ uint32_t s_list[WAVE];
    const uint32_t lane = threadIdx.x;
    uint32_t* const st_spill = spill + (size_t)blockIdx.x * (size_t)spill_levels * P4_POOL;
    const uint32_t n = n_ptr ? *n_ptr : n_direct;
    const char* __restrict__ node_base = reinterpret_cast<const char*>(sv.nodes);
    const char* __restrict__ prim_base = reinterpret_cast<const char*>(sv.prim_recs);
    uint32_t c_nodes = 0, c_prims = 0, c_rays = 0;
#pragma unroll
    for (int k = 0; k < P4_SPL; ++k)
        if (lane + 64u * k < (uint32_t)P4_POOL) s_meta[lane + 64u * k] = S4_IDLE;
    __syncthreads();

    bool exhausted = false;          // wave-uniform
    uint32_t w_next = 0, w_end = 0;  // wave-uniform private range of the queue
    const uint32_t n_waves = gridDim.x;
    uint32_t chunk = n / (n_waves * 8u);
    chunk = chunk < 64u ? 64u : (chunk > (uint32_t)K4_CHUNK_MAX ? (uint32_t)K4_CHUNK_MAX : chunk);
    chunk = (chunk + 63u) & ~63u;
    // queue partitions as in k_trace3: a wave starts in the partition of the XCD it runs on and steals from the next when dry
    const uint32_t n_parts = (uint32_t)queue_parts;
    const uint32_t part_size = ((n + n_parts * 64u - 1u) / (n_parts * 64u)) * 64u;
    uint32_t part = (n_parts > 1u) ? (__builtin_amdgcn_s_getreg(6164 /* HW_REG_XCC_ID, bits [3:0] */) & (n_parts - 1u)) : 0u;
    uint32_t parts_left = n_parts;

    for (;;) {
        // ---- scan the pool: which slots can take a node step, have a leaf pending, are free ----
        unsigned long long m_node[P4_SPL], m_leaf[P4_SPL], m_idle[P4_SPL];
        int n_node = 0, n_leaf = 0, n_idle = 0;
#pragma unroll
        for (int k = 0; k < P4_SPL; ++k) {
            const uint32_t slot = lane + 64u * k;
            const uint32_t st = slot < (uint32_t)P4_POOL ? (s_meta[slot] & 7u) : 7u;
            m_node[k] = __ballot(st == S4_NODE || st == S4_POP);
            m_leaf[k] = __ballot(st == S4_LEAF);
            m_idle[k] = __ballot(st == S4_IDLE);
            n_node += __popcll(m_node[k]);
            n_leaf += __popcll(m_leaf[k]);
            n_idle += __popcll(m_idle[k]);
        }
        // ---- refill free slots from the wave-private chunk [w_next, w_end); one atomic per `chunk` rays ----
        if (!exhausted && (n_idle >= refill_min || n_node + n_leaf == 0)) {
            while (w_next >= w_end && !exhausted) {
                const uint32_t p_begin = part * part_size;
                const uint32_t p_end = (p_begin < n) ? ((n - p_begin < part_size) ? n : p_begin + part_size) : p_begin;
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(head + part * 32u, chunk);
                base = __shfl(base, 0);
                if (base < p_end - p_begin) {
                    w_next = p_begin + base;
                    w_end = (p_end - w_next < chunk) ? p_end : w_next + chunk;
                } else {
                    part = (part + 1u == n_parts) ? 0u : part + 1u;
                    if (--parts_left == 0u) exhausted = true;
                }
            }
            if (!exhausted) {
                const uint32_t take = min((uint32_t)n_idle, w_end - w_next);
                uint32_t base = 0;
#pragma unroll
                for (int k = 0; k < P4_SPL; ++k) {
                    const uint32_t slot = lane + 64u * k;
                    if ((m_idle[k] >> lane) & 1ull) {
                        const uint32_t rank = base + (uint32_t)__popcll(m_idle[k] & ((1ull << lane) - 1ull));
                        if (rank < take) {
                            const uint32_t qi = w_next + rank;
                            const uint32_t path = queue ? queue[qi] : qi;
                            const float4* rp = reinterpret_cast<const float4*>(rays + path);
                            const float4 r0 = rp[0], r1 = rp[1];
                            s_path[slot] = path;
                            s_o[0][slot] = r0.x; s_o[1][slot] = r0.y; s_o[2][slot] = r0.z;
                            s_i[0][slot] = 1.0f / r0.w; s_i[1][slot] = 1.0f / r1.x; s_i[2][slot] = 1.0f / r1.y;  // aggregate.rs:76-81
                            s_t[slot] = r1.z;
                            s_cur[slot] = 0u;
                            s_meta[slot] = S4_NODE;
                            c_rays++;
                        }
                    }
                    base += (uint32_t)__popcll(m_idle[k]);
                }
                w_next += take;
                __syncthreads();
                continue;  // rescan: the new rays take their first node step together with the others
            }
        }
        if (n_node + n_leaf == 0) {
            if (exhausted) break;
            continue;
        }
        // ---- pick the work of this iteration and compact its slots into s_list ----
        const bool do_leaf = n_leaf >= leaf_min || n_node == 0;
        {
            uint32_t base = 0;
#pragma unroll
            for (int k = 0; k < P4_SPL; ++k) {
                const unsigned long long mk = do_leaf ? m_leaf[k] : m_node[k];
                if ((mk >> lane) & 1ull) {
                    const uint32_t pos = base + (uint32_t)__popcll(mk & ((1ull << lane) - 1ull));
                    if (pos < (uint32_t)WAVE) s_list[pos] = lane + 64u * k;
                }
                base += (uint32_t)__popcll(mk);
            }
        }
        const uint32_t count = min((uint32_t)(do_leaf ? n_leaf : n_node), (uint32_t)WAVE);
        __syncthreads();
        if (lane < count) {
            const uint32_t slot = s_list[lane];
            uint32_t meta = s_meta[slot];
            uint32_t cur = s_cur[slot];
            const V3 ro = v3(s_o[0][slot], s_o[1][slot], s_o[2][slot]);
            Float t_max = s_t[slot];
            if (!do_leaf) {
                // ---- one node step (aggregate.rs:95-137 / 160-199) ----
                uint32_t sp = (meta >> 8) & 0xffu;
                bool go = true;
                if ((meta & 7u) == S4_POP) {
                    if (sp == 0u) {
                        // the stack is empty: the ray is finished (aggregate.rs:128-131)
                        const uint32_t path = s_path[slot];
                        if (ANY) {
                            if (occluded_out) occluded_out[path] = 0;
                            if (L) {
                                float4 l = L[path], c = contrib[path];
                                l.x += c.x; l.y += c.y; l.z += c.z; l.w += c.w;
                                L[path] = l;
                            }
                        } else if (!(meta & 8u)) {
                            float4* hp = reinterpret_cast<float4*>(hits + path);
                            hp[0] = make_float4(__int_as_float(-1), 0.0f, 0.0f, 0.0f);
                            hp[1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                        }
                        meta = S4_IDLE;
                        go = false;
                    } else {
                        sp--;
                        if (sp < (uint32_t)P4_LDS_N) cur = s_stack[sp][slot];
                        else cur = __builtin_nontemporal_load(st_spill + (size_t)(sp - P4_LDS_N) * P4_POOL + slot);
                    }
                }
                if (go) {
                    const V3 inv_dir = v3(s_i[0][slot], s_i[1][slot], s_i[2][slot]);
                    const bool negx = inv_dir.x < 0.0f, negy = inv_dir.y < 0.0f, negz = inv_dir.z < 0.0f;
                    const float4* np = reinterpret_cast<const float4*>(node_base + ((size_t)cur << 5));
                    const float4 na = np[0], nb = np[1];
                    c_nodes++;
                    // Bounds3f::intersect_p_cached (bounding_box.rs:520-563)
                    const Float g = 1.0f + 2.0f * gamma(3);
                    Float t0 = ((negx ? na.w : na.x) - ro.x) * inv_dir.x;
                    Float t1 = ((negx ? na.x : na.w) - ro.x) * inv_dir.x;
                    Float ty0 = ((negy ? nb.x : na.y) - ro.y) * inv_dir.y;
                    Float ty1 = ((negy ? na.y : nb.x) - ro.y) * inv_dir.y;
                    t1 *= g;
                    ty1 *= g;
                    bool hit_box = !(t0 > ty1 || ty0 > t1);
                    if (ty0 > t0) t0 = ty0;
                    if (ty1 < t1) t1 = ty1;
                    Float tz0 = ((negz ? nb.y : na.z) - ro.z) * inv_dir.z;
                    Float tz1 = ((negz ? na.z : nb.y) - ro.z) * inv_dir.z;
                    tz1 *= g;
                    hit_box = hit_box && !(t0 > tz1 || tz0 > t1);
                    if (tz0 > t0) t0 = tz0;
                    if (tz1 < t1) t1 = tz1;
                    hit_box = hit_box && (t0 < t_max) && (t1 > 0.0f);
                    const uint32_t offset = __float_as_uint(nb.z);
                    const uint32_t nmeta = __float_as_uint(nb.w);
                    const uint32_t n_prims = nmeta & 0xffffu;
                    uint32_t st, leaf_n = 0u;
                    if (!hit_box) {
                        st = S4_POP;
                    } else if (n_prims > 0u) {
                        st = S4_LEAF;
                        cur = offset;
                        leaf_n = n_prims;
                    } else {
                        const uint32_t axis = (nmeta >> 16) & 0xffu;
                        const bool neg = (axis == 0u) ? negx : ((axis == 1u) ? negy : negz);
                        const uint32_t far_child = neg ? cur + 1u : offset;  // aggregate.rs:119-127
                        const uint32_t near_child = neg ? offset : cur + 1u;
                        if (sp < (uint32_t)P4_LDS_N) s_stack[sp][slot] = far_child;
                        else st_spill[(size_t)(sp - P4_LDS_N) * P4_POOL + slot] = far_child;
                        sp++;
                        cur = near_child;
                        st = S4_NODE;
                    }
                    s_cur[slot] = cur;
                    meta = st | (meta & 8u) | (sp << 8) | (leaf_n << 16);
                }
                s_meta[slot] = meta;
            } else {
                // ---- postponed leaf phase: the watertight triangle test (triangle.rs:173-302) with the ray's shear recomputed from d ----
                const uint32_t path = s_path[slot];
                const float4* rp = reinterpret_cast<const float4*>(rays + path);
                const float4 r0 = rp[0], r1 = rp[1];
                const RayShear rs = ray_shear(v3(r0.w, r1.x, r1.y));
                const uint32_t leaf_n = meta >> 16;
                bool found_any = false;
                for (uint32_t i = 0; i < leaf_n; ++i) {
                    const uint32_t pslot = cur + i;
                    c_prims++;
                    const float4* pr = reinterpret_cast<const float4*>(prim_base + (size_t)pslot * 48u);
                    const float4 q0 = pr[0], q1 = pr[1], q2 = pr[2];
                    TriangleIntersection ti;
                    if (intersect_triangle_pre(ro, rs, t_max, v3(q0.x, q0.y, q0.z), v3(q0.w, q1.x, q1.y), v3(q1.z, q1.w, q2.x), ti)) {
                        if (ANY) { found_any = true; break; }
                        t_max = ti.t;  // aggregate.rs:105-109
                        float4* hp = reinterpret_cast<float4*>(hits + path);
                        hp[0] = make_float4(__int_as_float((int32_t)pslot), ti.t, ti.b0, ti.b1);
                        hp[1] = make_float4(ti.b2, 0.0f, 0.0f, 0.0f);
                        meta |= 8u;
                    }
                }
                if (ANY && found_any) {
                    if (occluded_out) occluded_out[path] = 1;
                    meta = S4_IDLE;
                } else {
                    s_t[slot] = t_max;
                    meta = S4_POP | (meta & 8u) | (meta & 0xff00u);
                }
                s_meta[slot] = meta;
            }
        }
        __syncthreads();
    }
    unsigned long long w_nodes = c_nodes, w_prims = c_prims, w_rays = c_rays;
    for (int off = 32; off > 0; off >>= 1) {
        w_nodes += __shfl_down(w_nodes, off);
        w_prims += __shfl_down(w_prims, off);
        w_rays += __shfl_down(w_rays, off);
    }
    if (lane == 0 && w_rays) {
        if (ANY) {
            atomicAdd(&counters->rays_any, w_rays);
            atomicAdd(&counters->nodes_any, w_nodes);
            atomicAdd(&counters->tris_any, w_prims);
        } else {
            atomicAdd(&counters->rays_closest, w_rays);
            atomicAdd(&counters->nodes_closest, w_nodes);
            atomicAdd(&counters->tris_closest, w_prims);
        }
    }
}

__global__ void k_reset_heads4(uint32_t* heads) { for (uint32_t i = threadIdx.x; i < 8 * 32; i += blockDim.x) heads[i] = 0; }

}  // namespace

int wf_trace4_pool() { return P4_POOL; }
int wf_trace4_lds_levels() { return P4_LDS_N; }

int wf_launch_trace4(ShmScene* s, bool any, hipStream_t stream, const uint32_t* queue, const uint32_t* n_ptr, uint32_t n_direct, const ShmRay* rays,
                     ShmHit* hits, uint8_t* occluded, float4* L, const float4* contrib) {
    uint32_t* heads = s->d_heads3 + (any ? 8 * 32 : 0);
    uint32_t* spill = any ? s->d_spill4_any : s->d_spill4;
    hipLaunchKernelGGL(k_reset_heads4, dim3(1), dim3(64), 0, stream, heads);
    const int leaf_min = any ? s->leaf_min4_any : s->leaf_min4;
    if (any)
        hipLaunchKernelGGL((k_trace4<true>), dim3(s->trace4_blocks), dim3(WAVE), 0, stream, s->dsv, queue, n_ptr, n_direct, heads, rays, hits, occluded, L,
                           contrib, s->d_counters, spill, s->spill4_levels, s->refill_min4, leaf_min, s->queue_parts);
    else
        hipLaunchKernelGGL((k_trace4<false>), dim3(s->trace4_blocks), dim3(WAVE), 0, stream, s->dsv, queue, n_ptr, n_direct, heads, rays, hits, occluded, L,
                           contrib, s->d_counters, spill, s->spill4_levels, s->refill_min4, leaf_min, s->queue_parts);
    LAUNCH_TRY(any ? "k_trace4<any>" : "k_trace4<closest>");
    return SHM_OK;
}
