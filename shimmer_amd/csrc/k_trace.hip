// k_trace.hip — K2 / K3: BVH traversal of the gfx950 wavefront path tracer (see wavefront.h).
//   aggregate.rs:71-139    BvhAggregate::intersect           -> k_trace5<false, GEN>
//   aggregate.rs:141-203   BvhAggregate::intersect_predicate -> k_trace5<true, GEN>
// One body (trace5_body), the reference's algorithm node for node, executed as UNIFORM steps; GEN = false for scenes made of top-level triangles only, GEN = true for
// scenes that also hold spheres, bilinear patches or instances. (Rounds 1-4 shipped a second body beside it, the one-node step k_trace3 — test the current node, push the
// far child untested — as the kernel of non-triangle scenes and the A/B partner of this one: retired in round 5, when the both-children step learned the other shapes
// and measured 112-171 ms where that kernel took 228-278 on the same frames, profiles/r05_cliff_before.json beside r05_bench_final.json. What it established stays in the
// comments below and in DESIGN.md section 4.)
#include "wavefront.h"

namespace {

// ---------------------------------------------------------------------------------------------
// K2 / K3: BVH traversal. The profile of the first, reference-shaped kernel (one ray per lane from root to done; profiles/r01_v1) showed ~10 of 64 lanes active per
// VALU instruction: issue-bound by divergence, not by HBM (FETCH_SIZE is 3-6x below the algorithmic bytes). Since then:
//  * every loop iteration is one identical step for every lane that has a node to go on from; no nested per-lane loops;
//  * leaf (triangle) tests are POSTPONED: a lane that reached a leaf waits until at least `leaf_min` lanes of its wave have a pending leaf (or no lane can take a node
//    step), then the watertight test runs for all of them at once;
//  * finished lanes are refilled from the queue (one wave-aggregated atomic per chunk) once `refill_min` lanes are idle;
//  * stack levels [0, LDS_N) live in LDS as [level][lane]; any deeper level goes to a per-lane HBM region laid out the same way (the first version kept the LDS window at
//    levels 6..31 and spilled the BOTTOM levels — written at the start of every ray and whenever the traversal comes back near the root —: WRITE_SIZE 7x the algorithmic
//    hit writes, profiles/r01_v4; holding them in registers through select chains was measured too: slower than LDS).
// Node and primitive visit counts equal the reference's in both modes (it is the same algorithm, node for node).
// ---------------------------------------------------------------------------------------------
#ifndef K3_CHUNK_MAX
#define K3_CHUNK_MAX 1024
#endif

// v_cndmask with the per-ray sign held as a 64-bit lane mask in a SCALAR register pair (one bit per lane, rebuilt by three ballots whenever
// a lane takes a new ray): the select costs one VALU instruction and no compare, and — unlike a `bool` per lane, which the compiler keeps
// as such a mask too but has to merge at every control-flow join (3 SALU instructions per mask per join: profiles/r03_k_trace3_isa_before.txt
// counts 123 SALU in the refill section alone) — a wave-uniform integer needs no merging at all.
__device__ __forceinline__ Float sel_mask(unsigned long long mask, Float if_clear, Float if_set) {
    Float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(mask));
    return r;
}

// One-instruction max / min of three (IEEE maxNum / minNum; used only where no operand can be a NaN, see the slab test).
__device__ __forceinline__ Float vmax3(Float a, Float b, Float c) { Float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ Float vmin3(Float a, Float b, Float c) { Float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
#define K3_PARAMS SceneView sv, const uint32_t* __restrict__ queue, const uint32_t* __restrict__ n_ptr, uint32_t n_direct, uint32_t* head,                    \
                  const ShmRay* __restrict__ rays, ShmHit* __restrict__ hits, uint8_t* __restrict__ occluded_out, float4* __restrict__ L,                 \
                  const float4* __restrict__ contrib, DeviceCounters* counters, uint32_t* __restrict__ spill, int spill_levels, int refill_min, int leaf_min, \
                  int queue_parts, int rays_per_lane, int hit16
#define K3_ARGS sv, queue, n_ptr, n_direct, head, rays, hits, occluded_out, L, contrib, counters, spill, spill_levels, refill_min, leaf_min, queue_parts, rays_per_lane, hit16

// ---------------------------------------------------------------------------------------------
// k_trace5: the BOTH-CHILDREN step (round 4), for scenes made of triangles only. Same algorithm, node for node — the reference tests both children of
// every interior node it enters exactly once too (aggregate.rs:110-135: the near child next, the far child when it is popped) — but a step starts from a
// node that is already known to be hit, fetches the 64-byte block of its two children (sibling pairs share one block on the device, render.hip) and runs
// both slab tests at once. A ray's chain of dependent fetches is half as long (S3: 28-33 block fetches instead of 56-67 node fetches per ray), and the
// stack sees a quarter of the traffic: a far child is only pushed when the ray can hit its box at all (7-8 pushes per ray instead of 28-33).
//  * What makes the early test of the far child exact: the slab distances depend on the ray and the box only; the one thing that changes until the
//    reference would have tested it is t_max (a closer hit found in between). So the step evaluates everything but `t0 < t_max` and the stack entry
//    carries t0 — {link word, t0}, 8 bytes — for that compare to be made at the pop, against the t_max of THAT moment (t0 may be a NaN for an
//    irregular ray: the compare is false then, as in the reference's chain). A far child that fails the t_max-independent part is never pushed: the
//    reference would pop it, test it and drop it — one node visit, counted here at the step. Hits, tie order and both visit counters equal the oracle's
//    (tools/sim/pair_step_sim.cpp is the scalar model of this body; tests/test_gpu_parity.py compares the kernels themselves).
//  * any-hit: t_max never changes, so the far child's whole test is known at the step and no t0 is needed; but a ray ends at its first hit and the
//    reference never visits what is still on its stack then — a far child that already failed ("phantom") must count only if the traversal gets back to
//    it. Phantoms are not pushed: a lane counts those above its newest stack entry in a register, an entry carries the count of those below it in its
//    second word, and a pop adds what it passes. Same totals as the reference's loop for every ray (same sim, ANY mode).
//  * a lane's whole state is ONE register: `cur` is the link word of the node it goes on from (interior: fetch its children; leaf: bit 31, pending triangle
//    test) or one of three values no link word can take (axis 3): idle, pop pending, done.
//  * closest-hit writes the hit record when it FINDS a closer hit (1.3 stores per hit ray) instead of carrying prim / b0 / b1 / b2 in registers until the
//    ray retires; t is t_max itself; a miss is written at retirement.
// ---------------------------------------------------------------------------------------------
#ifdef K5_CENSUS
// development build (-DK5_CENSUS): per-phase lane census of the closest-hit kernel, printed by wf_trace_census() at scene destruction
__device__ unsigned long long g_census[32];
#define CENSUS(i, v) do { if (ANY == (K5_CENSUS == 2)) c_census[i] += (unsigned long long)(v); } while (0)  // -DK5_CENSUS=1: closest-hit, =2: any-hit
#else
#define CENSUS(i, v) do { } while (0)
#endif
enum : uint32_t { CUR_IDLE = 0x7fffffffu, CUR_POP = 0x7ffffffeu, CUR_DONE = 0x7ffffffdu, CUR_FIRST_SPECIAL = 0x60000000u,
                  CUR_MARKER = 0x60000000u };  // (GEN: CUR_MARKER | leaf slot of the instance — the stack entry that leads back out of it; never a lane's `cur`: slots are below 2^27)
constexpr uint32_t SGN_RAY = 15u, SGN_HIT = 16u, SGN_HIT_INSIDE = 32u, SGN_INST_SHIFT = 6u;  // trace5_body's `sgn` word (see there)

// GEN (round 5): scenes that hold anything but top-level triangles — spheres, bilinear patches, TransformedPrimitives. The node step is shape-agnostic and unchanged, and so
// is the triangle leaf phase, but for one select: a lane whose record turns out to be no triangle PARKS — bit 30 of its link word, "this slot's test is pending" — and the
// wave collects such lanes as it collects pending leaves: when `other_min` of them wait (or nothing else can run), one dense round runs Sphere::intersect (sphere.rs:95-196),
// BilinearPatch::intersect (bilinear_patch.rs:144-236) or the way into an instance (primitive.rs:158-176: a marker {CUR_MARKER | the instance's slot, t_max outside |
// phantoms below} goes on the stack, the ray is taken into the instance's space, the instance's root is tested by the reference's chain as a refill tests the scene's;
// popping the marker brings the outer ray back and names the instance in the hit record if the closest hit was found inside). Same primitives in the same order per ray as
// the reference's loop; postponing a test changes nothing a ray can see (as for triangles).
// The loop carries NO register the triangle kernel does not, and the heavy arithmetic of that round does not raise the loop's register pressure: the ray's direction is read
// back from the ray array where it is needed (inside an instance: through the instance's matrix again), the instance's index lives in the upper bits of `sgn`, its slot in
// the marker — and the ray state the loop keeps (origin, reciprocals, shear: 44 bytes) is WRITTEN to a per-lane save area in HBM before a quadric / patch test and read back
// behind it, so that nothing but path, t_max, stack pointer and link word is live across ~1 000 instructions of interval arithmetic (inlined with the state live, they made
// the loop itself spill; as real calls, the calling convention's caller-saved registers did the same). An instance's entry saves the OUTER ray state the same way: leaving is
// three loads, not a second ray set-up. The hit record is the ABI's 32-byte ShmHit (t, phi and the instance ride along).
// STRICT (any-hit, scenes with instances, ShmRenderParams::disable_reference_quirks; round 6): TransformedPrimitive::intersect_predicate maps the ray as intersect does —
// apply_ray_inverse, t_max shrinking with the origin's error step — where the reference writes the FORWARD apply_ray (primitive.rs:173-176: instanced objects cast their
// shadows from somewhere else; tests/test_instancing.py). The outer t_max rides in the spare word of save area 0. Own instantiations (k_trace5_any_strict): the
// reference-exact kernels are compiled without a trace of it.
template <bool ANY, bool GEN, int LDS_N, bool SAVE_LDS = false, bool STRICT = false>
__device__ __forceinline__ void trace5_body(const SceneView& sv, const uint32_t* __restrict__ queue, const uint32_t* __restrict__ n_ptr,
                                            uint32_t n_direct, uint32_t* head, const ShmRay* __restrict__ rays,
                                            ShmHit* __restrict__ hits, uint8_t* __restrict__ occluded_out,
                                            float4* __restrict__ L, const float4* __restrict__ contrib,
                                            DeviceCounters* counters, uint32_t* __restrict__ spill, int spill_levels,
                                            int refill_min, int leaf_min, int queue_parts, int rays_per_lane, int hit16, const uint32_t* __restrict__ big_leaf_n,
                                            float4* gen_save, int other_min, float4* __restrict__ hit2) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) u32x2 lds_u2;
    __shared__ u32x2 lds_stack5[(TRACE_BLOCK / WAVE) * LDS_N * WAVE];
    // SAVE_LDS (the five-wave builds): the save area around a quadric / patch test in LDS ([k][thread], 12 KB per workgroup: six stack levels' worth) instead of
    // the wave's HBM area — in a patch-heavy scene every other round is such a test (S3 as patches 3 740 -> 3 909 Mray/s, profiles/r06_patch_heavy_scenes.txt)
    __shared__ float4 lds_save1[SAVE_LDS ? 3 * TRACE_BLOCK : 1];
    const uint32_t lane = threadIdx.x & (WAVE - 1);
    const uint32_t wave_in_block = threadIdx.x / WAVE;
    // the per-lane stack of {link word, t0 | phantom count} entries: levels [0, LDS_N) in LDS as [level][lane], deeper ones in the per-lane HBM spill
    lds_u2* const st_base = (lds_u2*)lds_stack5 + wave_in_block * LDS_N * WAVE + lane;
    lds_u2* top = st_base;
    u32x2* const st_spill_wave = reinterpret_cast<u32x2*>(spill) + ((size_t)blockIdx.x * (TRACE_BLOCK / WAVE) + wave_in_block) * (size_t)spill_levels * WAVE;
    // GEN: this wave's two save areas of 3 float4 per lane, [area][k][lane] (area 0: the outer ray's state while an instance is traversed; area 1: around a quadric / patch test)
    float4* const save_wave = GEN ? gen_save + ((size_t)blockIdx.x * (TRACE_BLOCK / WAVE) + wave_in_block) * (size_t)(6 * WAVE) : nullptr;
    const uint32_t n = n_ptr ? *n_ptr : n_direct;
    // A small queue is traced by a part of the persistent grid: with only a ray or two per resident lane a launch is all ramp — every wave takes one 64-ray chunk, the rays
    // end at different times, and most iterations run with a few lanes. The first blocks of the grid (dispatched breadth first over the CUs) take all the rays, the others
    // return at once; a launch of more than rays_per_lane x the grid's lanes is unchanged. It pays where rays are short (Cornell box, 16 nodes per ray: 16.8 -> 16.1 ms per
    // frame at 4 rays per lane) and not where one ray is a long dependent chain (C4's late bounces, 71 nodes per ray, want every wave they can get: larger values cost
    // there) — profiles/r03_trace_rays_per_lane_sweep.txt.
    const uint32_t per_block = (uint32_t)TRACE_BLOCK * (uint32_t)(rays_per_lane > 0 ? rays_per_lane : 1);
    const uint32_t blocks_wanted = rays_per_lane > 0 ? (n + per_block - 1u) / per_block : gridDim.x;
    const uint32_t active_blocks = blocks_wanted < 64u ? (gridDim.x < 64u ? gridDim.x : 64u) : (blocks_wanted < gridDim.x ? blocks_wanted : gridDim.x);
    if (blockIdx.x >= active_blocks) return;
    const char* __restrict__ node_base = reinterpret_cast<const char*>(sv.nodes);
    const char* __restrict__ prim_base = reinterpret_cast<const char*>(sv.prim_recs);
    // the root's record (wave-uniform: scalar registers for the whole kernel): a new ray tests it where it is taken from the queue
    auto uniform4 = [](float4 v) {  // (readfirstlane: the compiler cannot prove a global load uniform, and eight VGPRs held for the refill path would spill)
        return make_float4(__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v.x))), __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v.y))),
                           __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v.z))), __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v.w))));
    };
    const float4 root_a = uniform4(reinterpret_cast<const float4*>(node_base)[0]), root_b = uniform4(reinterpret_cast<const float4*>(node_base)[1]);
    unsigned long long w_nodes = 0, w_rays = 0, w_prims = 0;  // closest-hit: all three per wave in scalar registers
    uint32_t c_nodes = 0, ph_top = 0;                         // any-hit: node visits per lane (phantoms make the increments differ), phantoms above the top entry
#ifdef K5_CENSUS
    unsigned long long c_census[32] = {0};
#endif

    bool exhausted = false;          // wave-uniform
    uint32_t w_next = 0, w_end = 0;  // wave-uniform private range of the queue
    uint32_t since_other = 0u;       // GEN, wave-uniform: iterations since the wave's last round of parked work
    const uint32_t n_waves = active_blocks * (TRACE_BLOCK / WAVE);
    uint32_t chunk = n / (n_waves * 8u);
    chunk = chunk < 64u ? 64u : (chunk > (uint32_t)K3_CHUNK_MAX ? (uint32_t)K3_CHUNK_MAX : chunk);
    chunk = (chunk + 63u) & ~63u;
    uint32_t path = 0;
    // (Measured and rejected on the refill — round 5: the next refill's queue entries fetched a refill ahead, K2 -0.8 ms, profiles/r05_refill_queue_prefetch.txt; round 6:
    //  the ray records requested beside the node step's fetches with the set-up behind it, and cache-line prefetches of leaf records / pushed far children: K2 +7-13 % /
    //  +4-12 %, profiles/r06_rejected_refill_and_prefetch.txt. The kernel is short of issue slots, not of loads in flight.)
    V3 ro = v3s(0.0f), inv_dir = v3s(0.0f);
    // dir_is_neg (aggregate.rs:76-81) twice: as three wave-wide lane masks in scalar registers for the slab tests' selects (sel_mask), and as three bits of `sgn` for the near /
    // far child choice, where the axis varies per lane; and the lanes whose ray is not "regular" (bit 3 of sgn): a non-finite origin, or a direction component that is 0,
    // infinite or so small that its reciprocal overflows. Only such a ray can turn a slab distance into a NaN ((plane - o) * (1 / d) = 0 * inf, inf * 0, inf - inf), and only
    // with NaNs does the reference's compare-and-select chain differ from max3 / min3 — see the slab test.
    unsigned long long m_negx = 0ull, m_negy = 0ull, m_negz = 0ull, m_irregular = 0ull;
    // any-hit, with the render's deferred contributions: L[path] + contrib[path], summed when the ray is TAKEN (the two loads travel with the ray's) and stored when it
    // ends unoccluded — at the end they were a round trip the whole wave waited for in most iterations (2.4 rays end per iteration); nobody else touches L[path] meanwhile
    float4 l_new = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    uint32_t sgn = 0;  // bits 0-2: dir_is_neg, bit 3: irregular ray, bit 4: a hit has been found (closest: its record is in `hits` already); GEN: bit 5: ... inside the instance
                       // being traversed (closest-hit: t_max stays the inner one when the marker is popped, as primitive.rs:158-171 returns it), bits 6-31: that instance's index + 1
    RayShear rs;
    rs.kx = 0; rs.ky = 1; rs.kz = 2; rs.d = v3s(0.0f); rs.sx = rs.sy = rs.sz = 0.0f;
    Float t_max = 0.0f;
    uint32_t cur = CUR_IDLE;

    // aggregate.rs:76-81 + the ray-constant part of the triangle test (the upper bits of sgn — what has been found so far, the instance — belong to the path, not to the ray)
    auto set_ray = [&](V3 o, V3 d) {
        ro = o;
        inv_dir = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        const bool regular = is_finite(o.x) && is_finite(o.y) && is_finite(o.z) && is_finite(inv_dir.x) && is_finite(inv_dir.y) && is_finite(inv_dir.z) &&
                             inv_dir.x != 0.0f && inv_dir.y != 0.0f && inv_dir.z != 0.0f;
        sgn = (sgn & ~SGN_RAY) | (inv_dir.x < 0.0f ? 1u : 0u) | (inv_dir.y < 0.0f ? 2u : 0u) | (inv_dir.z < 0.0f ? 4u : 0u) | (regular ? 0u : 8u);
        rs = ray_shear(d);
    };
    // a tree's root (aggregate.rs:92-97, first iteration), by the reference's own chain: once per ray and tree, and this lane's sign masks do not exist yet
    auto root_test = [&](const float4 ra, const float4 rb) -> bool {
        const Float g = 1.0f + 2.0f * gamma(3);
        const bool nx = (sgn & 1u) != 0u, ny = (sgn & 2u) != 0u, nz = (sgn & 4u) != 0u;
        Float t0 = ((nx ? ra.w : ra.x) - ro.x) * inv_dir.x;
        Float t1 = ((nx ? ra.x : ra.w) - ro.x) * inv_dir.x;
        const Float ty0 = ((ny ? rb.x : ra.y) - ro.y) * inv_dir.y;
        Float ty1 = ((ny ? ra.y : rb.x) - ro.y) * inv_dir.y;
        t1 *= g;
        ty1 *= g;
        bool ok = !(t0 > ty1 || ty0 > t1);
        if (ty0 > t0) t0 = ty0;
        if (ty1 < t1) t1 = ty1;
        const Float tz0 = ((nz ? rb.y : ra.z) - ro.z) * inv_dir.z;
        Float tz1 = ((nz ? ra.z : rb.y) - ro.z) * inv_dir.z;
        tz1 *= g;
        ok = ok && !(t0 > tz1 || tz0 > t1);
        if (tz0 > t0) t0 = tz0;
        if (tz1 < t1) t1 = tz1;
        return ok && (t0 < t_max) && (t1 > 0.0f);
    };

#ifndef K5_OUTER_IN_REGS
#define K5_OUTER_IN_REGS 0
#endif
    // GEN: the OUTER ray's state while an instance is traversed
    V3 o_ro = v3s(0.0f), o_inv = v3s(0.0f);
    Float o_sx = 0.0f, o_sy = 0.0f, o_sz = 0.0f;
    uint32_t o_kz_sgn = 0u;  // kz | ray bits of sgn << 2
    // GEN: the ray state the loop keeps, out to / back from one of the lane's two save areas
    auto save_ray_state = [&](int area) {
        if (K5_OUTER_IN_REGS && area == 0) {
            o_ro = ro; o_inv = inv_dir; o_sx = rs.sx; o_sy = rs.sy; o_sz = rs.sz; o_kz_sgn = (uint32_t)rs.kz | ((sgn & SGN_RAY) << 2);
            return;
        }
        if (SAVE_LDS && area == 1) {
            lds_save1[threadIdx.x] = make_float4(ro.x, ro.y, ro.z, inv_dir.x);
            lds_save1[TRACE_BLOCK + threadIdx.x] = make_float4(inv_dir.y, inv_dir.z, rs.sx, rs.sy);
            lds_save1[2 * TRACE_BLOCK + threadIdx.x] = make_float4(rs.sz, __int_as_float(rs.kz), __uint_as_float(sgn & SGN_RAY), 0.0f);
            return;
        }
        float4* a = save_wave + (size_t)area * (3 * WAVE) + lane;
        a[0] = make_float4(ro.x, ro.y, ro.z, inv_dir.x);
        a[WAVE] = make_float4(inv_dir.y, inv_dir.z, rs.sx, rs.sy);
        a[2 * WAVE] = make_float4(rs.sz, __int_as_float(rs.kz), __uint_as_float(sgn & SGN_RAY), (ANY && STRICT && area == 0) ? t_max : 0.0f);
    };
    auto restore_ray_state = [&](int area) {
        if (K5_OUTER_IN_REGS && area == 0) {
            ro = o_ro; inv_dir = o_inv; rs.sx = o_sx; rs.sy = o_sy; rs.sz = o_sz;
            rs.kz = (int)(o_kz_sgn & 3u);
            rs.kx = rs.kz == 2 ? 0 : rs.kz + 1;
            rs.ky = rs.kx == 2 ? 0 : rs.kx + 1;
            sgn = (sgn & ~SGN_RAY) | (o_kz_sgn >> 2);
            return;
        }
        asm volatile("" ::: "memory");  // (the loads below must be loads: with the stored values forwarded, the state would be live across what lies in between)
        const float4* a = save_wave + (size_t)area * (3 * WAVE) + lane;
        float4 s0, s1, s2;
        if (SAVE_LDS && area == 1) { s0 = lds_save1[threadIdx.x]; s1 = lds_save1[TRACE_BLOCK + threadIdx.x]; s2 = lds_save1[2 * TRACE_BLOCK + threadIdx.x]; }
        else { s0 = a[0]; s1 = a[WAVE]; s2 = a[2 * WAVE]; }
        ro = v3(s0.x, s0.y, s0.z);
        inv_dir = v3(s0.w, s1.x, s1.y);
        rs.sx = s1.z; rs.sy = s1.w; rs.sz = s2.x;
        rs.kz = __float_as_int(s2.y);
        rs.kx = rs.kz == 2 ? 0 : rs.kz + 1;
        rs.ky = rs.kx == 2 ? 0 : rs.kx + 1;
        sgn = (sgn & ~SGN_RAY) | __float_as_uint(s2.z);
        if (ANY && STRICT && area == 0) t_max = s2.w;  // (the outer ray's extent: apply_ray_inverse shortened the inner one)
    };

    // Bounds3f::intersect_p_cached (bounding_box.rs:520-563) without its `t0 < t_max`: the t_max-independent part of the verdict, and t0
    auto slab = [&](const float4 na, const float4 nb, Float& t0_out) -> bool {
        const Float g = 1.0f + 2.0f * gamma(3);
        const Float tx0 = (sel_mask(m_negx, na.x, na.w) - ro.x) * inv_dir.x;
        Float tx1 = (sel_mask(m_negx, na.w, na.x) - ro.x) * inv_dir.x;
        const Float ty0 = (sel_mask(m_negy, na.y, nb.x) - ro.y) * inv_dir.y;
        Float ty1 = (sel_mask(m_negy, nb.x, na.y) - ro.y) * inv_dir.y;
        const Float tz0 = (sel_mask(m_negz, na.z, nb.y) - ro.z) * inv_dir.z;
        Float tz1 = (sel_mask(m_negz, nb.y, na.z) - ro.z) * inv_dir.z;
        if (m_irregular == 0ull) {
            // Every ray of the wave is regular: no slab distance is a NaN (the node bounds are finite: shm_bvh_build refuses others), and then the reference's chain —
            // !(tx0 > ty1 || ty0 > tx1), t0 = max, t1 = min, !(t0 > tz1 || tz0 > t1), ..., t0 < t_max, t1 > 0 — is max3(near) <= min3(far * g) && min3 > 0 (&& max3 < t_max):
            // the chain tests every near_i against every far_j of ANOTHER axis; the same-axis pairs hold by construction (near_i <= far_i before the * g; a far_i that * g
            // pushed below near_i is negative, and then t1 > 0 fails on both sides). Two instructions and three compares instead of ten compares and four selects.
            // One product instead of three: g > 1 and rounding is monotonic, so min3(a g, b g, c g) = min3(a, b, c) g bit for bit
            // (the far distances are not NaNs here), and its sign is the sign of min3(a, b, c).
            const Float t0 = vmax3(tx0, ty0, tz0), t1 = vmin3(tx1, ty1, tz1);
            t0_out = t0;
            return (t0 <= t1 * g) && (t1 > 0.0f);
        }
        // bounding_box.rs:520-563 statement for statement (a lane may hold NaNs: comparisons with them are false, the selects keep them)
        tx1 *= g;
        ty1 *= g;
        tz1 *= g;
        Float t0 = tx0, t1 = tx1;
        bool ok = !(t0 > ty1 || ty0 > t1);
        if (ty0 > t0) t0 = ty0;
        if (ty1 < t1) t1 = ty1;
        ok = ok && !(t0 > tz1 || tz0 > t1);
        if (tz0 > t0) t0 = tz0;
        if (tz1 < t1) t1 = tz1;
        t0_out = t0;
        return ok && (t1 > 0.0f);
    };
    auto push = [&](uint32_t link, uint32_t word1) {
        // two different store flavours, so that the compiler cannot merge them into one flat_store of a selected pointer
        if (top < st_base + LDS_N * WAVE) *top = u32x2{link, word1};
        else st_spill_wave[(size_t)(top - (st_base + LDS_N * WAVE)) + lane] = u32x2{link, word1};
        top += WAVE;
    };

    // Queue partitions: the queue is cut into `queue_parts` contiguous ranges, each with its own head word (own 128-B line). A wave starts in the partition of the XCD it
    // runs on — queue order is image order (pix_group), so one XCD's L2 serves one image region's part of the BVH, and 8 head words see 1/8 of the atomics each (one word
    // saturates near 88 dequeues/us: MI355X_MICROARCH.md "dequeue") — and moves on to the next partition when its own has run dry. The chunk a wave claims with one atomic
    // is large enough that the head word sees few atomics and small enough that the last chunks balance across the resident waves.
    const uint32_t n_parts = (uint32_t)queue_parts;
    const uint32_t part_size = ((n + n_parts * 64u - 1u) / (n_parts * 64u)) * 64u;
    uint32_t part = (n_parts > 1u) ? (__builtin_amdgcn_s_getreg(6164 /* HW_REG_XCC_ID, bits [3:0] */) & (n_parts - 1u)) : 0u;
    uint32_t parts_left = n_parts;
#ifdef K5_CENSUS
    const unsigned long long t_loop0 = __builtin_readcyclecounter();
#endif
    for (;;) {
        // ---- refill idle lanes from the wave-private chunk [w_next, w_end); one atomic per `chunk` rays ----
        const unsigned long long idle = __ballot(cur == CUR_IDLE);
        if (GEN) since_other += 1u;
        if (idle != 0ull) {
            const int n_idle = __popcll(idle);
            // GEN: a lane parked on a non-triangle test takes no node step either — the refill threshold counts the lanes that cannot, not only the idle ones (with one
            // sphere in 4.3 M triangles a parked lane waits ~190 iterations for company: 4 parked + 17 idle lanes left 32 of 64 at a node, against 37 in the triangle scene)
#ifndef K5_REFILL_COUNTS_PARKED
#define K5_REFILL_COUNTS_PARKED 1
#endif
            // (... once they have waited: where parked work is frequent — every ray of an instanced object parks twice — a round comes every ~16 iterations and
            //  early refills only break its batches: S3 instanced 171 -> 182 ms when every parked lane counted)
            const int n_unavailable = (GEN && K5_REFILL_COUNTS_PARKED && since_other > 24u) ? n_idle + __popcll(__ballot(cur >= (LINK_LEAF | LINK_OTHER))) : n_idle;
            if (!exhausted && (n_unavailable >= refill_min || idle == ~0ull)) {
                while (w_next >= w_end && !exhausted) {
                    const uint32_t p_begin = part * part_size;
                    const uint32_t p_end = (p_begin < n) ? ((n - p_begin < part_size) ? n : p_begin + part_size) : p_begin;
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(head + part * 32u, chunk);
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (base < p_end - p_begin) {
                        w_next = p_begin + base;
                        w_end = (p_end - w_next < chunk) ? p_end : w_next + chunk;
                    } else {
                        part = (part + 1u == n_parts) ? 0u : part + 1u;
                        if (--parts_left == 0u) exhausted = true;
                    }
                }
                if (!exhausted) {
#ifdef K5_CENSUS
                    const unsigned long long t_refill0 = __builtin_readcyclecounter();
#endif
                    const uint32_t take = min((uint32_t)n_idle, w_end - w_next);
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                    if (cur == CUR_IDLE) {
                        if (rank < take) {
                            const uint32_t qi = w_next + rank;
                            path = queue ? queue[qi] : qi;
                            const float4* rp = reinterpret_cast<const float4*>(rays + path);
                            const float4 r0 = rp[0], r1 = rp[1];
#ifndef K5_L_AT_REFILL
#define K5_L_AT_REFILL 1  // (0: L and the contribution are loaded when the ray ends: A/B)
#endif
                            if (ANY && K5_L_AT_REFILL && L) {
                                const float4 l = L[path], c = contrib[path];
                                l_new = make_float4(l.x + c.x, l.y + c.y, l.z + c.z, l.w + c.w);
                            }
                            sgn = 0u;
                            set_ray(v3(r0.x, r0.y, r0.z), v3(r0.w, r1.x, r1.y));
                            t_max = r1.z;
                            top = st_base;
                            if (ANY) ph_top = 0u;
                            // the root: tested where the ray is taken from the queue
                            cur = root_test(root_a, root_b) ? __float_as_uint(root_b.z) : (uint32_t)CUR_POP;  // (a miss pops the empty stack: done, retired below)
                            if (ANY) c_nodes += 1u;
                        }
                    }
                    w_next += take;
                    w_rays += take;
                    if (!ANY) w_nodes += take;
                    CENSUS(11, 1); CENSUS(12, take);
                    m_negx = __ballot((sgn & 1u) != 0u);
                    m_negy = __ballot((sgn & 2u) != 0u);
                    m_negz = __ballot((sgn & 4u) != 0u);
                    m_irregular = __ballot((sgn & 8u) != 0u);
#ifdef K5_CENSUS
                    CENSUS(13, __builtin_readcyclecounter() - t_refill0);  // (the ballots above consume the loaded rays: the refill's memory wait is inside)
#endif
                }
            }
            if (__ballot(cur != CUR_IDLE) == 0ull) {
                if (exhausted) break;
                continue;  // private chunk was empty: fetch the next one
            }
        }
        // ---- one uniform step: every lane that stands on an interior node fetches the block of its two children and tests both (aggregate.rs:92-135) ----
        const bool at_node = cur < (uint32_t)CUR_FIRST_SPECIAL;
        if (!ANY) w_nodes += 2ull * (unsigned long long)__popcll(__ballot(at_node));
        CENSUS(0, 1); CENSUS(1, __popcll(__ballot(at_node))); CENSUS(2, __popcll(__ballot((cur & 0xC0000000u) == 0x80000000u))); CENSUS(21, __popcll(__ballot(cur >= 0xC0000000u))); CENSUS(3, __popcll(__ballot(cur == CUR_IDLE)));
        CENSUS(4, __popcll(__ballot(cur == CUR_POP))); CENSUS(10, __ballot(at_node) != 0ull ? 1 : 0);
        CENSUS(22, (__ballot(at_node) != 0ull && __popcll(__ballot(at_node)) <= 8) ? 1 : 0);  // (a VALU instruction with 8 or fewer lanes on costs 4.5 x one with 9: profiles/r05_valu_exec.txt)
        if (at_node) {
            // near child first (aggregate.rs:119-127: dir_is_neg[axis] picks it); pairs start at even indices, so the sibling's record is at byte offset ^ 32
            const uint32_t neg = (sgn >> (cur >> LINK_AXIS_SHIFT)) & 1u;
            const uint32_t off_near = (cur << 5) | (neg << 5);  // (cur << 5 drops the axis bits; a 32-bit byte offset on the uniform base)
            const float4* pn = reinterpret_cast<const float4*>(node_base + off_near);
            const float4* pf = reinterpret_cast<const float4*>(node_base + (off_near ^ 32u));
            const float4 na = pn[0], nb = pn[1], fa = pf[0], fb = pf[1];
            Float t0n, t0f;
            const bool pre_n = slab(na, nb, t0n), pre_f = slab(fa, fb, t0f);
            const bool hit_n = pre_n && (t0n < t_max), hit_f = pre_f && (t0f < t_max);
            const uint32_t link_n = __float_as_uint(nb.z), link_f = __float_as_uint(fb.z);
            if (ANY) {
                // t_max is fixed: the far child's verdict is final now. Visits: the near child now; the far child now if the traversal turns to it at
                // once (the near child missed: the reference's very next pop), at its pop if it is pushed, and as a phantom otherwise
                c_nodes += hit_n ? 1u : 2u;
                if (hit_n && hit_f) { push(link_f, ph_top); ph_top = 0u; }
                else if (hit_n) ph_top += 1u;
            } else {
                if (hit_n && pre_f) push(link_f, __float_as_uint(t0f));  // aggregate.rs:119-127: the far child waits; what is left of its test is `t0 < t_max`
            }
            cur = hit_n ? link_n : (hit_f ? link_f : (uint32_t)CUR_POP);
        }
        // ---- postponed leaf phase: lanes standing on a leaf wait until enough of them do (or nothing else can run) ----
        constexpr uint32_t NOT_A_TRIANGLE = PRIM_SPHERE_BIT | PRIM_PATCH_BIT | PRIM_INSTANCE_BIT;
        constexpr uint32_t LEAF_KIND = LINK_LEAF | LINK_OTHER;  // (GEN: a leaf link with bit 30 set is a parked non-triangle test, below)
        constexpr uint32_t LINK_LEAVE = LEAF_KIND | LINK_INDEX_MASK;  // (... and this one — no slot — a lane parked on its way back out of an instance)
        const unsigned long long leaf_mask = GEN ? __ballot((cur & LEAF_KIND) == LINK_LEAF) : __ballot((int32_t)cur < 0);
        if (leaf_mask != 0ull) {
            const unsigned long long node_mask = __ballot(cur < (uint32_t)CUR_FIRST_SPECIAL);
            if (__popcll(leaf_mask) >= leaf_min || node_mask == 0ull) {
                // ONE primitive per lane and phase: a leaf of several primitives (coincident centroids, aggregate.rs:345-356: the room's quads) stays pending with
                // its link word advanced to the next one — same primitives, same order, same t_max updates as the reference's inner loop, but every round of
                // triangle tests runs with all the pending lanes (as an inner loop, the second rounds ran for the few lanes on such leaves: 0.62 extra rounds
                // per phase, 13 % of the kernel's instructions on the headline frame)
                const bool on_leaf = GEN ? ((cur & LEAF_KIND) == LINK_LEAF) : ((int32_t)cur < 0);
                if (!GEN) w_prims += (unsigned long long)__popcll(__ballot(on_leaf));
                CENSUS(5, 1); CENSUS(6, 1); CENSUS(7, __popcll(__ballot(on_leaf))); CENSUS(23, __popcll(__ballot(on_leaf)) <= 8 ? 1 : 0);
                bool tested = false;  // (GEN: the lanes whose record was a triangle; the others park)
                if (on_leaf) {
                    const uint32_t slot = cur & LINK_INDEX_MASK;
                    uint32_t leaf_n = (cur >> LINK_COUNT_SHIFT) & LINK_COUNT_MAX;
                    if (leaf_n == LINK_COUNT_MAX) leaf_n = big_leaf_n[slot];  // (rare: 7 or more primitives in one leaf; the table holds the count from each slot on)
                    const float4* pr = reinterpret_cast<const float4*>(prim_base + (size_t)slot * sizeof(PrimRec));
                    const float4 q0 = pr[0], q1 = pr[1], q2 = pr[2];
                    const uint32_t kind = __float_as_uint(q2.y);
                    tested = !GEN || (kind & NOT_A_TRIANGLE) == 0u;
                    // (the precomputed degeneracy flag is applied to the RESULT: tested first, the compiler fetched the flag word, waited, and only then fetched the vertices — two dependent
                    //  round trips per leaf; degenerate triangles are rare and the test itself has no side effect; GEN: so is "this record is no triangle at all" — the test has no side effect)
                    TriangleIntersection ti;
                    bool got = intersect_triangle_nondegenerate(ro, rs, t_max, v3(q0.x, q0.y, q0.z), v3(q0.w, q1.x, q1.y), v3(q1.z, q1.w, q2.x), ti);
                    got = got && !(kind & PRIM_DEGENERATE_BIT) && tested;
                    if (got) {
                        sgn |= GEN ? (SGN_HIT | SGN_HIT_INSIDE) : SGN_HIT;
                        if (!ANY) {
                            t_max = ti.t;  // aggregate.rs:105-109 shrinks the ray to the hit
                            if (hit16) {  // (GEN: the split form — a triangle hit is its 16-byte record; only a sphere / patch hit has a second one, below)
                                reinterpret_cast<float4*>(hits)[path] = make_float4(__int_as_float((int32_t)slot), ti.b0, ti.b1, ti.b2);
                            } else {
                                float4* hp = reinterpret_cast<float4*>(hits + path);
                                hp[0] = make_float4(__int_as_float((int32_t)slot), ti.t, ti.b0, ti.b1);
                                hp[1] = make_float4(ti.b2, 0.0f, 0.0f, 0.0f);  // (GEN: a hit inside an instance gets the instance's slot when the marker is popped)
                            }
                        }
                    }
                    leaf_n -= 1u;
                    if (GEN && !tested) cur |= LINK_OTHER;  // no triangle: the lane parks on this slot until the wave runs its other tests
                    else if (ANY && got) cur = CUR_DONE;    // intersect_predicate returns at its first hit (aggregate.rs:160-166)
                    else if (leaf_n == 0u) cur = CUR_POP;
                    else cur = LINK_LEAF | ((leaf_n < LINK_COUNT_MAX ? leaf_n : LINK_COUNT_MAX) << LINK_COUNT_SHIFT) | (slot + 1u);
                }
                if (GEN) w_prims += (unsigned long long)__popcll(__ballot(tested));
            }
        }
        // ---- GEN: the parked non-triangle tests, run like the leaf phase — when enough lanes wait for one, or when nothing else can run ----
        if (GEN) {
            const unsigned long long other_mask = __ballot(cur >= LEAF_KIND);
            if (__builtin_expect(other_mask != 0ull, 0)) {  // (unlikely: block frequencies are what the register allocator places its spill code by)
                const bool busy_elsewhere = __ballot(cur < (uint32_t)CUR_FIRST_SPECIAL || (cur & LEAF_KIND) == LINK_LEAF) != 0ull;
                if (__builtin_expect(__popcll(other_mask) >= other_min || !busy_elsewhere, 0)) {
                    w_prims += (unsigned long long)__popcll(__ballot(cur >= LEAF_KIND && cur != LINK_LEAVE));
                    CENSUS(15, 1); CENSUS(16, __popcll(other_mask)); CENSUS(17, busy_elsewhere ? 0 : 1); CENSUS(27, __popcll(other_mask) <= 8 ? 1 : 0);
                    since_other = 0u;
                    bool entered = false, root_tested = false;
                    if (cur == LINK_LEAVE) {
                        // a lane that came back out of an instance (its marker was popped, below): the outer ray's state, as the entry left it; then it pops on
                        restore_ray_state(0);
                        cur = CUR_POP;
                        entered = true;  // (its ray changed)
                    } else if (cur >= LEAF_KIND && ((cur >> LINK_COUNT_SHIFT) & LINK_COUNT_MAX) == 0u) {
                        // TransformedPrimitive (primitive.rs:158-176), reached straight from the node step: the link word of its leaf holds the instance's index (render.hip,
                        // upload). ONE round trip — the ray's direction, the instance's matrix and its tree's root record are fetched side by side — then: the outer ray's
                        // state to save area 0, a marker on the stack — with t_max as it is outside (closest-hit) or the phantoms below (any-hit) —, the ray into the
                        // instance's space — apply_ray_inverse for intersect, the FORWARD apply_ray for intersect_predicate, as the reference writes them — and the
                        // instanced aggregate's root tested (BvhAggregate::intersect's first node)
                        const uint32_t idx = cur & LINK_INDEX_MASK;
                        const ShmInstance& in = sv.instances[idx];
                        const float4* rp = reinterpret_cast<const float4*>(rays + path);
                        const float4 r0 = rp[0], r1 = rp[1];
                        const float4* rn = reinterpret_cast<const float4*>(sv.inst_roots + idx);
                        const float4 ra = rn[0], rb = rn[1];
                        const uint32_t slot = in.pad[0];
                        const V3 rd = v3(r0.w, r1.x, r1.y);
                        save_ray_state(0);
                        if (ANY) { push((uint32_t)CUR_MARKER | slot, ph_top); ph_top = 0u; }
                        else push((uint32_t)CUR_MARKER | slot, __float_as_uint(t_max));
                        sgn = (sgn & (SGN_RAY | SGN_HIT)) | ((idx + 1u) << SGN_INST_SHIFT);
                        Ray r;
                        if (ANY && !STRICT) { Ray w; w.o = ro; w.d = rd; r = xf_ray(in.render_from_primitive, w); }
                        else r = xf_ray_inverse(in.primitive_from_render, ro, rd, t_max);
                        set_ray(r.o, r.d);
                        cur = root_test(ra, rb) ? __float_as_uint(rb.z) : (uint32_t)CUR_POP;  // (a miss pops the marker: back out)
                        if (ANY) c_nodes += 1u;
                        entered = true;
                        root_tested = true;
                    } else if (cur >= LEAF_KIND) {
                        uint32_t slot = cur & LINK_INDEX_MASK;
                        // (opaque to the optimiser: a lane may have parked in THIS iteration's leaf phase, and the compiler, seeing the same address, kept the record that
                        //  phase loaded alive for this one — 40 bytes of scratch stores per lane in EVERY leaf phase: 117 GB per 64-spp frame, K2 28 -> 42 ms on the
                        //  triangle-only headline scene traced by this kernel)
                        asm volatile("" : "+v"(slot));
                        uint32_t leaf_n = (cur >> LINK_COUNT_SHIFT) & LINK_COUNT_MAX;
                        if (leaf_n == LINK_COUNT_MAX) leaf_n = big_leaf_n[slot];
                        const float4* pr = reinterpret_cast<const float4*>(prim_base + (size_t)slot * sizeof(PrimRec));
                        const float4 q0 = pr[0], q1 = pr[1], q2 = pr[2], q3 = pr[3];  // (q3: a patch's p11.z in its third word — the whole 64-byte line in one round trip)
                        const uint32_t kind = __float_as_uint(q2.y);
                        // the ray's direction, read back from the ray array (the loop keeps the origin, the reciprocals and the shear); inside an instance it goes through that
                        // instance's matrix again — the product apply_ray_inverse / apply_ray made when the instance was entered, bit for bit
                        const float4* rp = reinterpret_cast<const float4*>(rays + path);
                        const float4 r0 = rp[0], r1 = rp[1];
                        V3 rd = v3(r0.w, r1.x, r1.y);
                        if (sgn >> SGN_INST_SHIFT) {
                            const ShmInstance& in = sv.instances[(sgn >> SGN_INST_SHIFT) - 1u];
                            rd = xf_vector((ANY && !STRICT) ? in.render_from_primitive : in.primitive_from_render, rd);
                        }
                        {
                            // (a sphere or a bilinear patch: an instance never parks here — it is alone in its leaf, and such a leaf's link word names it, above)
                            save_ray_state(1);  // (... and nothing of it is live across the test; the five-wave build WITHOUT this save / restore: 40 / 14 spilled VGPRs instead of 10 / 0, S3 as patches 3 736 -> 3 572 Mray/s)
                            bool got;
                            Float t_hit, h0, h1, h2, h_phi;
                            if (kind & PRIM_SPHERE_BIT) {
                                // Sphere::intersect (sphere.rs:95-196); the hit record carries p_obj and phi
                                QuadricIntersection qi;
                                qi.t_hit = 0.0f; qi.p_obj = v3s(0.0f); qi.phi = 0.0f;
                                got = sphere_basic_intersect(sv.spheres[kind & PRIM_INDEX_MASK], ro, rd, t_max, qi);
                                t_hit = qi.t_hit; h0 = qi.p_obj.x; h1 = qi.p_obj.y; h2 = qi.p_obj.z; h_phi = qi.phi;
                            } else {
                                // BilinearPatch::intersect (bilinear_patch.rs:144-236): the record holds p00, p10, p01; (u, v) go in b0, b1
                                BilinearIntersection bi;
                                bi.t = 0.0f; bi.u = 0.0f; bi.v = 0.0f;
                                got = blp_intersect(ro, rd, t_max, v3(q0.x, q0.y, q0.z), v3(q0.w, q1.x, q1.y), v3(q1.z, q1.w, q2.x), v3(q2.z, q2.w, q3.z), bi);  // (p11: in the record's spare words, flatten.h)
                                t_hit = bi.t; h0 = bi.u; h1 = bi.v; h2 = 0.0f; h_phi = 0.0f;
                            }
                            restore_ray_state(1);
                            if (got) {
                                sgn |= SGN_HIT | SGN_HIT_INSIDE;
                                if (!ANY) {
                                    t_max = t_hit;
                                    if (hit16) {  // the split form (scenes without instances): bit 30 of the primitive word says "read the second record" (t, phi)
                                        reinterpret_cast<float4*>(hits)[path] = make_float4(__int_as_float((int32_t)(slot | HIT_HAS_SECOND)), h0, h1, h2);
                                        hit2[path] = make_float4(t_hit, h_phi, 0.0f, 0.0f);
                                    } else {
                                        float4* hp = reinterpret_cast<float4*>(hits + path);
                                        hp[0] = make_float4(__int_as_float((int32_t)slot), t_hit, h0, h1);
                                        hp[1] = make_float4(h2, h_phi, 0.0f, 0.0f);
                                    }
                                }
                            }
                            leaf_n -= 1u;
                            if (ANY && got) cur = CUR_DONE;
                            else if (leaf_n == 0u) cur = CUR_POP;
                            else cur = LINK_LEAF | ((leaf_n < LINK_COUNT_MAX ? leaf_n : LINK_COUNT_MAX) << LINK_COUNT_SHIFT) | (slot + 1u);
                        }
                    }
                    CENSUS(18, __popcll(__ballot(entered)));
                    if (__ballot(entered) != 0ull) {  // a lane's ray, and with it its signs, changed
                        if (!ANY) w_nodes += (unsigned long long)__popcll(__ballot(root_tested));
                        m_negx = __ballot((sgn & 1u) != 0u);
                        m_negy = __ballot((sgn & 2u) != 0u);
                        m_negz = __ballot((sgn & 4u) != 0u);
                        m_irregular = __ballot((sgn & 8u) != 0u);
                    }
                }
            }
        }
        // ---- pop: a lane whose two children both missed, or that is through with a leaf, takes the next node from its stack (aggregate.rs:129-135) ----
#ifndef K5_POP_ROUNDS
#define K5_POP_ROUNDS 1
#endif
#ifndef K5_LEAVE_INLINE
#define K5_LEAVE_INLINE 1  // (0: a lane that popped an instance's marker parks as LINK_LEAVE and gets the outer ray back in the wave's next parked round — S3 instanced: K2 160 against 152.6 ms)
#endif
        bool left = false;  // GEN, K5_LEAVE_INLINE: this lane popped the marker of an instance
        for (int round = 0; round < (ANY ? 1 : K5_POP_ROUNDS); ++round) {  // (closest-hit: a culled entry costs no fetch; further rounds let its lane try the next one at once)
            if (__ballot(cur == CUR_POP) == 0ull) break;
            CENSUS(8, 1); CENSUS(9, __popcll(__ballot(cur == CUR_POP))); CENSUS(24, __popcll(__ballot(cur == CUR_POP)) <= 8 ? 1 : 0);
            if (cur == CUR_POP) {
                if (ANY) { c_nodes += ph_top; ph_top = 0u; }  // the phantoms above the newest entry: popped, tested, dropped, one after the other
                if (top == st_base) cur = CUR_DONE;
                else {
                    top -= WAVE;
                    u32x2 e;
                    // two different load flavours, so that the compiler cannot merge them into one flat_load of a selected pointer
                    if (top < st_base + LDS_N * WAVE) e = *top;
                    else e = __builtin_nontemporal_load(st_spill_wave + (size_t)(top - (st_base + LDS_N * WAVE)) + lane);
                    if (GEN && (e.x & ~LINK_INDEX_MASK) == (uint32_t)CUR_MARKER) {
                        // the instanced aggregate is exhausted: back to the ray of the enclosing tree (primitive.rs:158-171 returns); t_max is the hit found inside (in
                        // the instance's parameterisation, as the reference keeps it) — its record now names the instance — or what it was. The lane pops again in the
                        // next iteration.
                        if (ANY) ph_top = e.y;
                        else if (sgn & SGN_HIT_INSIDE) reinterpret_cast<int32_t*>(hits + path)[6] = (int32_t)(e.x & LINK_INDEX_MASK) + 1;
                        else t_max = __uint_as_float(e.y);
                        sgn &= SGN_RAY | SGN_HIT;
#if K5_LEAVE_INLINE
                        left = true;  // the outer ray's state comes back behind the pop rounds (three loads from save area 0), the lane pops on in the next iteration
#else
                        cur = LINK_LEAVE;  // parks: the outer ray's state comes back in the wave's next round of parked work (one ray leaves per three iterations: on its own, a round per leave)
#endif
                    }
                    else if (ANY) { c_nodes += 1u; ph_top = e.y; cur = e.x; }
                    else if (__uint_as_float(e.y) < t_max) cur = e.x;  // the rest of the far child's test, against the t_max of now
                    // (else: culled without a fetch; the lane pops again in the next iteration)
                }
            }
        }
        if (GEN && K5_LEAVE_INLINE && __builtin_expect(__ballot(left) != 0ull, 0)) {
            if (left) restore_ray_state(0);
            m_negx = __ballot((sgn & 1u) != 0u);
            m_negy = __ballot((sgn & 2u) != 0u);
            m_negz = __ballot((sgn & 4u) != 0u);
            m_irregular = __ballot((sgn & 8u) != 0u);
        }
        // ---- retire finished rays ----
        CENSUS(26, __ballot(cur == CUR_DONE) != 0ull ? 1 : 0); CENSUS(25, (__ballot(cur == CUR_DONE) != 0ull && __popcll(__ballot(cur == CUR_DONE)) <= 8) ? 1 : 0);
        if (cur == CUR_DONE) {
            const bool found = (sgn & SGN_HIT) != 0u;
            if (ANY) {
                if (occluded_out) occluded_out[path] = found ? 1 : 0;
                if (L && !found) {
                    if (K5_L_AT_REFILL) {
                        L[path] = l_new;
                    } else {
                        float4 l = L[path], c = contrib[path];
                        l.x += c.x; l.y += c.y; l.z += c.z; l.w += c.w;
                        L[path] = l;
                    }
                }
            } else if (!found) {
                if (hit16) {
                    reinterpret_cast<float4*>(hits)[path] = make_float4(__int_as_float(-1), 0.0f, 0.0f, 0.0f);
                } else {
                    float4* hp = reinterpret_cast<float4*>(hits + path);  // a miss is all zeros behind prim = -1
                    hp[0] = make_float4(__int_as_float(-1), 0.0f, 0.0f, 0.0f);
                    hp[1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                }
            }
            cur = CUR_IDLE;
        }
    }
#ifdef K5_CENSUS
    CENSUS(14, __builtin_readcyclecounter() - t_loop0);
    if (ANY == (K5_CENSUS == 2) && lane == 0) for (int i = 0; i < 32; ++i) if (c_census[i]) atomicAdd(&g_census[i], c_census[i]);
#endif
    if (ANY) {
        unsigned long long wn = c_nodes;
        for (int off = 32; off > 0; off >>= 1) wn += __shfl_down(wn, off);
        w_nodes = wn;
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

#ifndef K5_CLOSEST_WAVES
#define K5_CLOSEST_WAVES 8
#endif
#ifndef K5_ANY_WAVES
#define K5_ANY_WAVES 7  // (r04 sweep: 8 waves x 9 levels 65.5 ms of any-hit launches per headline frame, 7 waves x 11 levels 62.8)
#endif
// {LDS levels of 8-byte entries, workgroups per CU}: 256 lanes x 8 B = 2 KiB per level and workgroup; 9 levels x 8 workgroups = 144 of the CU's 160 KiB
// (S3: 98.5 % of the pushes land below level 9, 99.95 % below 12 — tools/sim/run_pair_sim.py; the rest goes to the HBM spill)
#ifndef K5_LDS_AT_8
#define K5_LDS_AT_8 9
#endif
template <int WAVES> struct K5Shape { static constexpr int LDS = (WAVES >= 8 ? K5_LDS_AT_8 : (WAVES == 7 ? 11 : (WAVES == 6 ? 13 : 15))), PER_CU = WAVES; };
#define K5_PARAMS K3_PARAMS, const uint32_t* __restrict__ big_leaf_n, float4* gen_save, int other_min, float4* hit2
#define K5_ARGS K3_ARGS, big_leaf_n, gen_save, other_min, hit2
template <bool ANY, bool GEN = false, bool HEAVY = false>
__global__ void __launch_bounds__(TRACE_BLOCK) k_trace5(K5_PARAMS);
template <>
__global__ void __launch_bounds__(TRACE_BLOCK) __attribute__((amdgpu_waves_per_eu(K5_CLOSEST_WAVES, K5_CLOSEST_WAVES))) k_trace5<false, false>(K5_PARAMS) {
    trace5_body<false, false, K5Shape<K5_CLOSEST_WAVES>::LDS>(K5_ARGS);
}
template <>
__global__ void __launch_bounds__(TRACE_BLOCK) __attribute__((amdgpu_waves_per_eu(K5_ANY_WAVES, K5_ANY_WAVES))) k_trace5<true, false>(K5_PARAMS) {
    trace5_body<true, false, K5Shape<K5_ANY_WAVES>::LDS>(K5_ARGS);
}
// scenes with spheres / bilinear patches / instances (GEN): the direction, the instance slot and the second sub-phase's live values take the closest-hit kernel past the
// 64 registers of eight waves; seven waves (<= 72 VGPRs) cost the triangle kernel under 1 % (r04 sweep: 98.0 -> 98.7 ms)
#ifndef K5_GEN_CLOSEST_WAVES
#define K5_GEN_CLOSEST_WAVES 7
#endif
#ifndef K5_GEN_ANY_WAVES
#define K5_GEN_ANY_WAVES 7
#endif
template <>
__global__ void __launch_bounds__(TRACE_BLOCK) __attribute__((amdgpu_waves_per_eu(K5_GEN_CLOSEST_WAVES, K5_GEN_CLOSEST_WAVES))) k_trace5<false, true>(K5_PARAMS) {
    trace5_body<false, true, K5Shape<K5_GEN_CLOSEST_WAVES>::LDS>(K5_ARGS);
}
template <>
__global__ void __launch_bounds__(TRACE_BLOCK) __attribute__((amdgpu_waves_per_eu(K5_GEN_ANY_WAVES, K5_GEN_ANY_WAVES))) k_trace5<true, true>(K5_PARAMS) {
    trace5_body<true, true, K5Shape<K5_GEN_ANY_WAVES>::LDS>(K5_ARGS);
}
// HEAVY (round 6): scenes where the non-triangle tests are the RULE — a quad PLY file becomes one BilinearPatch per face in the reference (shape/mesh.rs:233-256), so a real
// scene's object is patches, not triangles. At seven waves (72 VGPRs) the parked round's arithmetic lives in spill code: 124 / 94 spilled VGPRs, 164 / 148 B of scratch per
// lane — harmless where a round runs every 270 iterations (one sphere among 4.3 M triangles), and 0.7 TB of scratch traffic per frame where one runs every 17 (S3 as 2.15 M
// patches: K2 263 ms, K3 171 ms against 92 / 59 for the same object as triangles). The SAME body at five waves per SIMD holds it in registers (96 VGPRs + 10 spilled / 93 + 0):
// K2 158, K3 87 ms, 2 261 -> 3 496 Mray/s; with the refill at 24 idle lanes and the patch's fourth corner in its own record (flatten.h) K2 147, K3 77 ms, 3 736 Mray/s
// (profiles/r06_patch_heavy_scenes.txt), with the save area around the test in LDS K2 138, K3 72 ms, 3 909 Mray/s; with few non-triangles the seven-wave build stays ahead by 3-10 % (occupancy for the node step): chosen per scene, render.hip.
#ifndef K5_GEN_HEAVY_WAVES
#define K5_GEN_HEAVY_WAVES 5
#endif
#ifndef K5_GEN_HEAVY_LDS
#define K5_GEN_HEAVY_LDS (K5Shape<K5_GEN_HEAVY_WAVES>::LDS - 6)  // stack levels in LDS: the five-wave share (15 x 2 KB) less the 12 KB save area of trace5_body<SAVE_LDS>
#endif
template <>
__global__ void __launch_bounds__(TRACE_BLOCK) __attribute__((amdgpu_waves_per_eu(K5_GEN_HEAVY_WAVES, K5_GEN_HEAVY_WAVES))) k_trace5<false, true, true>(K5_PARAMS) {
    trace5_body<false, true, K5_GEN_HEAVY_LDS, true>(K5_ARGS);
}
template <>
__global__ void __launch_bounds__(TRACE_BLOCK) __attribute__((amdgpu_waves_per_eu(K5_GEN_HEAVY_WAVES, K5_GEN_HEAVY_WAVES))) k_trace5<true, true, true>(K5_PARAMS) {
    trace5_body<true, true, K5_GEN_HEAVY_LDS, true>(K5_ARGS);
}
// the any-hit kernels of scenes with instances under ShmRenderParams::disable_reference_quirks (trace5_body, STRICT): seven waves, and the five-wave build for patch-heavy scenes
template <bool HEAVY>
__global__ void __launch_bounds__(TRACE_BLOCK) k_trace5_any_strict(K5_PARAMS);
template <>
__global__ void __launch_bounds__(TRACE_BLOCK) __attribute__((amdgpu_waves_per_eu(K5_GEN_ANY_WAVES, K5_GEN_ANY_WAVES))) k_trace5_any_strict<false>(K5_PARAMS) {
    trace5_body<true, true, K5Shape<K5_GEN_ANY_WAVES>::LDS, false, true>(K5_ARGS);
}
template <>
__global__ void __launch_bounds__(TRACE_BLOCK) __attribute__((amdgpu_waves_per_eu(K5_GEN_HEAVY_WAVES, K5_GEN_HEAVY_WAVES))) k_trace5_any_strict<true>(K5_PARAMS) {
    trace5_body<true, true, K5_GEN_HEAVY_LDS, true, true>(K5_ARGS);
}

// (Round 5 built and measured k_trace6 here — TWO rays per lane, each phase run for whichever slot of a lane is ready: bit-exact, 23 % fewer wave iterations, 38.3 instead of 31.4
// lanes per VALU instruction, and 62 % more instructions per iteration for the selects that bring the chosen slot's state to each phase: K2 91 -> 133 ms, K3 58 -> 79 ms.
// Rejected with profiles/r05_rejected_two_rays_per_lane.txt; the code left the tree in round 6 (git history: round-5 commits of k_trace.hip).)

__global__ void k_reset_heads3(uint32_t* heads) { for (uint32_t i = threadIdx.x; i < 8 * 32; i += blockDim.x) heads[i] = 0; }
}  // namespace

void wf_trace_census() {
#ifdef K5_CENSUS
    unsigned long long c[32];
    if (hipMemcpyFromSymbol(c, HIP_SYMBOL(g_census), sizeof(c)) != hipSuccess || !c[0]) return;
    const double it = (double)c[0];
    fprintf(stderr, "[k5 census, %s] wave iterations %.3e | lanes per iteration: at a node %.1f, on a pending leaf %.1f, idle %.1f, pop pending %.1f | iterations with a node step %.3f\n"
                    "  leaf phases %.3e (one per %.2f iterations), primitive rounds %.3e at %.1f lanes | pop rounds %.3e at %.1f lanes | refills %.3e at %.1f rays\n",
            K5_CENSUS == 2 ? "any-hit" : "closest-hit", it, c[1] / it, c[2] / it, c[3] / it, c[4] / it, c[10] / it, (double)c[5], it / (double)c[5], (double)c[6], (double)c[7] / (double)c[6], (double)c[8], (double)c[9] / (double)c[8],
            (double)c[11], (double)c[12] / (double)c[11]);
    if (c[15]) fprintf(stderr, "  GEN: parked on a non-triangle test %.1f lanes per iteration | other phases %.3e (one per %.2f iterations) at %.1f lanes, %.1f %% of them because nothing else could run | instance entries %.3e | leave rounds %.3e at %.1f lanes\n",
                       c[21] / it, (double)c[15], it / (double)c[15], (double)c[16] / (double)c[15], 100.0 * (double)c[17] / (double)c[15], (double)c[18], (double)c[19], c[19] ? (double)c[20] / (double)c[19] : 0.0);
    fprintf(stderr, "  with 8 or fewer lanes on: %.1f %% of the node steps, %.1f %% of the leaf phases, %.1f %% of the pop rounds, %.1f %% of the %.3e retire rounds, %.1f %% of the other phases\n",
            100.0 * (double)c[22] / (double)(c[10] ? c[10] : 1), 100.0 * (double)c[23] / (double)(c[5] ? c[5] : 1), 100.0 * (double)c[24] / (double)(c[8] ? c[8] : 1), 100.0 * (double)c[25] / (double)(c[26] ? c[26] : 1), (double)c[26],
            100.0 * (double)c[27] / (double)(c[15] ? c[15] : 1));
    fprintf(stderr, "  s_memtime ticks: in refills %.3e of %.3e wave-loop ticks = %.1f %% (%.0f ticks per refill)\n", (double)c[13], (double)c[14], 100.0 * (double)c[13] / (double)c[14], (double)c[13] / (double)c[11]);
#endif
}

int wf_trace_prepare(ShmScene* s) {
    if (s->flat.nodes.size() + s->flat.instances.size() + 2 > (size_t)1 << 27) { shm_err() = "more than 2^27 BVH nodes (the traversal kernels address the node array with 32-bit byte offsets)"; return SHM_ERR_UNSUPPORTED; }
    const bool tri_only = !s->flat.has_spheres;
    for (int any = 0; any < 2; ++any) {
        int lds = any ? K5Shape<K5_ANY_WAVES>::LDS : K5Shape<K5_CLOSEST_WAVES>::LDS, per_cu = any ? K5Shape<K5_ANY_WAVES>::PER_CU : K5Shape<K5_CLOSEST_WAVES>::PER_CU;
        if (!tri_only) { lds = any ? K5Shape<K5_GEN_ANY_WAVES>::LDS : K5Shape<K5_GEN_CLOSEST_WAVES>::LDS; per_cu = any ? K5Shape<K5_GEN_ANY_WAVES>::PER_CU : K5Shape<K5_GEN_CLOSEST_WAVES>::PER_CU; }
        if (!tri_only && s->gen_heavy) { lds = K5_GEN_HEAVY_LDS; per_cu = K5Shape<K5_GEN_HEAVY_WAVES>::PER_CU; }
        s->trace3_blocks[any] = s->n_cu * per_cu;
        s->spill3_levels[any] = std::max(0, (int)s->flat.max_leaf_depth + 1 - lds) + 1;
        const size_t words = (size_t)s->trace3_blocks[any] * (TRACE_BLOCK / WAVE) * (size_t)s->spill3_levels[any] * WAVE * 2u;  // (8-byte stack entries)
        void* d = nullptr;
        if (hipMalloc(&d, words * sizeof(uint32_t)) != hipSuccess) { shm_err() = "hipMalloc of the traversal stack spill failed"; return SHM_ERR_OUT_OF_MEMORY; }
        s->allocs.push_back(d);
        (any ? s->d_spill3_any : s->d_spill3) = static_cast<uint32_t*>(d);
        if (!tri_only) {  // k_trace5<., GEN>: two 48-byte ray-state save areas per resident lane
            if (s->flat.instances.size() >= ((size_t)1 << 25)) { shm_err() = "more than 2^25 instances (the traversal kernels keep the instance index in 26 bits)"; return SHM_ERR_UNSUPPORTED; }
            void* g = nullptr;
            if (hipMalloc(&g, (size_t)s->trace3_blocks[any] * (TRACE_BLOCK / WAVE) * 6 * WAVE * sizeof(float4)) != hipSuccess) { shm_err() = "hipMalloc of the traversal save areas failed"; return SHM_ERR_OUT_OF_MEMORY; }
            s->allocs.push_back(g);
            s->d_gen_save[any] = static_cast<float4*>(g);
        }
    }
    return SHM_OK;
}

int wf_launch_trace(ShmScene* s, bool any, hipStream_t stream, const uint32_t* queue, const uint32_t* n_ptr, uint32_t n_direct, const ShmRay* rays,
                    ShmHit* hits, uint8_t* occluded, float4* L, const float4* contrib, int hit16) {
    uint32_t* heads = s->d_heads3 + (any ? 8 * 32 : 0);
    uint32_t* spill = any ? s->d_spill3_any : s->d_spill3;
    hipLaunchKernelGGL(k_reset_heads3, dim3(1), dim3(64), 0, stream, heads);
    const int leaf_min = any ? s->leaf_min_any : s->leaf_min;
    const bool tri_only = !s->flat.has_spheres;
    // hit16 with the GEN kernels: the split record form (wavefront.h, load_hit_tri) — the second records follow the `capacity` first ones in the hit allocation
    float4* const hit2 = (hit16 && !tri_only && hits) ? reinterpret_cast<float4*>(hits) + s->capacity : nullptr;
#define TRACE5_LAUNCH(ANY, ...)                                                                                                               \
    hipLaunchKernelGGL((k_trace5<ANY, __VA_ARGS__>), dim3(s->trace3_blocks[ANY]), dim3(TRACE_BLOCK), 0, stream, s->dsv, queue, n_ptr, n_direct, heads, rays,  \
                       hits, occluded, L, contrib, s->d_counters, spill, s->spill3_levels[ANY], (ANY ? s->refill_min_any : s->refill_min), leaf_min, s->queue_parts, s->trace_rays_per_lane, hit16, s->d_big_leaf_n, \
                       s->d_gen_save[ANY ? 1 : 0], (ANY ? s->other_min_any : s->other_min), hit2)
    if (any && s->flat.has_instances && s->dsv.quirks_off) {  // (the PBRT-v4 form of the shadow ray's way into an instance: its own instantiations)
        if (s->gen_heavy) hipLaunchKernelGGL((k_trace5_any_strict<true>), dim3(s->trace3_blocks[1]), dim3(TRACE_BLOCK), 0, stream, s->dsv, queue, n_ptr, n_direct, heads, rays, hits, occluded, L, contrib,
                                             s->d_counters, spill, s->spill3_levels[1], s->refill_min_any, leaf_min, s->queue_parts, s->trace_rays_per_lane, hit16, s->d_big_leaf_n, s->d_gen_save[1], s->other_min_any, hit2);
        else hipLaunchKernelGGL((k_trace5_any_strict<false>), dim3(s->trace3_blocks[1]), dim3(TRACE_BLOCK), 0, stream, s->dsv, queue, n_ptr, n_direct, heads, rays, hits, occluded, L, contrib,
                                s->d_counters, spill, s->spill3_levels[1], s->refill_min_any, leaf_min, s->queue_parts, s->trace_rays_per_lane, hit16, s->d_big_leaf_n, s->d_gen_save[1], s->other_min_any, hit2);
    }
    else if (tri_only) { if (any) TRACE5_LAUNCH(true, false); else TRACE5_LAUNCH(false, false); }
    else if (s->gen_heavy) { if (any) TRACE5_LAUNCH(true, true, true); else TRACE5_LAUNCH(false, true, true); }
    else { if (any) TRACE5_LAUNCH(true, true); else TRACE5_LAUNCH(false, true); }
#undef TRACE5_LAUNCH
    LAUNCH_TRY(any ? "k_trace<any>" : "k_trace<closest>");
    return SHM_OK;
}
