/* shimmer_hip_probe.h — the C ABI of libshimmer_hip_probe.so, a TEST library built beside libshimmer_hip.so from the same shared arithmetic headers
 * (shimmer_amd/csrc/shm/, csrc/probe/k_leaf_probe.hip). No host of the product binds it; the `-m gpu` suite does (tests/device_leaves.py). */
#ifndef SHIMMER_HIP_PROBE_H
#define SHIMMER_HIP_PROBE_H
#include "shimmer_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Test entry (DEVICE): one leaf function of the shared arithmetic — intersect_triangle, intersect_p_cached, every BxDF's f / sample_f / pdf incl. the
 * LayeredBxDF walks, shape sampling, camera rays, film accumulation ... (shimmer_amd/csrc/shm/probe.h lists the `op` codes and each one's argument layout) —
 * evaluated by one wave of `device` on flat 32-bit words; *fn_result receives the wrapped function's integer result (Some / None). The `-m gpu` suite replays
 * the committed golden vectors (the reference's in-source known answers: aggregate.rs:575-702, bxdf.rs:1871-1903 ..., and the independent re-evaluations of
 * tests/golden/golden_leaves.json / golden_layered.json) through it: the device code against the vectors themselves, not through the CPU oracle. Not a render path.
 * Arguments are copied into a zero-padded device buffer and every lane writes a region of its own: an op cannot read or write out of bounds whatever n_in / n_out say;
 * the one op whose output length is an argument (the sampler stream) is checked against n_out. */
SHM_API int shm_debug_eval_leaf(int device, int op, const uint32_t* in_words, uint32_t n_in, uint32_t* out_words, uint32_t n_out, int* fn_result);
/* thread-local message of the last failing shm_debug_eval_leaf call */
SHM_API const char* shm_probe_last_error(void);
#ifdef __cplusplus
}
#endif
#endif
