// k_leaf_probe.hip — TEST-ONLY entry point of the library: one leaf function of the shared arithmetic (shm/probe.h) evaluated ON THE DEVICE for flat arguments, so that the
// `-m gpu` suite replays the committed golden vectors through the device code itself and compares with the committed expected values directly
// (tests/test_gpu_leaf_replay.py). Not on any render path.
#include "wavefront.h"
#include "shm/probe.h"

namespace {
// every lane of one wave evaluates the same function on the same arguments (what a render's wave does for coherent lanes); lane `check_lane` writes, and the kernel also
// reports whether all 64 lanes agreed bit for bit
__global__ void __launch_bounds__(64) k_leaf_probe(int op, const uint32_t* in, uint32_t* out, uint32_t n_out, int* result) {
    uint32_t local[64];
    for (uint32_t i = 0; i < 64; ++i) local[i] = 0u;
    const int r = shm::leaf_probe(op, in, n_out <= 64u ? local : out /* large outputs (the sampler stream): written in place by every lane, same values */);
    bool same = true;
    if (n_out <= 64u) {
        for (uint32_t i = 0; i < n_out; ++i) same = same && (__shfl(local[i], 0) == local[i]);
        if (threadIdx.x == 0) for (uint32_t i = 0; i < n_out; ++i) out[i] = local[i];
    }
    same = same && (__shfl(r, 0) == r);
    const unsigned long long agree = __ballot(same);
    if (threadIdx.x == 0) { result[0] = r; result[1] = agree == ~0ull ? 1 : 0; }
}
}  // namespace

extern "C" __attribute__((visibility("default"))) int shm_debug_eval_leaf(int device, int op, const uint32_t* in_words, uint32_t n_in, uint32_t* out_words, uint32_t n_out, int* fn_result) {
    if (!in_words || !out_words || n_in == 0 || n_out == 0 || op <= 0 || op >= shm::PROBE_N_OPS) { shm_err() = "shm_debug_eval_leaf: invalid arguments"; return SHM_ERR_INVALID_ARGUMENT; }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) { shm_err() = "no HIP device visible (libshimmer_hip has no CPU fallback)"; return SHM_ERR_NO_DEVICE; }
    HIP_TRY(hipSetDevice(device));
    uint32_t *d_in = nullptr, *d_out = nullptr;
    int* d_res = nullptr;
    HIP_TRY(hipMalloc((void**)&d_in, (size_t)n_in * 4));
    if (hipMalloc((void**)&d_out, (size_t)n_out * 4) != hipSuccess || hipMalloc((void**)&d_res, 8) != hipSuccess) { hipFree(d_in); hipFree(d_out); shm_err() = "hipMalloc"; return SHM_ERR_OUT_OF_MEMORY; }
    int rc = SHM_OK, res[2] = {0, 0};
    if (hipMemcpy(d_in, in_words, (size_t)n_in * 4, hipMemcpyHostToDevice) != hipSuccess || hipMemset(d_out, 0, (size_t)n_out * 4) != hipSuccess) rc = SHM_ERR_DEVICE;
    if (rc == SHM_OK) {
        hipLaunchKernelGGL(k_leaf_probe, dim3(1), dim3(64), 0, 0, op, d_in, d_out, n_out, d_res);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = SHM_ERR_DEVICE;
    }
    if (rc == SHM_OK && (hipMemcpy(out_words, d_out, (size_t)n_out * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(res, d_res, 8, hipMemcpyDeviceToHost) != hipSuccess)) rc = SHM_ERR_DEVICE;
    hipFree(d_in); hipFree(d_out); hipFree(d_res);
    if (rc != SHM_OK) { shm_err() = "shm_debug_eval_leaf: device error"; return rc; }
    if (!res[1]) { shm_err() = "shm_debug_eval_leaf: the lanes of the wave disagree"; return SHM_ERR_INTERNAL; }
    if (fn_result) *fn_result = res[0];
    return SHM_OK;
}
