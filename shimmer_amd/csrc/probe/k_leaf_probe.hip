// probe/k_leaf_probe.hip — TEST LIBRARY (libshimmer_hip_probe.so, built beside the product library by the same Makefile; NOT linked into libshimmer_hip.so since round 6):
// one leaf function of the shared arithmetic (shm/probe.h — the headers the render kernels are made of) evaluated ON THE DEVICE for flat arguments, so that the `-m gpu`
// suite replays the committed golden vectors through the device code itself and compares with the committed expected values directly (tests/test_gpu_leaf_replay.py).
// Declared in include/shimmer_hip_probe.h. Not on any render path; no host of the product binds it.
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../../include/shimmer_hip_probe.h"
#include "../shm/probe.h"

namespace {
thread_local std::string g_probe_err;
constexpr uint32_t PROBE_IN_WORDS = 256;   // the device copy of the arguments is zero-padded to this many words: an op never reads past it (the longest layout — a layered BxDF + two directions + samples — is under 64)
constexpr uint32_t PROBE_OUT_WORDS = 256;  // ... and every lane writes its results into a region of its own of max(n_out, this) words

// every lane of one wave evaluates the same function on the same arguments (what a render's wave does for coherent lanes) into its OWN output region; the kernel reports
// whether all 64 lanes agreed bit for bit, lane 0's region is the result
__global__ void __launch_bounds__(64) k_leaf_probe(int op, const uint32_t* in, uint32_t* scratch, uint32_t region_words, uint32_t n_out, int* result) {
    uint32_t* mine = scratch + (size_t)threadIdx.x * region_words;
    const int r = shm::leaf_probe(op, in, mine);
    __threadfence();  // (every lane's stores are visible before any lane reads lane 0's region)
    bool same = __shfl(r, 0) == r;
    for (uint32_t i = 0; i < n_out; ++i) same = same && (mine[i] == scratch[i]);  // (lane 0's region starts the scratch; every lane has written its own by now: one wave, in lockstep per store)
    const unsigned long long agree = __ballot(same);
    if (threadIdx.x == 0) { result[0] = r; result[1] = agree == ~0ull ? 1 : 0; }
}
}  // namespace

extern "C" __attribute__((visibility("default"))) const char* shm_probe_last_error(void) { return g_probe_err.c_str(); }

extern "C" __attribute__((visibility("default"))) int shm_debug_eval_leaf(int device, int op, const uint32_t* in_words, uint32_t n_in, uint32_t* out_words, uint32_t n_out, int* fn_result) {
    if (!in_words || !out_words || n_in == 0 || n_out == 0 || op <= 0 || op >= shm::PROBE_N_OPS || n_in > (1u << 20) || n_out > (1u << 20)) { g_probe_err = "shm_debug_eval_leaf: invalid arguments"; return SHM_ERR_INVALID_ARGUMENT; }
    // the one op whose output length is an ARGUMENT (the sampler stream: in[5] draws): it must fit what the caller provided
    if (op == shm::PROBE_SAMPLER_STREAM && (n_in < 6 || in_words[5] > n_out)) { g_probe_err = "shm_debug_eval_leaf: the sampler stream's length exceeds n_out"; return SHM_ERR_INVALID_ARGUMENT; }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) { g_probe_err = "no HIP device visible (the probe has no CPU fallback)"; return SHM_ERR_NO_DEVICE; }
    if (hipSetDevice(device) != hipSuccess) { g_probe_err = "hipSetDevice"; return SHM_ERR_DEVICE; }
    const uint32_t in_padded = n_in > PROBE_IN_WORDS ? n_in : PROBE_IN_WORDS, region = n_out > PROBE_OUT_WORDS ? n_out : PROBE_OUT_WORDS;
    uint32_t *d_in = nullptr, *d_scratch = nullptr;
    int* d_res = nullptr;
    if (hipMalloc((void**)&d_in, (size_t)in_padded * 4) != hipSuccess || hipMalloc((void**)&d_scratch, (size_t)region * 64 * 4) != hipSuccess || hipMalloc((void**)&d_res, 8) != hipSuccess) {
        hipFree(d_in); hipFree(d_scratch); hipFree(d_res);
        g_probe_err = "hipMalloc";
        return SHM_ERR_OUT_OF_MEMORY;
    }
    int rc = SHM_OK, res[2] = {0, 0};
    if (hipMemset(d_in, 0, (size_t)in_padded * 4) != hipSuccess || hipMemcpy(d_in, in_words, (size_t)n_in * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(d_scratch, 0, (size_t)region * 64 * 4) != hipSuccess) rc = SHM_ERR_DEVICE;
    if (rc == SHM_OK) {
        hipLaunchKernelGGL(k_leaf_probe, dim3(1), dim3(64), 0, 0, op, d_in, d_scratch, region, n_out, d_res);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = SHM_ERR_DEVICE;
    }
    if (rc == SHM_OK && (hipMemcpy(out_words, d_scratch, (size_t)n_out * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(res, d_res, 8, hipMemcpyDeviceToHost) != hipSuccess)) rc = SHM_ERR_DEVICE;
    hipFree(d_in); hipFree(d_scratch); hipFree(d_res);
    if (rc != SHM_OK) { g_probe_err = "shm_debug_eval_leaf: device error"; return rc; }
    if (!res[1]) { g_probe_err = "shm_debug_eval_leaf: the lanes of the wave disagree"; return SHM_ERR_INTERNAL; }
    if (fn_result) *fn_result = res[0];
    return SHM_OK;
}
