// Test infrastructure: lets host/host_mirror.cpp (BVH build, tiling, cameras, get_image, PFM, PLY) link on its own for the
// sanitizer build of tests/test_sanitizers.py. In the product these two symbols live in shimmer_hip.hip.
#include <string>
static thread_local std::string g_err;
extern "C" __attribute__((visibility("default"))) void shm_set_last_error(const char* msg) { g_err = msg ? msg : ""; }
extern "C" __attribute__((visibility("default"))) const char* shm_last_error(void) { return g_err.c_str(); }
