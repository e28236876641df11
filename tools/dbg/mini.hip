#include <hip/hip_runtime.h>
#include <stdio.h>
namespace { __global__ void k_anon(unsigned* p) { *p = 42; } }
__global__ void k_named(unsigned* p) { *p = 43; }
extern "C" int mini_run(int which) {
    unsigned* d = nullptr; unsigned h = 0;
    hipStream_t st; hipStreamCreate(&st);
    if (hipMalloc(&d, 4) != hipSuccess) return -1;
    fprintf(stderr, "[mini] launching %d\n", which); fflush(stderr);
    if (which == 0) hipLaunchKernelGGL(k_anon, dim3(1), dim3(1), 0, st, d);
    else hipLaunchKernelGGL(k_named, dim3(1), dim3(1), 0, st, d);
    fprintf(stderr, "[mini] launched: %s\n", hipGetErrorString(hipGetLastError())); fflush(stderr);
    hipStreamSynchronize(st);
    hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    return (int)h;
}
