// valu_exec.hip — does a wave64 VALU instruction on gfx950 cost less when half of its lanes are switched off? (round 5: the traversal kernels issue with 32 of 64
// lanes; if the SIMD skipped the passes of an all-inactive half, PLACING the active lanes would pay.) Every SIMD runs 4 waves; each wave executes ITER x 64
// instructions of one kind on 8 independent registers under an EXEC mask given by the host, between two clock reads taken with all lanes on.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/valu_exec.hip -o valu_exec && ./valu_exec
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

enum Kind { FMA = 0, CNDMASK_SGPR, CMP_SGPR, MAX3, N_KINDS };
static const char* kNames[N_KINDS] = {"v_fma_f32", "v_cndmask_b32 (SGPR mask)", "v_cmp_gt_f32 -> SGPR pair", "v_max3_f32"};

template <int KIND>
__global__ void __launch_bounds__(1024) k_rate(float* out, unsigned long long* cycles, int iters, float seed, unsigned long long lanes) {
    float r0 = seed + threadIdx.x, r1 = r0 + 1.f, r2 = r0 + 2.f, r3 = r0 + 3.f, r4 = r0 + 4.f, r5 = r0 + 5.f, r6 = r0 + 6.f, r7 = r0 + 7.f;
    const float b = 1.0000001f, c = 1e-9f;
    unsigned long long mask = 0x5555aaaa3333ccccull ^ (unsigned long long)blockIdx.x;
    unsigned long long s0 = mask;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if ((lanes >> (threadIdx.x & 63)) & 1ull) {
        for (int i = 0; i < iters; ++i) {
#define OP_FMA(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r##k) : "v"(b), "v"(c));
#define OP_CND(k) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r##k) : "v"(b), "s"(mask));
#define OP_CMP(k) asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(s0) : "v"(r##k), "v"(b));
#define OP_MAX3(k) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r##k) : "v"(b), "v"(c));
            if (KIND == FMA) { REP64(OP_FMA) }
            else if (KIND == CNDMASK_SGPR) { REP64(OP_CND) }
            else if (KIND == CMP_SGPR) { REP64(OP_CMP) }
            else { REP64(OP_MAX3) }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + (float)(s0 & 1ull);
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
static void run(unsigned long long lanes, const char* what, int iters, float* d_out, unsigned long long* d_cyc) {
    const int W = 4, block = 64 * 4 * W, grid = 256;
    hipLaunchKernelGGL(k_rate<KIND>, dim3(grid), dim3(block), 0, 0, d_out, d_cyc, 16, 1.0f, lanes);  // warm-up
    hipLaunchKernelGGL(k_rate<KIND>, dim3(grid), dim3(block), 0, 0, d_out, d_cyc, iters, 1.0f, lanes);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("%s: %s\n", kNames[KIND], hipGetErrorString(e)); exit(1); }
    std::vector<unsigned long long> c((size_t)grid * block / 64);
    hipMemcpy(c.data(), d_cyc, c.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    for (auto v : c) if (v > mx) mx = v;
    printf("%-28s EXEC = %-34s clocks/inst/SIMD = %6.3f\n", kNames[KIND], what, (double)mx / ((double)iters * 64.0 * W));
}

int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    float* d_out; unsigned long long* d_cyc;
    if (hipMalloc(&d_out, 256 * 1024 * sizeof(float)) != hipSuccess || hipMalloc(&d_cyc, 256 * 16 * 8) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    const int iters = 512;
    struct { unsigned long long m; const char* what; } masks[] = {
        {~0ull, "all 64"}, {0x00000000ffffffffull, "lanes 0-31"}, {0xffffffff00000000ull, "lanes 32-63"}, {0x000000000000ffffull, "lanes 0-15"},
        {0x0000ffff0000ffffull, "lanes 0-15 + 32-47"}, {0x5555555555555555ull, "every second lane"}, {0x0001000100010001ull, "one lane in 16"}, {1ull, "lane 0"}};
    for (auto& m : masks) {
        run<FMA>(m.m, m.what, iters, d_out, d_cyc); run<CNDMASK_SGPR>(m.m, m.what, iters, d_out, d_cyc);
        run<CMP_SGPR>(m.m, m.what, iters, d_out, d_cyc); run<MAX3>(m.m, m.what, iters, d_out, d_cyc);
        printf("\n");
    }
    return 0;
}
