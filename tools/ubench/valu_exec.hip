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

enum Kind { FMA = 0, CNDMASK_SGPR, CMP_SGPR, MAX3, ADD_E32, MUL_E32, MOV_E32, CND_VCC, N_KINDS };
static const char* kNames[N_KINDS] = {"v_fma_f32", "v_cndmask_b32 (SGPR mask)", "v_cmp_gt_f32 -> SGPR pair", "v_max3_f32", "v_add_f32_e32 (VOP2)", "v_mul_f32_e32 (VOP2)", "v_mov_b32_e32 (VOP1)", "v_cndmask_b32_e32 (VOP2, vcc)"};

template <int KIND>
__global__ void __launch_bounds__(1024) k_rate(float* out, unsigned long long* cycles, int iters, float seed, unsigned long long lanes_lo, unsigned long long lanes_hi) {
    const unsigned long long lanes = threadIdx.x < blockDim.x / 2 ? lanes_lo : lanes_hi;  // (waves 0-7 | waves 8-15 of the workgroup: two of each on every SIMD)
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
#define OP_ADD2(k) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(r##k) : "v"(c));
#define OP_MUL2(k) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(r##k) : "v"(b));
#define OP_MOV1(k) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(r##k) : "v"(b));
#define OP_CNDV(k) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(r##k) : "v"(b) : "vcc");
            if (KIND == FMA) { REP64(OP_FMA) }
            else if (KIND == ADD_E32) { REP64(OP_ADD2) }
            else if (KIND == MUL_E32) { REP64(OP_MUL2) }
            else if (KIND == MOV_E32) { REP64(OP_MOV1) }
            else if (KIND == CND_VCC) { REP64(OP_CNDV) }
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
static void run(unsigned long long lanes, const char* what, int iters, float* d_out, unsigned long long* d_cyc, unsigned long long lanes_hi = 0, bool mixed = false) {
    if (!mixed) lanes_hi = lanes;
    const int W = 4, block = 64 * 4 * W, grid = 256;
    hipLaunchKernelGGL(k_rate<KIND>, dim3(grid), dim3(block), 0, 0, d_out, d_cyc, 16, 1.0f, lanes, lanes_hi);  // warm-up
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate<KIND>, dim3(grid), dim3(block), 0, 0, d_out, d_cyc, iters, 1.0f, lanes, lanes_hi);
    (void)hipEventRecord(e1);
    hipError_t e = hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (e != hipSuccess) { printf("%s: %s\n", kNames[KIND], hipGetErrorString(e)); exit(1); }
    std::vector<unsigned long long> c((size_t)grid * block / 64);
    hipMemcpy(c.data(), d_cyc, c.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long mx = 0, mx_lo = 0, mx_hi = 0;
    const int wpb = block / 64;
    for (size_t i = 0; i < c.size(); ++i) { if (c[i] > mx) mx = c[i]; if ((int)(i % wpb) < wpb / 2) { if (c[i] > mx_lo) mx_lo = c[i]; } else if (c[i] > mx_hi) mx_hi = c[i]; }
    if (mixed) printf("%-28s EXEC = %-34s clocks until the last full wave ends %8.0f, until the last sparse wave ends %8.0f (instructions per wave %d)\n", kNames[KIND], what, (double)mx_lo, (double)mx_hi, iters * 64);
    else printf("%-28s EXEC = %-34s clocks/inst/SIMD = %6.3f   wall %.3f ms (HIP events) = %.2f ns per instruction per SIMD\n", kNames[KIND], what, (double)mx / ((double)iters * 64.0 * W), ms, ms * 1e6 / ((double)iters * 64.0 * W));
}

int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    float* d_out; unsigned long long* d_cyc;
    if (hipMalloc(&d_out, 256 * 1024 * sizeof(float)) != hipSuccess || hipMalloc(&d_cyc, 256 * 16 * 8) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    const int iters = 512;
    struct { unsigned long long m; const char* what; } masks[] = {
        {~0ull, "all 64"}, {0x00000000ffffffffull, "lanes 0-31"}, {0xffffffff00000000ull, "lanes 32-63"}, {0x000000000000ffffull, "lanes 0-15"},
        {0x0000ffff0000ffffull, "lanes 0-15 + 32-47"}, {0x5555555555555555ull, "every second lane"}, {0x0001000100010001ull, "one lane in 16"}, {1ull, "lane 0"},
        {0x3ull, "lanes 0-1"}, {0xfull, "lanes 0-3"}, {0xffull, "lanes 0-7"}, {0x1ffull, "lanes 0-8"}, {0x3ffull, "lanes 0-9"}, {0x7ffull, "lanes 0-10"}, {0xfffull, "lanes 0-11"},
        {0x0000000100000000ull | 0xffull, "lanes 0-7 + 32"}, {0x8000000000000000ull | 0xffull, "lanes 0-7 + 63"}, {0x8000400020001000ull | 0x0008000400020001ull, "8 lanes, two per 16"}, {0x8040201008040201ull | 0x100ull, "9 lanes spread"}, {0xffffffull, "lanes 0-23"},
        {0x0101010101010101ull, "one lane in 8"}, {0x1111111111111111ull, "one lane in 4"}, {0x00000000000f000full, "lanes 0-3 + 16-19"}};
    for (auto& m : masks) {
        run<FMA>(m.m, m.what, iters, d_out, d_cyc); run<CNDMASK_SGPR>(m.m, m.what, iters, d_out, d_cyc);
        run<CMP_SGPR>(m.m, m.what, iters, d_out, d_cyc); run<MAX3>(m.m, m.what, iters, d_out, d_cyc);
        printf("\n");
    }
    for (unsigned long long m : {~0ull, 0xfull, 0x1ffull}) {
        const char* what = m == ~0ull ? "all 64" : (m == 0xfull ? "lanes 0-3" : "lanes 0-8");
        run<ADD_E32>(m, what, iters, d_out, d_cyc); run<MUL_E32>(m, what, iters, d_out, d_cyc); run<MOV_E32>(m, what, iters, d_out, d_cyc); run<CND_VCC>(m, what, iters, d_out, d_cyc);
    }
    // the few-lane rows above: 5 x the clocks per instruction with 8 or fewer lanes on. A property of the instruction, or of a chip that runs nothing else? Two waves of
    // every SIMD with all lanes, two with four lanes:
    run<FMA>(~0ull, "2 waves all 64 | 2 waves lanes 0-3", iters, d_out, d_cyc, 0xfull, true);
    run<CNDMASK_SGPR>(~0ull, "2 waves all 64 | 2 waves lanes 0-3", iters, d_out, d_cyc, 0xfull, true);
    run<MAX3>(~0ull, "2 waves all 64 | 2 waves lanes 0-3", iters, d_out, d_cyc, 0xfull, true);
    run<MAX3>(~0ull, "4 waves all 64 (for comparison)", iters, d_out, d_cyc, ~0ull, true);
    run<MAX3>(0xfull, "4 waves lanes 0-3 (for comparison)", iters, d_out, d_cyc, 0xfull, true);
    return 0;
}
