// valu_rate.hip — how many clocks one wave64 VALU instruction occupies a SIMD's issue port on gfx950, per instruction kind, measured
// (bench.py's roofline_valu prices SQ_ACTIVE_INST_VALU quad-cycles against the kernel's clocks: VERDICT r02 item 6 asks which rate that
// assumes). Every SIMD runs W waves; each wave executes ITER x 64 instructions of one kind on 8 independent registers (no dependent chain
// shorter than 8 instructions) between two s_memtime reads. clocks per instruction per SIMD = (t1 - t0) / (W x ITER x 64), taken from the
// wave that finishes last on its SIMD (all waves of a SIMD start together: one workgroup of 4 x W waves per CU, 256 workgroups).
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

enum Kind { FMA = 0, ADD, MUL, CNDMASK_SGPR, CMP_SGPR, PK_MUL, PK_ADD, MOV, ADD_U32, LSHL_ADD, MAX, RCP, SALU_AND, N_KINDS };
static const char* kNames[N_KINDS] = {"v_fma_f32", "v_add_f32", "v_mul_f32", "v_cndmask_b32 (SGPR mask)", "v_cmp_gt_f32 -> SGPR pair", "v_pk_mul_f32 (2 lanes-ops per lane)",
                                      "v_pk_add_f32", "v_mov_b32", "v_add_u32", "v_lshl_add_u32", "v_max_f32", "v_rcp_f32 (transcendental)", "s_and_b64 (scalar, for comparison)"};

template <int KIND>
__global__ void __launch_bounds__(1024) k_rate(float* out, unsigned long long* cycles, int iters, float seed) {
    float r0 = seed + threadIdx.x, r1 = r0 + 1.f, r2 = r0 + 2.f, r3 = r0 + 3.f, r4 = r0 + 4.f, r5 = r0 + 5.f, r6 = r0 + 6.f, r7 = r0 + 7.f;
    float p0 = r0, p1 = r1, p2 = r2, p3 = r3, p4 = r4, p5 = r5, p6 = r6, p7 = r7;  // second halves of the packed pairs
    const float b = 1.0000001f, c = 1e-9f;
    unsigned long long mask = 0x5555aaaa3333ccccull ^ (unsigned long long)blockIdx.x;
    unsigned long long s0 = mask, s1 = ~mask;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#define R(k) r##k
#define OP_FMA(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r##k) : "v"(b), "v"(c));
#define OP_ADD(k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r##k) : "v"(c));
#define OP_MUL(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r##k) : "v"(b));
#define OP_CND(k) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r##k) : "v"(b), "s"(mask));
#define OP_CMP(k) asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(s0) : "v"(r##k), "v"(b));
#define OP_PKM(k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double*)&q##k) : "v"(bb));
#define OP_PKA(k) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&q##k) : "v"(bb));
#define OP_MOV(k) asm volatile("v_mov_b32 %0, %1" : "=v"(r##k) : "v"(p##k));
#define OP_ADDU(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r##k) : "v"(p##k));
#define OP_LSHL(k) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(r##k) : "v"(p##k));
#define OP_MAX(k) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r##k) : "v"(p##k));
#define OP_RCP(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(r##k));
#define OP_SAND(k) asm volatile("s_and_b64 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");  // (SCC declared: without it the loop's own compare was clobbered and the kernel never ended)
        if (KIND == FMA) { REP64(OP_FMA) }
        else if (KIND == ADD) { REP64(OP_ADD) }
        else if (KIND == MUL) { REP64(OP_MUL) }
        else if (KIND == CNDMASK_SGPR) { REP64(OP_CND) }
        else if (KIND == CMP_SGPR) { REP64(OP_CMP) }
        else if (KIND == PK_MUL || KIND == PK_ADD) {
            float2 q0 = make_float2(r0, p0), q1 = make_float2(r1, p1), q2 = make_float2(r2, p2), q3 = make_float2(r3, p3), q4 = make_float2(r4, p4), q5 = make_float2(r5, p5),
                   q6 = make_float2(r6, p6), q7 = make_float2(r7, p7);
            const double bb = __hiloint2double(__float_as_int(b), __float_as_int(b));
            if (KIND == PK_MUL) { REP64(OP_PKM) } else { REP64(OP_PKA) }
            r0 = q0.x; r1 = q1.x; r2 = q2.x; r3 = q3.x; r4 = q4.x; r5 = q5.x; r6 = q6.x; r7 = q7.x;
            p0 = q0.y; p1 = q1.y; p2 = q2.y; p3 = q3.y; p4 = q4.y; p5 = q5.y; p6 = q6.y; p7 = q7.y;
        }
        else if (KIND == MOV) { REP64(OP_MOV) }
        else if (KIND == ADD_U32) { REP64(OP_ADDU) }
        else if (KIND == LSHL_ADD) { REP64(OP_LSHL) }
        else if (KIND == MAX) { REP64(OP_MAX) }
        else if (KIND == RCP) { REP64(OP_RCP) }
        else { REP64(OP_SAND) }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7 + (float)(s0 & 1ull);
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
static void run(int waves_per_simd, int iters, float* d_out, unsigned long long* d_cyc) {
    const int block = 64 * 4 * waves_per_simd, grid = 256;  // one workgroup per CU (1024 threads = 4 waves per SIMD at most: W = 1, 2, 4)
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rate<KIND>, dim3(grid), dim3(block), 0, 0, d_out, d_cyc, 16, 1.0f);  // warm-up
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate<KIND>, dim3(grid), dim3(block), 0, 0, d_out, d_cyc, iters, 1.0f);
    hipEventRecord(e1);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("%s: %s\n", kNames[KIND], hipGetErrorString(e)); exit(1); }
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c((size_t)grid * block / 64);
    hipMemcpy(c.data(), d_cyc, c.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    double mean = 0;
    for (auto v : c) { if (v > mx) mx = v; mean += (double)v; }
    mean /= (double)c.size();
    const double insts_per_wave = (double)iters * 64.0;
    printf("%-40s W=%d  clocks/inst/SIMD = %6.3f (mean wave %.3f)   wall %.3f ms -> %.2f GHz-equivalent at that rate\n", kNames[KIND], waves_per_simd,
           (double)mx / (insts_per_wave * waves_per_simd), mean / (insts_per_wave * waves_per_simd), ms,
           ((double)mx / 1e6) / ms);
}

int main() {
    setvbuf(stdout, NULL, _IONBF, 0);  // every line as it is measured (run it under `timeout`: a first version never returned on the GPU box)
    float* d_out; unsigned long long* d_cyc;
    if (hipMalloc(&d_out, 256 * 1024 * sizeof(float)) != hipSuccess || hipMalloc(&d_cyc, 256 * 16 * 8) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    const int iters = 512;
    for (int w : {1, 2, 4}) {
        run<FMA>(w, iters, d_out, d_cyc); run<ADD>(w, iters, d_out, d_cyc); run<MUL>(w, iters, d_out, d_cyc); run<CNDMASK_SGPR>(w, iters, d_out, d_cyc);
        run<CMP_SGPR>(w, iters, d_out, d_cyc); run<PK_MUL>(w, iters, d_out, d_cyc); run<PK_ADD>(w, iters, d_out, d_cyc); run<MOV>(w, iters, d_out, d_cyc);
        run<ADD_U32>(w, iters, d_out, d_cyc); run<LSHL_ADD>(w, iters, d_out, d_cyc); run<MAX>(w, iters, d_out, d_cyc); run<RCP>(w, iters, d_out, d_cyc);
        run<SALU_AND>(w, iters, d_out, d_cyc);
        printf("\n");
    }
    return 0;
}
