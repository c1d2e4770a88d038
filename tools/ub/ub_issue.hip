// micro-benchmark: vector-instruction ISSUE cost on gfx950 by instruction class, at W = 1..4 waves per SIMD.
// What does one wave64 instruction of each class hold a SIMD's vector unit for when W waves compete for it?  (The list decoder's
// instruction mix is ~53 % float64 arithmetic, ~17 % integer, ~30 % moves / selects / compares: profiles/r03_scl_pmc.json; the SQ
// counters count quad-cycles and cannot tell a 2-cycle instruction from a 4-cycle one.)
// Every wave runs ITERS x 32 independent instructions of the class (8 registers, no dependence closer than 8 instructions apart);
// cycles per instruction and SIMD = elapsed shader cycles (s_memtime) x 1 / (ITERS x 32 x W).
//   hipcc -O3 --offload-arch=gfx950 -o ub_issue tools/ub/ub_issue.hip && ./ub_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)

enum { FMA64, ADD64, MUL64, RCP64, ADD32, CND32, MOV32, AND32, LSHL64, FMA32, MIX_FMA64_ADD32, MIX_FMA64_CND32, MIX_ADD64_MOV32, CMP64_CND, NCLASS };
static const char* kName[NCLASS] = {"v_fma_f64", "v_add_f64", "v_mul_f64", "v_rcp_f64", "v_add_u32", "v_cndmask_b32", "v_mov_b32", "v_and_b32", "v_lshlrev_b64", "v_fma_f32",
                                    "mix 1:1 v_fma_f64 / v_add_u32", "mix 1:1 v_fma_f64 / v_cndmask_b32", "mix 1:1 v_add_f64 / v_mov_b32", "v_cmp_gt_f64 + v_cndmask_b32 pairs"};

template <int CLS, int WPS>
__global__ __launch_bounds__(64, WPS) void ub(double* out, unsigned long long* cyc, int iters, double seed)
{
    double d[8]; uint32_t u[8]; float f[8];
    #pragma unroll
    for (int k = 0; k < 8; ++k) { d[k] = seed + k * 0.001 + threadIdx.x * 1e-6; u[k] = (uint32_t)(threadIdx.x * 8 + k + 1); f[k] = (float)d[k]; }
    const double c1 = 0.999999, c2 = 1e-9;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#define I_FMA64(k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[k]) : "v"(c1), "v"(c2));
#define I_ADD64(k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[k]) : "v"(c2));
#define I_MUL64(k) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[k]) : "v"(c1));
#define I_RCP64(k) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[k]));
#define I_ADD32(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));
#define I_CND32(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[k]) : "v"(u[(k + 1) & 7]) : );
#define I_MOV32(k) asm volatile("v_mov_b32 %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));
#define I_AND32(k) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));
#define I_LSHL64(k) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(d[k]));
#define I_FMA32(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[k]) : "v"(f[(k + 1) & 7]), "v"(f[(k + 2) & 7]));
#define I_MIXA(k) I_FMA64(k) I_ADD32(k)
#define I_MIXB(k) I_FMA64(k) I_CND32(k)
#define I_MIXC(k) I_ADD64(k) I_MOV32(k)
#define I_CMPC(k) asm volatile("v_cmp_gt_f64 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(u[k]) : "v"(d[k]), "v"(c1), "v"(u[(k + 1) & 7]) : "vcc");
        if constexpr (CLS == FMA64) { REP32(I_FMA64) }
        if constexpr (CLS == ADD64) { REP32(I_ADD64) }
        if constexpr (CLS == MUL64) { REP32(I_MUL64) }
        if constexpr (CLS == RCP64) { REP32(I_RCP64) }
        if constexpr (CLS == ADD32) { REP32(I_ADD32) }
        if constexpr (CLS == CND32) { REP32(I_CND32) }
        if constexpr (CLS == MOV32) { REP32(I_MOV32) }
        if constexpr (CLS == AND32) { REP32(I_AND32) }
        if constexpr (CLS == LSHL64) { REP32(I_LSHL64) }
        if constexpr (CLS == FMA32) { REP32(I_FMA32) }
        if constexpr (CLS == MIX_FMA64_ADD32) { REP8(I_MIXA) REP8(I_MIXA) }
        if constexpr (CLS == MIX_FMA64_CND32) { REP8(I_MIXB) REP8(I_MIXB) }
        if constexpr (CLS == MIX_ADD64_MOV32) { REP8(I_MIXC) REP8(I_MIXC) }
        if constexpr (CLS == CMP64_CND) { REP8(I_CMPC) REP8(I_CMPC) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double acc = 0.0;
    #pragma unroll
    for (int k = 0; k < 8; ++k) acc += d[k] + (double)u[k] + (double)f[k];
    out[blockIdx.x * 64 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int CLS, int WPS>
void run(double* dout, unsigned long long* dcyc, int iters)
{
    const int blocks = 1024 * WPS;                     // one-wave blocks: WPS waves on each of 256 CUs x 4 SIMDs
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((ub<CLS, WPS>), dim3(blocks), dim3(64), 0, 0, dout, dcyc, iters, 1.0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), dcyc, blocks * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += (double)v; mean /= blocks;
    const double n = (double)iters * 32;               // instructions per wave
    printf("  %-38s W=%d  %7.3f ms  %6.2f shader cycles per instruction and SIMD (s_memtime)   %6.2f at 2.4 GHz from the wall clock\n",
           kName[CLS], WPS, ms, mean / (n * WPS), ms * 1e-3 * 2.4e9 / (n * WPS));
    hipEventDestroy(e0); hipEventDestroy(e1);
}

template <int CLS>
void run_all(double* dout, unsigned long long* dcyc, int iters)
{
    run<CLS, 1>(dout, dcyc, iters); run<CLS, 2>(dout, dcyc, iters); run<CLS, 3>(dout, dcyc, iters); run<CLS, 4>(dout, dcyc, iters);
}

int main()
{
    double* dout; unsigned long long* dcyc;
    hipMalloc(&dout, (size_t)4096 * 64 * 8); hipMalloc(&dcyc, 4096 * 8);
    const int iters = 20000;
    run_all<FMA64>(dout, dcyc, iters); run_all<ADD64>(dout, dcyc, iters); run_all<MUL64>(dout, dcyc, iters); run_all<RCP64>(dout, dcyc, iters / 4);
    run_all<ADD32>(dout, dcyc, iters); run_all<CND32>(dout, dcyc, iters); run_all<MOV32>(dout, dcyc, iters); run_all<AND32>(dout, dcyc, iters);
    run_all<LSHL64>(dout, dcyc, iters); run_all<FMA32>(dout, dcyc, iters);
    run_all<MIX_FMA64_ADD32>(dout, dcyc, iters); run_all<MIX_FMA64_CND32>(dout, dcyc, iters); run_all<MIX_ADD64_MOV32>(dout, dcyc, iters); run_all<CMP64_CND>(dout, dcyc, iters);
    return 0;
}
