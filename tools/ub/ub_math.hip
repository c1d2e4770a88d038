// micro-benchmark: dependent-chain latency (cycles) of the float64 primitives, one wave, one block
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "es_math.h"
static const uint64_t kTab[256] = ES_EXP_TAB_INIT;

template <int MODE>
__global__ void ub(const uint64_t* tabg, const double* in, double* out, unsigned long long* cyc, int iters)
{
    __shared__ __attribute__((aligned(16))) uint64_t tab[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) tab[i] = tabg[i];
    __syncthreads();
    double x = in[threadIdx.x], y = in[64 + threadIdx.x];
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) x = es_exp(-__builtin_fabs(x) * 0.5 - 0.1, tab) + y;           // exp
        if (MODE == 1) x = es_log1p(__builtin_fabs(x) * 0.25 + 0.01) + y * 0.1;       // log1p
        if (MODE == 2) x = es_softplus_neg(-__builtin_fabs(x) - 0.05, tab) + y;       // softplus
        if (MODE == 3) x = es_polar_f(x, y, tab) + y * 0.5;                           // f
        if (MODE == 6) { double a1[1] = {x}, b1[1] = {y}, o1[1]; es_polar_fN<1>(a1, b1, tab, o1); x = o1[0] + y * 0.5; }
        if (MODE == 7) { double a2[2] = {x, x + 0.3}, b2[2] = {y, y - 0.2}, o2[2]; es_polar_fN<2>(a2, b2, tab, o2); x = o2[0] + o2[1] * 0.5; }
        if (MODE == 4) x = x / (y + 2.0) + 1.0;                                       // one division
        if (MODE == 5) x = __builtin_fma(x, 0.999, y);                                // one fma
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main()
{
    uint64_t* dtab; double *din, *dout; unsigned long long* dc;
    hipMalloc(&dtab, sizeof kTab); hipMemcpy(dtab, kTab, sizeof kTab, hipMemcpyHostToDevice);
    std::vector<double> h(128);
    for (int i = 0; i < 128; ++i) h[i] = ((i * 7919) % 1000) / 100.0 - 5.0;
    hipMalloc(&din, 128 * 8); hipMemcpy(din, h.data(), 128 * 8, hipMemcpyHostToDevice);
    hipMalloc(&dout, 64 * 8); hipMalloc(&dc, 8);
    const int iters = 2000;
    const char* names[8] = {"exp", "log1p", "softplus", "polar_f", "div", "fma", "fN<1>", "fN<2> (2 f)"};
    for (int mode = 0; mode < 8; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            switch (mode) {
                case 0: hipLaunchKernelGGL(ub<0>, 1, 64, 0, 0, dtab, din, dout, dc, iters); break;
                case 1: hipLaunchKernelGGL(ub<1>, 1, 64, 0, 0, dtab, din, dout, dc, iters); break;
                case 2: hipLaunchKernelGGL(ub<2>, 1, 64, 0, 0, dtab, din, dout, dc, iters); break;
                case 3: hipLaunchKernelGGL(ub<3>, 1, 64, 0, 0, dtab, din, dout, dc, iters); break;
                case 4: hipLaunchKernelGGL(ub<4>, 1, 64, 0, 0, dtab, din, dout, dc, iters); break;
                case 5: hipLaunchKernelGGL(ub<5>, 1, 64, 0, 0, dtab, din, dout, dc, iters); break;
                case 6: hipLaunchKernelGGL(ub<6>, 1, 64, 0, 0, dtab, din, dout, dc, iters); break;
                case 7: hipLaunchKernelGGL(ub<7>, 1, 64, 0, 0, dtab, din, dout, dc, iters); break;
            }
            hipDeviceSynchronize();
        }
        unsigned long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
        printf("%-9s %8.1f cycles per dependent call\n", names[mode], (double)c / iters);
    }
    return 0;
}
