// micro-benchmark: how busy does the vector unit get on the list decoder's f() alone -- no slab, no sort -- at W waves per SIMD?
// (every lane runs a chain of f evaluations on register operands; the exp table in LDS as in the kernels)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "es_math.h"
static const uint64_t kTab[256] = ES_EXP_TAB_INIT;

template <int WPS>
__global__ __launch_bounds__(64, WPS) void ub(const uint64_t* tabg, const double* in, double* out, int iters)
{
    __shared__ __attribute__((aligned(16))) uint64_t tab[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) tab[i] = tabg[i];
    __syncthreads();
    const int g = blockIdx.x * 64 + threadIdx.x;
    double x = in[g & 127], y = in[(g + 37) & 127], acc = 0.0;
    for (int i = 0; i < iters; ++i) {
        const double r = es_polar_f(x, y, tab);
        acc += r; x = y * 0.75 + r * 0.01; y = r - x * 0.5;        // (new operands each time, as in the tree)
    }
    out[g] = acc;
}

int main()
{
    uint64_t* dtab; double *din, *dout;
    hipMalloc(&dtab, sizeof kTab); hipMemcpy(dtab, kTab, sizeof kTab, hipMemcpyHostToDevice);
    std::vector<double> h(128);
    for (int i = 0; i < 128; ++i) h[i] = ((i * 7919) % 1000) / 100.0 - 5.0;
    hipMalloc(&din, 1024); hipMemcpy(din, h.data(), 1024, hipMemcpyHostToDevice);
    const int iters = 20000;
    for (int wps : {1, 2, 3, 4}) {
        const int blocks = 1024 * wps;                  // one-wave blocks: wps waves per SIMD on 256 CUs x 4 SIMDs
        hipMalloc(&dout, (size_t)blocks * 64 * 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (wps == 1) hipLaunchKernelGGL(ub<1>, dim3(blocks), dim3(64), 0, 0, dtab, din, dout, iters);
            if (wps == 2) hipLaunchKernelGGL(ub<2>, dim3(blocks), dim3(64), 0, 0, dtab, din, dout, iters);
            if (wps == 3) hipLaunchKernelGGL(ub<3>, dim3(blocks), dim3(64), 0, 0, dtab, din, dout, iters);
            if (wps == 4) hipLaunchKernelGGL(ub<4>, dim3(blocks), dim3(64), 0, 0, dtab, din, dout, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double f_per_s = (double)blocks * iters / (ms * 1e-3);      // wave-level f evaluations per second
        printf("%d wave(s) per SIMD: %.2f ms, %.1f G wave-f/s, %.0f cycles per f and SIMD at 2.4 GHz\n", wps, ms, f_per_s / 1e9, 1024.0 * 2.4e9 / f_per_s);
        hipFree(dout);
    }
    return 0;
}
