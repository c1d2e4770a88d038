// es_resample.hip -- input conditioning: the polyphase FIR of resample_to (rtwm/utils.py:58-66 =
// scipy.signal.resample_poly -> upfirdn, mode 'constant').  The filter design and padding arithmetic stay on the host
// (they are Python in SciPy); this is upfirdn's inner loop: one lane per output sample, the products x[i] * h[...] added
// to an accumulator that starts at 0, in ascending input index, multiply and add rounded separately, in the arithmetic
// of SciPy's output type (float32 for float32 signals, else float64).  Consecutive lanes read consecutive phases of the
// filter and (nearly) the same input samples: all of it is served from L2.
// Build with -ffp-contract=off.
#include "es_internal.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void es_resample_kernel(const T* __restrict__ x, long long B, long long n_x,
        const T* __restrict__ h_tf, int hpp, int up, int down, long long y0, long long n_out, T* __restrict__ out)
{
    const long long total = B * n_out;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
        const long long r = g / n_out, k = g - r * n_out;
        const long long yy = y0 + k;
        const long long t = (yy * down) % up, x_idx = (yy * down) / up;
        long long lo = x_idx - hpp + 1, hi = x_idx;
        long long hidx = t * hpp;
        if (lo < 0) { hidx -= lo; lo = 0; }
        if (hi > n_x - 1) hi = n_x - 1;
        const T* xr = x + r * n_x;
        T acc = (T)0;
        for (long long i = lo; i <= hi; ++i) { const T p = xr[i] * h_tf[hidx++]; acc = acc + p; }
        out[g] = acc;
    }
}

}  // namespace

int es_launch_resample(es_ctx* ctx, const void* x, int dtype, int64_t B, int64_t n_x, const void* h_tf, int hpp, int up, int down,
                       int64_t y0, int64_t n_out, void* out, hipStream_t st)
{
    long long blocks = (B * n_out + 255) / 256;
    const long long cap = (long long)ctx->num_cu * 32;
    if (blocks > cap) blocks = cap;
    if (dtype == ES_DTYPE_F32)
        hipLaunchKernelGGL(es_resample_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)x, (long long)B,
                           (long long)n_x, (const float*)h_tf, hpp, up, down, (long long)y0, (long long)n_out, (float*)out);
    else
        hipLaunchKernelGGL(es_resample_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, st, (const double*)x, (long long)B,
                           (long long)n_x, (const double*)h_tf, hpp, up, down, (long long)y0, (long long)n_out, (double*)out);
    ES_HIP_CHECK(ctx, hipGetLastError());
    return ES_OK;
}
