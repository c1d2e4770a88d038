// es_tx.hip -- frame generator: the +-1 chip sequence of a frame and the finishing step after
// the band-pass, so that whole batches of synthetic frames are produced on the device:
//   polar code bits (es_polar_encode_batch) + PN row (es_schedule_batch) + counter -> 63 preamble | 128 header | 1024
//   spread payload chips (rtwm/embedder.py:78-115 _make_frame_chips: header = lo16(ctr) MSB first, each bit repeated 8
//   times, times the header PN pn_bits(0, 128); payload chip i = code bit i times PN bit 191 + i);
//   band-pass with zero initial state carried through the frame (rtwm/embedder.py:117-136: two lfilter calls with the
//   state handed over = one pass; es_bpf_batch, bit-exact SciPy order); then the peak rule of :137-141
//   (peak = max|chips| + 1e-12; if peak > 3: chips *= 1/peak) and the cast to float32.
#include "es_internal.h"

namespace {

__global__ __launch_bounds__(256) void es_tx_symbols_kernel(const uint8_t* __restrict__ code, const uint8_t* __restrict__ pn_rows,
        const uint32_t* __restrict__ ctr, unsigned long long pre_bits, const uint8_t* __restrict__ hdr_pn_g, long long B,
        float* __restrict__ sym)
{
    __shared__ uint8_t hdr_pn[16];
    if (threadIdx.x < 16) hdr_pn[threadIdx.x] = hdr_pn_g[threadIdx.x];
    __syncthreads();
    for (long long f = blockIdx.x; f < B; f += gridDim.x) {
        const uint32_t lo16 = ctr[f] & 0xFFFFu;
        const uint8_t* pn = pn_rows + f * ES_PN_BYTES;
        for (int i = threadIdx.x; i < ES_FRAME_LEN; i += 256) {
            float v;
            if (i < ES_PRE_L) {
                v = ((pre_bits >> (63 - i)) & 1ull) ? 1.0f : -1.0f;                       // bit i of the packed MLS, MSB first
            } else if (i < ES_PRE_L + ES_HDR_L) {
                const int k = i - ES_PRE_L;
                const uint32_t hb = (lo16 >> (15 - (k >> 3))) & 1u;                       // 16 bits, each repeated 8 times
                const uint32_t pb = (hdr_pn[k >> 3] >> (7 - (k & 7))) & 1u;
                v = (hb ? 1.0f : -1.0f) * (pb ? 1.0f : -1.0f);
            } else {
                const int k = i - (ES_PRE_L + ES_HDR_L);
                const uint32_t pb = (pn[i >> 3] >> (7 - (i & 7))) & 1u;                   // PN bit 191 + k
                v = (code[f * ES_POLAR_N + k] ? 1.0f : -1.0f) * (pb ? 1.0f : -1.0f);
            }
            sym[f * ES_FRAME_LEN + i] = v;
        }
    }
}

// one wave per frame: peak over the float64 chips, optional rescale, cast
__global__ __launch_bounds__(256) void es_tx_finish_kernel(const double* __restrict__ y, long long B, float* __restrict__ out)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long stride = (long long)gridDim.x * 4;
    for (long long f = (long long)blockIdx.x * 4 + wv; f < B; f += stride) {
        const double* yr = y + f * ES_FRAME_LEN;
        double m = 0.0;
        for (int i = lane; i < ES_FRAME_LEN; i += 64) { const double a = __builtin_fabs(yr[i]); m = (a > m || a != a) ? a : m; }   // NaN wins, as np.max
        #pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { const double t = __shfl_xor(m, o); m = (t > m || t != t) ? t : m; }
        const double peak = m + 1e-12;
        const bool scale = peak > 3.0;
        const double inv = 1.0 / peak;
        for (int i = lane; i < ES_FRAME_LEN; i += 64) {
            double v = yr[i];
            if (scale) v = v * inv;
            out[f * ES_FRAME_LEN + i] = (float)v;
        }
    }
}

}  // namespace

int es_launch_tx_frames(es_ctx* ctx, const uint8_t* code, const uint8_t* pn_rows, const uint8_t* band, const uint32_t* ctr,
                        unsigned long long pre_bits, const uint8_t* hdr_pn16, int64_t B, double* y_ws, float* frames, hipStream_t st)
{
    if (!ctx->d_hdr_pn) ES_HIP_CHECK(ctx, hipMalloc(&ctx->d_hdr_pn, 16));
    ES_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_hdr_pn, hdr_pn16, 16, hipMemcpyHostToDevice, st));
    long long blocks = B;
    const long long cap = (long long)ctx->num_cu * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(es_tx_symbols_kernel, dim3((unsigned)blocks), dim3(256), 0, st, code, pn_rows, ctr, pre_bits,
                       ctx->d_hdr_pn, (long long)B, frames);
    ES_HIP_CHECK(ctx, hipGetLastError());
    const int rc = es_launch_bpf(ctx, frames, ES_DTYPE_F32, B, ES_FRAME_LEN, band, y_ws, nullptr, st);
    if (rc != ES_OK) return rc;
    long long fb = (B + 3) / 4;
    if (fb > cap) fb = cap;
    hipLaunchKernelGGL(es_tx_finish_kernel, dim3((unsigned)fb), dim3(256), 0, st, y_ws, (long long)B, frames);
    ES_HIP_CHECK(ctx, hipGetLastError());
    return ES_OK;
}
