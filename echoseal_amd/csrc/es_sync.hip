// es_sync.hip -- sync stage of the detector for gfx950: band-pass, 63-chip normalised
// cross-correlation, median/MAD threshold and non-maximum suppression.
//
//   es_bpf_kernel    rtwm/detector.py:59-60   y = lfilter(b, a, x.astype(float32))  (float64)
//   es_xcorr_kernel  rtwm/detector.py:76-79   corr = correlate(y,tpl,'valid') / (sqrt(conv(y^2,1)) + 1e-12)
//   es_pick_kernel   rtwm/detector.py:83-99   thr = min(med + 4.5*1.4826*MAD, 0.95); NMS +-607; top-5 fallback
//
// All arithmetic is float64 because the reference is (SciPy promotes on the first multiply) and
// sync offsets must match it exactly.  Operation order (tests/ compare bit-for-bit with
// oracle/c/eso_dsp.c): the IIR uses SciPy's direct-form-II-transposed loop with separate
// multiply and add; the correlation numerator is an FMA chain over ascending tap index; the
// window energy is a plain sum of squares over ascending index.
//
// Build with -ffp-contract=off.
#include "es_internal.h"

namespace {

__device__ __forceinline__ void wave_fence_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------------ BPF
// The recursion is serial in time, so parallelism is ACROSS records: one lane per record, 64
// records per wave.  A [64 records] x [32 samples] tile is staged through LDS both ways so that
// HBM sees whole 128-byte (input) / 256-byte (output) row segments: lanes 0..31 cover one row
// segment, two rows per wave instruction.  LDS rows are padded to 33 words: lane r reading
// column t hits bank (33 r + t) mod 32 = (r + t) mod 32, conflict free.
constexpr int BPF_TT = 32;          // samples per tile
constexpr int BPF_WAVES = 2;        // waves per block (LDS: 2 x (8.4 + 16.9) KB)

template <bool I16>
__global__ __launch_bounds__(64 * BPF_WAVES) void es_bpf_kernel(const void* __restrict__ frames,
        long long B, int T, const uint8_t* __restrict__ band, const es_band_tables* __restrict__ tabs,
        double* __restrict__ y)
{
    __shared__ float  s_x[BPF_WAVES][64][BPF_TT + 1];
    __shared__ double s_y[BPF_WAVES][64][BPF_TT + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long rec0 = ((long long)blockIdx.x * BPF_WAVES + wv) * 64;
    if (rec0 >= B) return;
    const long long rec = rec0 + lane;
    const bool live = rec < B;
    const int bi = live ? band[rec] : 0;
    double cb[9], ca[9], z[8];
    #pragma unroll
    for (int k = 0; k < 9; ++k) { cb[k] = tabs->ba[bi][k]; ca[k] = tabs->ba[bi][9 + k]; }
    #pragma unroll
    for (int k = 0; k < 8; ++k) z[k] = 0.0;

    const int half = lane >> 5, col = lane & 31;
    for (int t0 = 0; t0 < T; t0 += BPF_TT) {
        // stage in: two record rows per instruction
        #pragma unroll 4
        for (int r = half; r < 64; r += 2) {
            const long long rr = rec0 + r;
            float v = 0.0f;
            if (rr < B && t0 + col < T) {
                if (I16) v = (float)((const int16_t*)frames)[rr * T + t0 + col] / 32767.0f;
                else v = ((const float*)frames)[rr * T + t0 + col];
            }
            s_x[wv][r][col] = v;
        }
        wave_fence_lds();
        const int nt = (T - t0 < BPF_TT) ? T - t0 : BPF_TT;
        for (int t = 0; t < nt; ++t) {
            const double xn = (double)s_x[wv][lane][t];
            const double yn = z[0] + cb[0] * xn;
            #pragma unroll
            for (int k = 0; k < 7; ++k) z[k] = (z[k + 1] + xn * cb[k + 1]) - yn * ca[k + 1];
            z[7] = xn * cb[8] - yn * ca[8];
            s_y[wv][lane][t] = yn;
        }
        wave_fence_lds();
        #pragma unroll 4
        for (int r = half; r < 64; r += 2) {
            const long long rr = rec0 + r;
            if (rr < B && t0 + col < T) y[rr * T + t0 + col] = s_y[wv][r][col];
        }
        wave_fence_lds();
    }
}

// ---------------------------------------------------------------------------------------- xcorr
// One 256-thread block per record.  The record (float64) is staged in LDS; every thread owns
// XC_R consecutive lags and slides a register window over the 62+XC_R samples they share, so each
// sample is read from LDS once per thread instead of once per lag.  XC_R is odd: lane l starts at
// sample l*XC_R, i.e. 8-byte word l*XC_R, and l*XC_R mod 32 is a permutation for odd XC_R, so the
// ds_read_b64 stream is bank-conflict free.  Template taps are wave-uniform (scalar loads).
constexpr int XC_R = 5;
constexpr int XC_THREADS = 256;

__global__ __launch_bounds__(XC_THREADS) void es_xcorr_kernel(const double* __restrict__ y, long long B,
        int T, const uint8_t* __restrict__ band, const es_band_tables* __restrict__ tabs,
        double* __restrict__ corr)
{
    extern __shared__ double s_rec[];                 // T doubles
    const int n_lags = T - (ES_PRE_L - 1);
    for (long long rec = blockIdx.x; rec < B; rec += gridDim.x) {
        const double* yr = y + rec * T;
        for (int i = threadIdx.x; i < T; i += XC_THREADS) s_rec[i] = yr[i];
        // wave-uniform band index -> template taps come through scalar loads
        const double* tpl = tabs->tpl[__builtin_amdgcn_readfirstlane((int)band[rec])];
        __syncthreads();
        for (int base = threadIdx.x * XC_R; base < n_lags; base += XC_THREADS * XC_R) {
            double num[XC_R], en[XC_R];
            #pragma unroll
            for (int r = 0; r < XC_R; ++r) { num[r] = 0.0; en[r] = 0.0; }
            // sample s = base + m contributes to lag base + r with tap k = m - r
            #pragma unroll
            for (int m = 0; m < ES_PRE_L - 1 + XC_R; ++m) {
                const int idx = base + m;
                const double v = (idx < T) ? s_rec[idx] : 0.0;
                const double v2 = v * v;
                #pragma unroll
                for (int r = 0; r < XC_R; ++r) {
                    const int k = m - r;
                    if (k >= 0 && k < ES_PRE_L) {
                        num[r] = __builtin_fma(v, tpl[k], num[r]);
                        en[r] = en[r] + v2;
                    }
                }
            }
            #pragma unroll
            for (int r = 0; r < XC_R; ++r)
                if (base + r < n_lags)
                    corr[rec * n_lags + base + r] = num[r] / (__builtin_sqrt(en[r]) + 1e-12);
        }
        __syncthreads();
    }
}

// ----------------------------------------------------------------------------------------- pick
// One 256-thread block per record; the correlation row is read from global memory (it is L1/L2
// resident: 9 KB per record).  Medians are exact order statistics found by an 8-bit-digit radix
// select over the monotone 64-bit image of the doubles; np.median's even-length case is the mean
// of the two middle order statistics.
constexpr int PK_THREADS = 256;

__device__ __forceinline__ uint64_t f64_key(double x)
{
    uint64_t b; __builtin_memcpy(&b, &x, 8);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ULL);
}
__device__ __forceinline__ double key_f64(uint64_t k)
{
    const uint64_t b = (k >> 63) ? (k & 0x7fffffffffffffffULL) : ~k;
    double x; __builtin_memcpy(&x, &b, 8); return x;
}

// k-th smallest (0-based) of v[i] (ABSDEV: |v[i] - center|).  All threads return the value.
template <bool ABSDEV>
__device__ double block_select(const double* __restrict__ v, int n, int k, double center,
                               uint32_t* s_hist, uint64_t* s_pref, int* s_k)
{
    uint64_t prefix = 0;
    int kk = k;
    for (int shift = 56; shift >= 0; shift -= 8) {
        s_hist[threadIdx.x] = 0;
        __syncthreads();
        const uint64_t himask = (shift == 56) ? 0ULL : (~0ULL << (shift + 8));
        for (int i = threadIdx.x; i < n; i += PK_THREADS) {
            double x = v[i];
            if (ABSDEV) x = __builtin_fabs(x - center);
            const uint64_t key = f64_key(x);
            if ((key & himask) == prefix) atomicAdd(&s_hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        {
            // exclusive prefix over the 256 digit bins: wave scan + 4 wave totals
            const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
            const uint32_t h = s_hist[threadIdx.x];
            uint32_t incl = h;
            #pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t up = __shfl_up(incl, o);
                if (lane >= o) incl += up;
            }
            __syncthreads();
            if (lane == 63) s_hist[wv] = incl;            // reuse bins 0..3 as wave totals
            __syncthreads();
            uint32_t basec = 0;
            for (int w = 0; w < wv; ++w) basec += s_hist[w];
            incl += basec;
            const uint32_t excl = incl - h;
            if ((int)excl <= kk && kk < (int)incl) {
                *s_k = kk - (int)excl;
                *s_pref = prefix | ((uint64_t)threadIdx.x << shift);
            }
        }
        __syncthreads();
        prefix = *s_pref;
        kk = *s_k;
        __syncthreads();
    }
    return key_f64(prefix);
}

template <bool ABSDEV>
__device__ double block_median(const double* v, int n, double center, uint32_t* s_hist,
                               uint64_t* s_pref, int* s_k)
{
    if (n & 1) return block_select<ABSDEV>(v, n, n / 2, center, s_hist, s_pref, s_k);
    const double lo = block_select<ABSDEV>(v, n, n / 2 - 1, center, s_hist, s_pref, s_k);
    const double hi = block_select<ABSDEV>(v, n, n / 2, center, s_hist, s_pref, s_k);
    return (lo + hi) / 2.0;
}

__global__ __launch_bounds__(PK_THREADS) void es_pick_kernel(const double* __restrict__ corr, long long B,
        int n, double* __restrict__ thr_out, int32_t* __restrict__ peaks, int32_t* __restrict__ npeaks)
{
    __shared__ uint32_t s_hist[256];
    __shared__ uint64_t s_pref;
    __shared__ int s_k;
    __shared__ int s_cnt[PK_THREADS + 1];
    __shared__ double s_bv[PK_THREADS];
    __shared__ int s_bi[PK_THREADS];
    __shared__ int s_taken[5];
    const int min_distance = ES_FRAME_LEN / 2;        // 607

    for (long long rec = blockIdx.x; rec < B; rec += gridDim.x) {
        const double* c = corr + rec * n;
        const double med = block_median<false>(c, n, 0.0, s_hist, &s_pref, &s_k);
        const double mad = block_median<true>(c, n, med, s_hist, &s_pref, &s_k) + 1e-12;
        double thr = med + 4.5 * 1.4826 * mad;
        if (0.95 < thr) thr = 0.95;

        // contiguous chunk per thread so that peaks come out in ascending order
        const int chunk = (n + PK_THREADS - 1) / PK_THREADS;
        const int i0 = threadIdx.x * chunk;
        const int i1 = (i0 + chunk < n) ? i0 + chunk : n;
        int mine[8]; int nm = 0, tot = 0;
        for (int i = i0; i < i1; ++i) {
            const double ci = c[i];
            if (ci < thr) continue;
            int lo = i - min_distance; if (lo < 0) lo = 0;
            int hi = i + min_distance + 1; if (hi > n) hi = n;
            bool is_peak = true;
            for (int j = lo; j < hi; ++j) if (c[j] > ci) { is_peak = false; break; }
            if (is_peak) { if (nm < 8) mine[nm++] = i; ++tot; }
        }
        s_cnt[threadIdx.x] = tot;
        __syncthreads();
        if (threadIdx.x == 0) {
            int acc = 0;
            for (int t = 0; t < PK_THREADS; ++t) { const int v = s_cnt[t]; s_cnt[t] = acc; acc += v; }
            s_cnt[PK_THREADS] = acc;
        }
        __syncthreads();
        const int total = s_cnt[PK_THREADS];
        const int off = s_cnt[threadIdx.x];
        // a thread can only hold more than 8 peaks when ties flood the row; later ones are dropped
        // from the list (the count is still exact) -- the detector reads at most 25 anyway.
        for (int k = 0; k < nm; ++k) if (off + k < ES_MAX_PEAKS) peaks[rec * ES_MAX_PEAKS + off + k] = mine[k];
        __syncthreads();

        if (total == 0) {
            // fallback: five largest correlations, descending; equal values -> higher index first
            const int kmax = n < 5 ? n : 5;
            for (int r = 0; r < kmax; ++r) {
                double bv = 0.0; int bidx = -1;
                for (int i = threadIdx.x; i < n; i += PK_THREADS) {
                    bool used = false;
                    for (int qd = 0; qd < r; ++qd) used |= (s_taken[qd] == i);
                    if (used) continue;
                    const double ci = c[i];
                    if (bidx < 0 || ci > bv || (ci == bv && i > bidx)) { bv = ci; bidx = i; }
                }
                s_bv[threadIdx.x] = bv; s_bi[threadIdx.x] = bidx;
                __syncthreads();
                for (int s = PK_THREADS / 2; s > 0; s >>= 1) {
                    if (threadIdx.x < s) {
                        const double ov = s_bv[threadIdx.x + s]; const int oi = s_bi[threadIdx.x + s];
                        const double mv = s_bv[threadIdx.x]; const int mi = s_bi[threadIdx.x];
                        if (oi >= 0 && (mi < 0 || ov > mv || (ov == mv && oi > mi))) {
                            s_bv[threadIdx.x] = ov; s_bi[threadIdx.x] = oi;
                        }
                    }
                    __syncthreads();
                }
                if (threadIdx.x == 0) { s_taken[r] = s_bi[0]; peaks[rec * ES_MAX_PEAKS + r] = s_bi[0]; }
                __syncthreads();
            }
            if (threadIdx.x == 0) npeaks[rec] = kmax | (1 << 30);
        } else if (threadIdx.x == 0) {
            npeaks[rec] = total;
        }
        if (threadIdx.x == 0) thr_out[rec] = thr;
        __syncthreads();
    }
}

}  // namespace

int es_launch_bpf(es_ctx* ctx, const void* frames, int dtype, int64_t B, int T, const uint8_t* band,
                  double* y, hipStream_t st)
{
    const long long recs_per_block = 64LL * BPF_WAVES;
    const unsigned blocks = (unsigned)((B + recs_per_block - 1) / recs_per_block);
    if (dtype == ES_DTYPE_I16)
        hipLaunchKernelGGL(es_bpf_kernel<true>, dim3(blocks), dim3(64 * BPF_WAVES), 0, st, frames,
                           (long long)B, T, band, ctx->d_tables, y);
    else
        hipLaunchKernelGGL(es_bpf_kernel<false>, dim3(blocks), dim3(64 * BPF_WAVES), 0, st, frames,
                           (long long)B, T, band, ctx->d_tables, y);
    ES_HIP_CHECK(ctx, hipGetLastError());
    return ES_OK;
}

int es_launch_xcorr(es_ctx* ctx, const double* y, int64_t B, int T, const uint8_t* band, double* corr,
                    hipStream_t st)
{
    const size_t lds = (size_t)T * sizeof(double);
    if (lds > 64 * 1024) { ctx->err = "es_xcorr_batch: record longer than 8192 samples"; return ES_EINVAL; }
    long long blocks = B;
    const long long cap = (long long)ctx->num_cu * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(es_xcorr_kernel, dim3((unsigned)blocks), dim3(XC_THREADS), lds, st, y, (long long)B,
                       T, band, ctx->d_tables, corr);
    ES_HIP_CHECK(ctx, hipGetLastError());
    return ES_OK;
}

int es_launch_pick(es_ctx* ctx, const double* corr, int64_t B, int n_lags, double* thr, int32_t* peaks,
                   int32_t* npeaks, hipStream_t st)
{
    long long blocks = B;
    const long long cap = (long long)ctx->num_cu * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(es_pick_kernel, dim3((unsigned)blocks), dim3(PK_THREADS), 0, st, corr, (long long)B,
                       n_lags, thr, peaks, npeaks);
    ES_HIP_CHECK(ctx, hipGetLastError());
    return ES_OK;
}
