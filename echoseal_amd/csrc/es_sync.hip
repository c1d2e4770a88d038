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
        double* __restrict__ y, float* __restrict__ y32)
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
                if (I16) v = (float)((const int16_t*)frames)[rr * T + t0 + col] * (1.0f / 32768.0f);   // PCM16 as soundfile.read hands it to the reference (rx_app.py:26): exact in float32
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
            if (rr < B && t0 + col < T) {
                const double v = s_y[wv][r][col];
                y[rr * T + t0 + col] = v;
                if (y32) y32[rr * T + t0 + col] = (float)v;
            }
        }
        wave_fence_lds();
    }
}

// ---------------------------------------------------------------------------------------- BPF (quad)
// Same recursion, four lanes per record: lane j of a quad owns delay elements z[2j], z[2j+1] and
// the coefficient pairs that update them, so a wave holds 16 records and a batch of B records
// gives B/16 waves instead of B/64 -- the better shape up to a few hundred thousand records,
// where the lane-per-record kernel cannot fill the chip.  Per sample: every lane forms
// z[2j] + b0*x, lane 0's value (= y) is broadcast with a quad_perm DPP move, z[2j+2] comes from
// the right-hand neighbour with another DPP move; arithmetic and its order are unchanged
// (SciPy's direct-form-II-transposed loop, separate multiply and add).
// The next tile of input is fetched into registers while the current one is filtered.
constexpr int BQ_TT = 32;
constexpr int BQ_RECS = 16;                         // records per wave
constexpr int BQ_WAVES = 4;

__device__ __forceinline__ double dpp_quad_f64(double v, const int ctrl_bcast0)
{
    uint64_t u; __builtin_memcpy(&u, &v, 8);
    int lo = (int)(uint32_t)u, hi = (int)(uint32_t)(u >> 32);
    if (ctrl_bcast0) {
        lo = __builtin_amdgcn_mov_dpp(lo, 0x00, 0xf, 0xf, true);      // quad_perm [0,0,0,0]
        hi = __builtin_amdgcn_mov_dpp(hi, 0x00, 0xf, 0xf, true);
    } else {
        lo = __builtin_amdgcn_mov_dpp(lo, 0xF9, 0xf, 0xf, true);      // quad_perm [1,2,3,3]
        hi = __builtin_amdgcn_mov_dpp(hi, 0xF9, 0xf, 0xf, true);
    }
    u = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
    double r; __builtin_memcpy(&r, &u, 8); return r;
}

template <bool I16>
__global__ __launch_bounds__(64 * BQ_WAVES) void es_bpf_quad_kernel(const void* __restrict__ frames,
        long long B, int T, const uint8_t* __restrict__ band, const es_band_tables* __restrict__ tabs,
        double* __restrict__ y, float* __restrict__ y32)
{
    __shared__ float  s_x[BQ_WAVES][BQ_RECS][BQ_TT + 1];
    __shared__ double s_y[BQ_WAVES][BQ_RECS][BQ_TT + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __builtin_amdgcn_s_setprio(3);      // 16 records per wave and a serial recurrence: never queue behind a long-running wave
    const long long rec0 = ((long long)blockIdx.x * BQ_WAVES + wv) * BQ_RECS;
    if (rec0 >= B) return;
    const int rloc = lane >> 2, j = lane & 3;
    const long long rec = rec0 + rloc;
    const int bi = (rec < B) ? band[rec] : 0;
    const double b0 = tabs->ba[bi][0];
    const double b_lo = tabs->ba[bi][2 * j + 1], a_lo = tabs->ba[bi][9 + 2 * j + 1];
    const double b_hi = tabs->ba[bi][2 * j + 2], a_hi = tabs->ba[bi][9 + 2 * j + 2];
    double z_lo = 0.0, z_hi = 0.0;

    // staging: lanes 0..31 cover one record row segment, two rows per wave instruction
    const int half = lane >> 5, col = lane & 31;
    float pre[BQ_RECS / 2];
    auto fetch = [&](int t0) {
        #pragma unroll
        for (int i = 0; i < BQ_RECS / 2; ++i) {
            const long long rr = rec0 + 2 * i + half;
            float v = 0.0f;
            if (rr < B && t0 + col < T) {
                if (I16) v = (float)((const int16_t*)frames)[rr * T + t0 + col] * (1.0f / 32768.0f);   // PCM16 as soundfile.read hands it to the reference (rx_app.py:26): exact in float32
                else v = ((const float*)frames)[rr * T + t0 + col];
            }
            pre[i] = v;
        }
    };
    fetch(0);
    for (int t0 = 0; t0 < T; t0 += BQ_TT) {
        #pragma unroll
        for (int i = 0; i < BQ_RECS / 2; ++i) s_x[wv][2 * i + half][col] = pre[i];
        wave_fence_lds();
        if (t0 + BQ_TT < T) fetch(t0 + BQ_TT);          // in flight while this tile is filtered
        // samples past the end of the record are zeros (their outputs are never stored), so the
        // tile is always walked in full, eight samples at a time: the eight LDS reads are issued
        // up front and the eight results leave through LDS afterwards, keeping memory latency out
        // of the serial recurrence.
        #pragma unroll 1
        for (int tb = 0; tb < BQ_TT; tb += 8) {
            float xs[8]; double ys[8];
            #pragma unroll
            for (int u = 0; u < 8; ++u) xs[u] = s_x[wv][rloc][tb + u];
            #pragma unroll
            for (int u = 0; u < 8; ++u) {
                const double xn = (double)xs[u];
                const double yn = dpp_quad_f64(z_lo + b0 * xn, 1);        // lane 0: z[0] + b[0] x
                const double z_nb = dpp_quad_f64(z_lo, 0);                 // z[2j+2] (old value)
                const double t_hi = xn * b_hi;
                const double u_hi = (j == 3) ? t_hi : z_nb + t_hi;         // z[7] has no upper neighbour
                const double n_lo = (z_hi + xn * b_lo) - yn * a_lo;
                const double n_hi = u_hi - yn * a_hi;
                z_lo = n_lo; z_hi = n_hi;
                ys[u] = yn;
            }
            if (j == 0) {
                #pragma unroll
                for (int u = 0; u < 8; ++u) s_y[wv][rloc][tb + u] = ys[u];
            }
        }
        wave_fence_lds();
        #pragma unroll
        for (int i = 0; i < BQ_RECS / 2; ++i) {
            const long long rr = rec0 + 2 * i + half;
            if (rr < B && t0 + col < T) {
                const double v = s_y[wv][2 * i + half][col];
                y[rr * T + t0 + col] = v;
                if (y32) y32[rr * T + t0 + col] = (float)v;   // what _llr and the f32 correlation screen read
            }
        }
        wave_fence_lds();
    }
}

// ---------------------------------------------------------------------------------------- xcorr
// One WAVE per (record, segment of 64 x 19 = 1216 lags).  The 1216 + 62 samples the segment needs
// are staged in LDS with coalesced 8-byte loads; lane l then owns the chunk of XC_R = 19
// consecutive lags starting at 19 l and walks the 81 samples they share ONCE (register sliding
// window, fully unrolled): per template tap one ds_read_b64 feeds 19 FMAs whose tap operand is a
// scalar register.  19 is odd, so lane l reads 8-byte word 19 l + m and (19 l) mod 32 is a
// permutation: the read stream is bank-conflict free.  The window energy comes from all-positive
// partial sums shared by the chunk (core / head / tail, see oracle/c/eso_dsp.c) instead of 63 adds
// per lag.  Results go back through the same LDS buffer so that HBM sees whole rows.
// FP64 VALU work per lag: 63 FMA + ~7 for the energy + sqrt/div; no MFMA (no shared operand).
constexpr int XC_R = 19;
constexpr int XC_SEG = 64 * XC_R;                  // lags per wave
constexpr int XC_NS = XC_SEG + ES_PRE_L - 1;       // samples per wave: 1278
constexpr int XC_WAVES = 4;

__global__ __launch_bounds__(64 * XC_WAVES) void es_xcorr_kernel(const double* __restrict__ y, long long B,
        int T, const uint8_t* __restrict__ band, const es_band_tables* __restrict__ tabs,
        double* __restrict__ corr, const uint8_t* __restrict__ only_flagged, const int* __restrict__ nflag)
{
    if (nflag && *nflag == 0) return;                          // redo pass with nothing flagged: leave at once
    __shared__ double s_buf[XC_WAVES][XC_NS + 2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double* s = s_buf[wv];
    const int n_lags = T - (ES_PRE_L - 1);
    const int nseg = (n_lags + XC_SEG - 1) / XC_SEG;
    const long long n_items = B * nseg;
    const long long stride = (long long)gridDim.x * XC_WAVES;
    for (long long item = (long long)blockIdx.x * XC_WAVES + wv; item < n_items; item += stride) {
        const long long rec = item / nseg;
        if (only_flagged && !only_flagged[rec]) continue;      // redo pass: only records the f32 screen gave up on
        const int lag0 = (int)(item % nseg) * XC_SEG;
        const double* yr = y + rec * T + lag0;
        const int nsamp = (T - lag0 < XC_NS) ? T - lag0 : XC_NS;
        {   // all 20 row loads are issued before the first one is consumed (one HBM round trip, not 20)
            double stage[(XC_NS + 63) / 64];
            #pragma unroll
            for (int u = 0; u < (XC_NS + 63) / 64; ++u) { const int i = lane + 64 * u; stage[u] = (i < nsamp) ? yr[i] : 0.0; }
            #pragma unroll
            for (int u = 0; u < (XC_NS + 63) / 64; ++u) { const int i = lane + 64 * u; if (i < XC_NS) s[i] = stage[u]; }
        }
        // wave-uniform band index -> template taps come through scalar loads
        const double* tpl = tabs->tpl[__builtin_amdgcn_readfirstlane((int)band[rec])];
        wave_fence_lds();

        const double* w = s + lane * XC_R;
        double num[XC_R];
        #pragma unroll
        for (int r = 0; r < XC_R; ++r) num[r] = 0.0;
        // en[r] first collects head[r] (descending partial sums of the first 18 squares), then
        // + core (squares 18..62, ascending), then + tail (squares 63..62+r, ascending): all >= 0.
        double en[XC_R], sq_head[XC_R - 1];
        double core = 0.0, tail_run = 0.0;
        // sample m meets lag r at tap k = m - r (0 <= k < 63)
        #define XC_FMAS(m, v)                                                                   \
            _Pragma("unroll") for (int r = 0; r < XC_R; ++r) {                                  \
                const int k = (m) - r;                                                          \
                if (k >= 0 && k < ES_PRE_L) num[r] = __builtin_fma((v), tpl[k], num[r]);        \
            }
        #pragma unroll
        for (int m = 0; m < XC_R - 1; ++m) {                     // samples 0..17: head squares
            const double v = w[m];
            sq_head[m] = v * v;
            XC_FMAS(m, v)
        }
        en[XC_R - 1] = 0.0;
        #pragma unroll
        for (int r = XC_R - 2; r >= 0; --r) en[r] = en[r + 1] + sq_head[r];
        #pragma unroll
        for (int m = XC_R - 1; m < ES_PRE_L; ++m) {              // samples 18..62: common core
            const double v = w[m];
            core = core + v * v;
            XC_FMAS(m, v)
        }
        #pragma unroll
        for (int r = 0; r < XC_R; ++r) en[r] = en[r] + core;
        #pragma unroll
        for (int m = ES_PRE_L; m < ES_PRE_L - 1 + XC_R; ++m) {   // samples 63..80: m = 62 + r closes lag r
            const double v = w[m];
            tail_run = tail_run + v * v;
            en[m - (ES_PRE_L - 1)] = en[m - (ES_PRE_L - 1)] + tail_run;
            XC_FMAS(m, v)
        }
        #undef XC_FMAS
        wave_fence_lds();                           // every lane has finished reading its window
        #pragma unroll
        for (int r = 0; r < XC_R; ++r) s[lane * XC_R + r] = num[r] / (__builtin_sqrt(en[r]) + 1e-12);
        wave_fence_lds();
        const int nl = (n_lags - lag0 < XC_SEG) ? n_lags - lag0 : XC_SEG;
        double* cr = corr + rec * n_lags + lag0;
        for (int i = lane; i < nl; i += 64) cr[i] = s[i];
        wave_fence_lds();
    }
}

// ----------------------------------------------------------------------------------------- pick
// One 256-thread block per record.  The correlation row is staged in LDS when it fits (<= 4096
// lags; longer recordings are read from global/L2).  Medians are exact order statistics found by
// an 8-bit-digit radix select over the monotone 64-bit image of the doubles (LDS histogram +
// wave scan); np.median's even-length case is the mean of the two middle order statistics.
// NMS: lags >= thr are marked in an LDS bitmap and visited in ascending order; for each one the
// whole block scans its +-607 window (a handful of loads per thread + one block vote), so peaks
// come out sorted and a flood of equal values cannot serialise on one thread.
constexpr int PK_THREADS = 256;
constexpr int PK_LDS_N = 4096;

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
// (NT threads per block: 256, or 1024 for rows that do not fit LDS -- a 5 s recording is 240 000 lags per band)
template <bool ABSDEV, int NT>
__device__ double block_select(const double* v, int n, int k, double center,
                               uint32_t* s_hist, uint64_t* s_pref, int* s_k)
{
    uint64_t prefix = 0;
    int kk = k;
    for (int shift = 56; shift >= 0; shift -= 8) {
        if (threadIdx.x < 256) s_hist[threadIdx.x] = 0;
        __syncthreads();
        const uint64_t himask = (shift == 56) ? 0ULL : (~0ULL << (shift + 8));
        for (int i = threadIdx.x; i < n; i += NT) {
            double x = v[i];
            if (ABSDEV) x = __builtin_fabs(x - center);
            const uint64_t key = f64_key(x);
            if ((key & himask) == prefix) atomicAdd(&s_hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        {
            // exclusive prefix over the 256 digit bins: wave scan + 4 wave totals (the first four waves; barriers by all)
            const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
            const bool bin = threadIdx.x < 256;
            const uint32_t h = bin ? s_hist[threadIdx.x] : 0u;
            uint32_t incl = h;
            #pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t up = __shfl_up(incl, o);
                if (lane >= o) incl += up;
            }
            __syncthreads();
            if (bin && lane == 63) s_hist[wv] = incl;     // reuse bins 0..3 as wave totals
            __syncthreads();
            uint32_t basec = 0;
            for (int w = 0; w < wv && w < 4; ++w) basec += s_hist[w];
            incl += basec;
            const uint32_t excl = incl - h;
            if (bin && (int)excl <= kk && kk < (int)incl) {
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

template <bool ABSDEV, int NT>
__device__ double block_median(const double* v, int n, double center, uint32_t* s_hist,
                               uint64_t* s_pref, int* s_k)
{
    if (n & 1) return block_select<ABSDEV, NT>(v, n, n / 2, center, s_hist, s_pref, s_k);
    const double lo = block_select<ABSDEV, NT>(v, n, n / 2 - 1, center, s_hist, s_pref, s_k);
    const double hi = block_select<ABSDEV, NT>(v, n, n / 2, center, s_hist, s_pref, s_k);
    return (lo + hi) / 2.0;
}

template <bool IN_LDS, int NT>
__global__ __launch_bounds__(NT) void es_pick_kernel(const double* __restrict__ corr, long long B,
        int n, double* __restrict__ thr_out, int32_t* __restrict__ peaks, int32_t* __restrict__ npeaks,
        const uint8_t* __restrict__ only_flagged, const int* __restrict__ nflag)
{
    if (nflag && *nflag == 0) return;                          // redo pass with nothing flagged: leave at once
    __shared__ double s_row[IN_LDS ? PK_LDS_N : 1];
    __shared__ uint32_t s_hist[256];
    __shared__ uint64_t s_pref;
    __shared__ int s_k;
    __shared__ double s_bv[NT];
    __shared__ int s_bi[NT];
    __shared__ int s_taken[5];
    __shared__ int s_flag;
    __shared__ uint32_t s_cnt;
    const int min_distance = ES_FRAME_LEN / 2;        // 607

    for (long long rec = blockIdx.x; rec < B; rec += gridDim.x) {
        if (only_flagged && !only_flagged[rec]) continue;
        const double* cg = corr + rec * n;
        const double* c = cg;
        if (IN_LDS) {
            for (int i = threadIdx.x; i < n; i += NT) s_row[i] = cg[i];
            c = s_row;
        }
        __syncthreads();
        const double med = block_median<false, NT>(c, n, 0.0, s_hist, &s_pref, &s_k);
        const double mad = block_median<true, NT>(c, n, med, s_hist, &s_pref, &s_k) + 1e-12;
        double thr = med + 4.5 * 1.4826 * mad;
        if (0.95 < thr) thr = 0.95;

        // ascending scan over lags >= thr; each candidate is checked by the whole block
        int total = 0;
        for (int base = 0; base < n; base += NT) {
            const int i = base + threadIdx.x;
            const bool cand = (i < n) && !(c[i] < thr);
            unsigned long long mask[NT / 64];
            if (threadIdx.x == 0) s_cnt = 0;
            __syncthreads();
            const unsigned long long bal = __ballot(cand);
            if ((threadIdx.x & 63) == 0) { ((unsigned long long*)s_bv)[threadIdx.x >> 6] = bal; if (bal) atomicOr(&s_cnt, 1u); }
            __syncthreads();
            if (s_cnt == 0) continue;                  // no candidate among these 256 lags (uniform)
            #pragma unroll
            for (int w = 0; w < NT / 64; ++w) mask[w] = ((unsigned long long*)s_bv)[w];
            __syncthreads();
            for (int w = 0; w < NT / 64; ++w) {
                unsigned long long m = mask[w];
                while (m) {                            // uniform across the block
                    const int bit = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const int ci = base + 64 * w + bit;
                    const double cv = c[ci];
                    int lo = ci - min_distance; if (lo < 0) lo = 0;
                    int hi = ci + min_distance + 1; if (hi > n) hi = n;
                    int bigger = 0;
                    for (int j = lo + threadIdx.x; j < hi; j += NT) bigger |= (c[j] > cv);
                    if (__syncthreads_or(bigger) == 0) {
                        if (threadIdx.x == 0 && total < ES_MAX_PEAKS) peaks[rec * ES_MAX_PEAKS + total] = ci;
                        ++total;
                    }
                }
            }
        }

        if (total == 0) {
            // fallback: five largest correlations, descending; equal values -> higher index first
            const int kmax = n < 5 ? n : 5;
            for (int r = 0; r < kmax; ++r) {
                double bv = 0.0; int bidx = -1;
                for (int i = threadIdx.x; i < n; i += NT) {
                    bool used = false;
                    for (int qd = 0; qd < r; ++qd) used |= (s_taken[qd] == i);
                    if (used) continue;
                    const double ci = c[i];
                    if (bidx < 0 || ci > bv || (ci == bv && i > bidx)) { bv = ci; bidx = i; }
                }
                s_bv[threadIdx.x] = bv; s_bi[threadIdx.x] = bidx;
                __syncthreads();
                for (int sft = NT / 2; sft > 0; sft >>= 1) {
                    if (threadIdx.x < sft) {
                        const double ov = s_bv[threadIdx.x + sft]; const int oi = s_bi[threadIdx.x + sft];
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
            total = kmax;
        } else if (threadIdx.x == 0) {
            npeaks[rec] = total;
        }
        if ((int)threadIdx.x >= total && threadIdx.x < ES_MAX_PEAKS) peaks[rec * ES_MAX_PEAKS + threadIdx.x] = -1;   // unused tail
        if (threadIdx.x == 0) thr_out[rec] = thr;
        (void)s_flag;
        __syncthreads();
    }
}

}  // namespace

int es_launch_bpf(es_ctx* ctx, const void* frames, int dtype, int64_t B, int T, const uint8_t* band,
                  double* y, float* y32, hipStream_t st)
{
    if (B < 262144) {                               // four lanes per record: 4x the waves
        const long long per_block = (long long)BQ_RECS * BQ_WAVES;
        const unsigned blocks = (unsigned)((B + per_block - 1) / per_block);
        if (dtype == ES_DTYPE_I16)
            hipLaunchKernelGGL(es_bpf_quad_kernel<true>, dim3(blocks), dim3(64 * BQ_WAVES), 0, st, frames,
                               (long long)B, T, band, ctx->d_tables, y, y32);
        else
            hipLaunchKernelGGL(es_bpf_quad_kernel<false>, dim3(blocks), dim3(64 * BQ_WAVES), 0, st, frames,
                               (long long)B, T, band, ctx->d_tables, y, y32);
        ES_HIP_CHECK(ctx, hipGetLastError());
        return ES_OK;
    }
    const long long recs_per_block = 64LL * BPF_WAVES;
    const unsigned blocks = (unsigned)((B + recs_per_block - 1) / recs_per_block);
    if (dtype == ES_DTYPE_I16)
        hipLaunchKernelGGL(es_bpf_kernel<true>, dim3(blocks), dim3(64 * BPF_WAVES), 0, st, frames,
                           (long long)B, T, band, ctx->d_tables, y, y32);
    else
        hipLaunchKernelGGL(es_bpf_kernel<false>, dim3(blocks), dim3(64 * BPF_WAVES), 0, st, frames,
                           (long long)B, T, band, ctx->d_tables, y, y32);
    ES_HIP_CHECK(ctx, hipGetLastError());
    return ES_OK;
}


int es_launch_xcorr(es_ctx* ctx, const double* y, int64_t B, int T, const uint8_t* band, double* corr,
                    hipStream_t st)
{
    return es_launch_xcorr_flagged(ctx, y, B, T, band, corr, nullptr, nullptr, st);
}

int es_launch_xcorr_flagged(es_ctx* ctx, const double* y, int64_t B, int T, const uint8_t* band, double* corr,
                            const uint8_t* flags, const int* nflag, hipStream_t st)
{
    const int n_lags = T - (ES_PRE_L - 1);
    const long long nseg = (n_lags + XC_SEG - 1) / XC_SEG;
    long long blocks = (B * nseg + XC_WAVES - 1) / XC_WAVES;
    const long long cap = (long long)ctx->num_cu * (nflag ? 2 : 16);          // a redo pass rarely has anything to do
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(es_xcorr_kernel, dim3((unsigned)blocks), dim3(64 * XC_WAVES), 0, st, y, (long long)B,
                       T, band, ctx->d_tables, corr, flags, nflag);
    ES_HIP_CHECK(ctx, hipGetLastError());
    return ES_OK;
}

int es_launch_pick(es_ctx* ctx, const double* corr, int64_t B, int n_lags, double* thr, int32_t* peaks,
                   int32_t* npeaks, hipStream_t st)
{
    return es_launch_pick_flagged(ctx, corr, B, n_lags, thr, peaks, npeaks, nullptr, nullptr, st);
}

int es_launch_pick_flagged(es_ctx* ctx, const double* corr, int64_t B, int n_lags, double* thr, int32_t* peaks,
                           int32_t* npeaks, const uint8_t* flags, const int* nflag, hipStream_t st)
{
    long long blocks = B;
    const long long cap = (long long)ctx->num_cu * (nflag ? 2 : 16);
    if (blocks > cap) blocks = cap;
    if (n_lags <= PK_LDS_N)
        hipLaunchKernelGGL((es_pick_kernel<true, PK_THREADS>), dim3((unsigned)blocks), dim3(PK_THREADS), 0, st, corr,
                           (long long)B, n_lags, thr, peaks, npeaks, flags, nflag);
    else                                              // long rows (recordings): one block per row, so make it a big one
        hipLaunchKernelGGL((es_pick_kernel<false, 1024>), dim3((unsigned)blocks), dim3(1024), 0, st, corr,
                           (long long)B, n_lags, thr, peaks, npeaks, flags, nflag);
    ES_HIP_CHECK(ctx, hipGetLastError());
    return ES_OK;
}
