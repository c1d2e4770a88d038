// es_sync32.hip -- float32 correlation SCREEN with float64 exact fix-ups.
//
// Why: sync offsets must equal the float64 reference exactly, but a float64 correlation kernel is
// bound by FP64 vector issue (es_sync.hip: ~25 % of the HBM roofline at best).  Almost none of the
// 1153 correlation values of a record ever matter at full precision: only the ones next to a
// decision boundary (the median, the MAD, the threshold, a local maximum).  So:
//
//   es_xcorr32_kernel   y32 (float32, written by the band-pass next to y64) -> corr32 (float32).
//                       4 860 B in, 4 612 B out per record: the traffic SURVEY.md section 8(d)
//                       prices the correlation kernel at.  FP32 FMAs, taps in scalar registers.
//   es_pick_exact_wave_kernel  works on corr32 with a rigorous error bound DELTA (|corr32 - corr64| <=
//                       DELTA), and re-evaluates in float64 -- with exactly the arithmetic of
//                       es_xcorr_kernel / oracle/c/eso_dsp.c -- every value that lies within reach
//                       of a decision: the order statistics' neighbourhoods (order statistics are
//                       1-Lipschitz in the sup norm, so the exact k-th value is the (k - #below)-th
//                       of the band [m32 - 2 DELTA, m32 + 2 DELTA]), the threshold crossers and the
//                       rivals of a local maximum.  thr / peaks / npeaks come out bit-identical to
//                       the all-float64 path; records with too many ambiguous values (constant
//                       signals, exact repeats) are flagged and redone by the float64 kernels.
//
// Error bound: inputs rounded to f32 (2 x 2^-24 relative), 63-term FMA chain (63 x 2^-24 of
// sum|y||tpl| <= e_y by Cauchy-Schwarz, |tpl| = 1), energy and sqrt (~2e-6 relative), approximate
// rcp/sqrt (1e-6): |corr32 - corr64| < 8e-6 for |corr| <= 1.  DELTA = 3e-5.
#include "es_internal.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void wave_fence_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

constexpr int XC_R = 19;                            // must equal es_sync.hip / oracle XC_CHUNK
constexpr int XC_SEG = 64 * XC_R;
constexpr int XC_WAVES = 4;
#ifndef XC_GRID_PER_CU
#define XC_GRID_PER_CU 16                            // blocks per CU the grid is capped at (grid-stride beyond)
#endif
constexpr int XC_MIN_WAVES = 4;                     // waves per SIMD the register allocation must allow
constexpr int XC_T_WINDOW = 2048;                   // BASELINE config 3: 2 048-sample windows (1 986 lags)
constexpr int XC_R_WINDOW = 17;                     // two waves per window: 2 x 64 x 17 = 2 176 >= 1 986 lags (19 would compute 2 432)
constexpr int XC_R_SMALL = 5;                       // lags per lane of the small-batch screen kernel
constexpr double DELTA = 3e-5;

// ------------------------------------------------------------------------------------ fused sync (declarations)
// es_sync_fused_batch runs the correlation kernel below with FUSED = true: a wave owns whole records, the screen row
// stays in LDS, and sync_pick_row() (defined after the picker's helpers) settles threshold and peaks from it -- HBM
// sees 4 bytes per sample in and <= 150 bytes per record out ("4 860 + <= 64 B" per frame).
constexpr int XF_WAVES = 2;                         // waves per block of the fused kernel (a block = 2 records in flight)
constexpr int XF_MIN_WAVES = 3;                     // <= 168 VGPRs: a fused-sync wave fits the slot one list-decoder wave (168) leaves behind
struct FusedArgs {
    const double* y64;                              // [B][T] float64 band-passed records (exact re-evaluations)
    double* thr; int32_t* peaks; int32_t* npeaks; uint8_t* flags;
};
struct PwFixed;
__device__ void sync_pick_row(PwFixed& S, float* c, int n, const double* yr, const double* tpl, long long rec, int lane,
                              const FusedArgs& fo);
__host__ __device__ __forceinline__ size_t xf_lds_per_wave(int ns2, int n_lags);
__host__ __device__ __forceinline__ int xf_head_floats(int ns2);

// ------------------------------------------------------------------------------------ xcorr32
// R = lags per lane.  R = 19 (one wave per frame-sized record) moves the fewest LDS bytes per FMA and is the
// large-batch kernel; R = 5 spreads a record over four waves so that a 1 024-record launch still puts four
// waves on every SIMD (latency-bound regime).  The screen value may differ in the last ulps between the two
// (energy summation order); both satisfy the DELTA bound, and es_pick_exact_wave_kernel is exact for either.
// TC = record length known at compile time (0: use the argument).  With TC = 1 215 (a frame) every bounds
// test of the load / store loops folds away.
// FUSED = true (es_sync_fused_batch): an item is still a (record, segment) pair, but a wave walks ALL segments of its record,
// writes the screen values into the record's LDS row instead of HBM, and after the last segment calls sync_pick_row().
template <int R, int TC, bool FUSED>
__global__ __launch_bounds__(64 * (FUSED ? XF_WAVES : XC_WAVES), FUSED ? XF_MIN_WAVES : XC_MIN_WAVES)
void es_xcorr32_kernel(const float* __restrict__ y, long long B,
        int T_arg, const uint8_t* __restrict__ band, const es_band_tables* __restrict__ tabs,
        float* __restrict__ corr, FusedArgs fo)
{
    const int T = TC ? TC : T_arg;
    __builtin_amdgcn_s_setprio(3);      // a short kernel: when it shares a SIMD with a long-running list-decoder wave it should not queue behind it
    constexpr int SEG = 64 * R;
    constexpr int NS = SEG + ES_PRE_L - 1;
    constexpr int WAVES = FUSED ? XF_WAVES : XC_WAVES;
    __shared__ float s_buf[FUSED ? 1 : XC_WAVES][FUSED ? 4 : NS + 2];
    extern __shared__ __attribute__((aligned(16))) unsigned char xf_smem[];      // FUSED: per wave [samples, later the picker's PwFixed | screen row]
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform: item, record, band are scalar
    const int n_lags = T - (ES_PRE_L - 1);
    float* s = FUSED ? reinterpret_cast<float*>(xf_smem + wv * xf_lds_per_wave(NS + 2, n_lags)) : s_buf[wv];
    float* const crow = s + xf_head_floats(NS + 2);                           // FUSED: the record's screen row
    const int nseg = (n_lags + SEG - 1) / SEG;
    const long long n_items = B * nseg;
    // non-fused: consecutive items go to consecutive waves.  Fused: a wave takes the nseg items of one record, then
    // jumps to the record gridDim.x * WAVES further on.
    const long long stride = FUSED ? (long long)gridDim.x * WAVES * nseg - (nseg - 1) : (long long)gridDim.x * WAVES;
    // item -> (record, segment) with 32-bit arithmetic (the launcher keeps n_items < 2^31); nseg is 1
    // for frame-sized records, so the division disappears on the hot path
    auto rec_of = [&](unsigned it) { return nseg == 1 ? it : it / (unsigned)nseg; };
    constexpr int NST = (NS + 63) / 64;
    float stage[NST];                               // next record's samples, in flight while this one is computed
    int bi_next = 0;
    auto prefetch = [&](unsigned it) {
        const long long r = rec_of(it);
        const int l0 = (nseg == 1) ? 0 : (int)(it - (unsigned)r * (unsigned)nseg) * SEG;
        const float* src = y + r * T + l0;
        const int ns = (T - l0 < NS) ? T - l0 : NS;
        #pragma unroll
        for (int u = 0; u < NST; ++u) { const int i = lane + 64 * u; stage[u] = (i < ns) ? src[i] : 0.0f; }
        bi_next = (int)band[r];
    };
    static_assert(R % 2 == 1, "R - 1 must be even (packed core energy)");
    if (lane < 2) s[NS + lane] = 0.0f;               // the zero-tap partner of the last lag reads one past the samples
    unsigned item = FUSED ? (unsigned)(blockIdx.x * WAVES + wv) * (unsigned)nseg : (unsigned)(blockIdx.x * WAVES + wv);
    if (item >= (unsigned)n_items) return;
    // the item after `it`: fused waves finish their record first
    auto next_item = [&](unsigned it) -> unsigned {
        if (!FUSED) return it + (unsigned)stride;
        return ((it + 1u) % (unsigned)nseg != 0u) ? it + 1u : it + (unsigned)stride;
    };
    prefetch(item);
    // The body is instantiated twice, once peeled in front of the loop: inside the loop the 19 stores of
    // the previous record are then ALWAYS younger than the loads being waited for, so the compiler waits
    // with a counted vmcnt and a wave never stalls on its own output stores.
    auto body = [&](unsigned item) __attribute__((always_inline)) {
        const long long rec = rec_of(item);
        const int lag0 = (nseg == 1) ? 0 : (int)(item - (unsigned)rec * (unsigned)nseg) * SEG;
        const int bi = __builtin_amdgcn_readfirstlane(bi_next);
        #pragma unroll
        for (int u = 0; u < NST; ++u) { const int i = lane + 64 * u; if (i < NS) s[i] = stage[u]; }
        if (next_item(item) < (unsigned)n_items) prefetch(next_item(item));
        const float* tpg = tabs->tpl32[bi];
        f32x2 tp2[32];                                       // tap pairs (2i, 2i+1), tap 63 = 0: SGPR pairs for the record
        #pragma unroll
        for (int i = 0; i < 32; ++i) { tp2[i].x = tpg[2 * i]; tp2[i].y = tpg[2 * i + 1]; }
        wave_fence_lds();

        // Packed FP32 (v_pk_fma_f32: two FMAs per lane per issue slot).  Each lag keeps two partial sums,
        // .x over the even taps and .y over the odd taps; a pair of consecutive samples (one ds_read2_b32)
        // meets the aligned tap pair (2i, 2i+1).  Even lags use the sample pairs that start at an even
        // offset, odd lags the ones that start at an odd offset, so the tap pairs are the same 32 aligned
        // SGPR pairs for every lag.  The 64th "tap" is zero; the sample it meets is the next lane's (or
        // the zero pad), finite unless the record already holds a NaN/Inf, which flags the record anyway.
        const float* w = s + lane * R;
        f32x2 acc[R];
        #pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = f32x2{0.0f, 0.0f};
        f32x2 core2 = f32x2{0.0f, 0.0f};                     // squares of samples R-1 .. 61 (R-1 is even)
        #define XC32_PK(par, jj, v2)                                                            \
            _Pragma("unroll") for (int r = (par); r < R; r += 2) {                              \
                const int k = 2 * (jj) + (par) - r;                                             \
                if (k >= 0 && k <= ES_PRE_L - 1) acc[r] = __builtin_elementwise_fma((v2), tp2[k / 2], acc[r]); \
            }
        #pragma unroll
        for (int j = 0; j < (R - 1) / 2; ++j) {
            const f32x2 e = f32x2{w[2 * j], w[2 * j + 1]}; XC32_PK(0, j, e)
            const f32x2 o = f32x2{w[2 * j + 1], w[2 * j + 2]}; XC32_PK(1, j, o)
        }
        #pragma unroll
        for (int j = (R - 1) / 2; j <= (ES_PRE_L - 2) / 2; ++j) {
            const f32x2 e = f32x2{w[2 * j], w[2 * j + 1]}; core2 = __builtin_elementwise_fma(e, e, core2); XC32_PK(0, j, e)
            const f32x2 o = f32x2{w[2 * j + 1], w[2 * j + 2]}; XC32_PK(1, j, o)
        }
        #pragma unroll
        for (int j = (ES_PRE_L - 2) / 2 + 1; j <= (R - 1 + ES_PRE_L - 1) / 2; ++j) {
            const f32x2 e = f32x2{w[2 * j], w[2 * j + 1]}; XC32_PK(0, j, e)
            const f32x2 o = f32x2{w[2 * j + 1], w[2 * j + 2]}; XC32_PK(1, j, o)
        }
        #undef XC32_PK
        // window energies: en[r] = (head_r + core) + tail_r, all terms >= 0 (well conditioned).  The edge
        // samples are re-read from LDS here so that no energy state is live during the FMA loop.
        const float v62 = w[ES_PRE_L - 1];
        const float core = __builtin_fmaf(v62, v62, core2.x + core2.y);
        float en[R];
        en[R - 1] = 0.0f;
        #pragma unroll
        for (int r = R - 2; r >= 0; --r) { const float v = w[r]; en[r] = __builtin_fmaf(v, v, en[r + 1]); }
        #pragma unroll
        for (int r = 0; r < R; ++r) en[r] = en[r] + core;
        float tail_run = 0.0f;
        #pragma unroll
        for (int r = 1; r < R; ++r) { const float v = w[ES_PRE_L - 1 + r]; tail_run = __builtin_fmaf(v, v, tail_run); en[r] = en[r] + tail_run; }
        float emin = en[0], emax = en[0];
        #pragma unroll
        for (int r = 1; r < R; ++r) { emin = __builtin_fminf(emin, en[r]); emax = __builtin_fmaxf(emax, en[r]); }
        wave_fence_lds();
        // sqrt(en) >= 1e-6 makes the reference's "+ 1e-12" a relative 1e-6 effect (inside DELTA): one rsq
        // instead of sqrt + rcp.  Near-silent windows, and energies outside float32 range (|y| beyond
        // ~1e18, which would make the quotient silently wrong), take the careful form for the whole wave;
        // an overflowed energy becomes NaN, which sends the record to the float64 kernels.
        float* const dst = FUSED ? crow + lag0 : s;                           // FUSED: straight into the record's row
        const int nl = (n_lags - lag0 < SEG) ? n_lags - lag0 : SEG;
        if (__builtin_amdgcn_ballot_w64(!(emin >= 1.0e-12f && emax < 3.0e38f)) == 0) {
            #pragma unroll
            for (int r = 0; r < R; ++r) if (!FUSED || lane * R + r < nl) dst[lane * R + r] = (acc[r].x + acc[r].y) * __builtin_amdgcn_rsqf(en[r]);
        } else {
            #pragma unroll
            for (int r = 0; r < R; ++r) {
                const float q32 = (acc[r].x + acc[r].y) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(en[r]) + 1e-12f);
                if (!FUSED || lane * R + r < nl) dst[lane * R + r] = (en[r] < 3.0e38f) ? q32 : __builtin_nanf("");
            }
        }
        wave_fence_lds();
        if constexpr (FUSED) {
            if (lag0 + SEG >= n_lags) {                                       // last segment: the row is complete
                PwFixed& S = *reinterpret_cast<PwFixed*>(s);             // the samples are dead by now: their LDS serves the picker
                sync_pick_row(S, crow, n_lags, fo.y64 + rec * T, tabs->tpl[bi], rec, lane, fo);
            }
        } else {
            float* cr = corr + rec * n_lags + lag0;
            // straight-line stores (no loop): the compiler then knows how many are outstanding and waits for the
            // NEXT record's loads with a counted vmcnt instead of draining these stores first
            #pragma unroll
            for (int u = 0; u < R; ++u) { const int i = lane + 64 * u; if (i < nl) cr[i] = s[i]; }
            wave_fence_lds();
        }
    };
    if constexpr (FUSED) {
        for (; item < (unsigned)n_items; item = next_item(item)) body(item);
    } else {
        body(item);
        for (item += (unsigned)stride; item < (unsigned)n_items; item += (unsigned)stride) body(item);
    }
}

// ------------------------------------------------------------------------------------ exact value
// corr64 of one lag, bit-identical to es_xcorr_kernel / eso_ncc: FMA chain over ascending taps;
// energy = (head + core) + tail over the 19-lag chunk the lag belongs to.
__device__ double corr64_at(const double* __restrict__ yr, int i, const double* __restrict__ tpl)
{
    const int c = i - i % XC_R;
    // (unrolled by nine: the nine loads of a group are issued together -- one round trip per group instead of one per sample; the value's
    //  ~145 samples were ~145 dependent L1/L2 round trips, 45 % of the fused kernel's time by ablation)
    double num = 0.0;
    #pragma unroll 9
    for (int k = 0; k < ES_PRE_L; ++k) num = __builtin_fma(yr[i + k], tpl[k], num);
    double core = 0.0;
    #pragma unroll 9
    for (int m = 0; m < ES_PRE_L - XC_R + 1; ++m) core = core + yr[c + XC_R - 1 + m] * yr[c + XC_R - 1 + m];
    // (head and tail: a fixed number of unconditional loads from addresses clamped into the range the value uses, the additions predicated --
    //  with the bound inside the loop condition every sample was a round trip of its own)
    double head = 0.0;
    #pragma unroll 9
    for (int u = 0; u < XC_R - 1; ++u) { const int j = c + XC_R - 2 - u; const double v = yr[j]; if (j >= i) head = head + v * v; }
    double tail = 0.0;
    #pragma unroll 9
    for (int u = 0; u < XC_R - 1; ++u) {
        const int j = c + ES_PRE_L + u, last = i + ES_PRE_L - 1;
        const double v = yr[j <= last ? j : last];
        if (j <= last) tail = tail + v * v;
    }
    const double en = (head + core) + tail;
    return num / (__builtin_sqrt(en) + 1e-12);
}

// The same value from a window of the record staged in LDS: w[x] = yr[c + x], x = 0 .. 80, c = i - i % XC_R, r = i - c.  Same operands in
// the same order as corr64_at: bit-identical.
// (w is an LDS pointer, not a generic one: through a generic pointer the compiler merges neighbouring doubles into 16-byte flat loads at
// 8-byte alignment -- fine for global memory, a memory violation when the address resolves to LDS on this target: that is how the first
// version of this routine aborted the queue.  With the address space known it emits 8-byte-aligned ds_read2_b64.)
typedef __attribute__((address_space(3))) double xl_lds_double;
__device__ __forceinline__ double corr64_window(const xl_lds_double* w, int r, const double* __restrict__ tpl)
{
    double num = 0.0;
    #pragma unroll 9
    for (int k = 0; k < ES_PRE_L; ++k) num = __builtin_fma(w[r + k], tpl[k], num);
    double core = 0.0;
    #pragma unroll 9
    for (int m = 0; m < ES_PRE_L - XC_R + 1; ++m) core = core + w[XC_R - 1 + m] * w[XC_R - 1 + m];
    double head = 0.0;
    #pragma unroll 9
    for (int u = 0; u < XC_R - 1; ++u) { const int j = XC_R - 2 - u; const double v = w[j]; if (j >= r) head = head + v * v; }
    double tail = 0.0;
    #pragma unroll 9
    for (int u = 0; u < XC_R - 1; ++u) {
        const int j = ES_PRE_L + u, last = r + ES_PRE_L - 1;
        const double v = w[j <= last ? j : last];
        if (j <= last) tail = tail + v * v;
    }
    const double en = (head + core) + tail;
    return num / (__builtin_sqrt(en) + 1e-12);
}

// Exact correlations of `cnt` lags, lag_of(w) wave-uniform, w = 0 .. cnt-1, handed to out(w, value) in the lane w % XL_G.  The ~81 float64
// samples a value reads (its own 63 and the rest of its 19-lag energy chunk) are fetched by the WHOLE wave -- two coalesced loads per value,
// XL_G values' loads in flight together, one trip to memory per group -- into 4 KB of LDS (`stage`: the picker's histogram copies, dead
// whenever this runs), and the value is then formed from LDS by one lane.  Before, each lane walked its value's samples through global
// loads of its own: ~16 dependent trips to a row that no cache holds (0.4 ms of the fused kernel's 1.0 ms per 65 536 windows, by ablation).
constexpr int XL_G = 6, XL_W = ES_PRE_L + XC_R - 1, XL_STRIDE = 84;    // 81 samples per window; 6 x 84 doubles = 4 032 B
template <typename FL, typename FO>
__device__ __forceinline__ void corr64_list(double* stage, int cnt, const double* __restrict__ yr, int n, const double* __restrict__ tpl, int lane, FL lag_of, FO out)
{
    static_assert(XL_G * XL_STRIDE * 8 <= 4096 && XL_W <= XL_STRIDE && XL_W - 64 <= 64, "stage area");
    const int last = n + ES_PRE_L - 2;                                  // last sample of the record (n lags of ES_PRE_L taps)
    xl_lds_double* const st = (xl_lds_double*)stage;                    // (`stage` is LDS: see corr64_window)
    for (int g0 = 0; g0 < cnt; g0 += XL_G) {
        const int ng = cnt - g0 < XL_G ? cnt - g0 : XL_G;
        double v0[XL_G], v1[XL_G];
        int rr = 0;
        #pragma unroll
        for (int w = 0; w < XL_G; ++w) {
            const int i = lag_of(g0 + (w < ng ? w : 0));
            const int c0 = i - i % XC_R;
            if (w == lane) rr = i - c0;
            const int a0 = c0 + lane, a1 = c0 + 64 + (lane < XL_W - 64 ? lane : 0);
            v0[w] = yr[a0 <= last ? a0 : last];
            v1[w] = yr[a1 <= last ? a1 : last];
        }
        #pragma unroll
        for (int w = 0; w < XL_G; ++w) {
            st[w * XL_STRIDE + lane] = v0[w];
            if (lane < XL_W - 64) st[w * XL_STRIDE + 64 + lane] = v1[w];
        }
        wave_fence_lds();
        if (lane < ng) out(g0 + lane, corr64_window(st + lane * XL_STRIDE, rr, tpl));
        wave_fence_lds();
    }
}

// ------------------------------------------------------------------------------------ pick (exact)
constexpr int PX_MAXN = 4096;

__device__ __forceinline__ uint32_t f32_key(float x)
{
    uint32_t b; __builtin_memcpy(&b, &x, 4);
    return (b >> 31) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_f32(uint32_t k)
{
    const uint32_t b = (k >> 31) ? (k & 0x7fffffffu) : ~k;
    float x; __builtin_memcpy(&x, &b, 4); return x;
}

// One WAVE per record (a block-per-record version spent its time in ~100 block barriers per record): the float32
// row sits in LDS, order statistics are 8-bit-digit radix selects on per-wave LDS histograms (ds_add, wave scan, no
// barrier), band / candidate lists are built with ballots, and all the float64 re-evaluations of a step run in
// parallel, one per lane.
//   Order statistics: |screen - exact| <= d, and order statistics are 1-Lipschitz in the sup norm, so the exact k-th
//   value is the (k - #below)-th of the exact values of the band [m32 - 2d, m32 + 2d] around the float32 order statistic.
// Records that would need more than PW_CAP exact values (or more than 64 rivals of one candidate) are flagged
// (reason codes 1..5) and redone by the float64 kernels inside the same es_pick_exact_batch call.
constexpr int PW_WAVES = 4;
constexpr int PW_CAP = 192;

constexpr int PW_HCOPIES = 4;                       // private copies of the histogram (lane & 3): the leading byte of a
                                                    // correlation value takes a handful of values, and 64 lanes adding to
                                                    // one LDS word serialise
struct PwFixed {
    uint32_t hist[PW_HCOPIES][256];
    int      list[PW_CAP];
    double   val[PW_CAP];
};

__device__ __forceinline__ int lanes_below(unsigned long long m)
{
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// k-th smallest 32-bit key of key(i), i in [0,n), by one wave: 4 passes of 8 bits
template <typename F>
__device__ uint32_t pw_select(PwFixed& S, int n, int k, int lane, F key)
{
    uint32_t prefix = 0;
    int kk = k;
    for (int shift = 24; shift >= 0; shift -= 8) {
        #pragma unroll
        for (int b = 0; b < 4 * PW_HCOPIES; ++b) (&S.hist[0][0])[lane + 64 * b] = 0;
        wave_fence_lds();
        const uint32_t himask = (shift == 24) ? 0u : (~0u << (shift + 8));
        uint32_t* const myh = S.hist[lane & (PW_HCOPIES - 1)];
        for (int i = lane; i < n; i += 64) {
            const uint32_t kx = key(i);
            if ((kx & himask) == prefix) atomicAdd(&myh[(kx >> shift) & 255u], 1u);
        }
        wave_fence_lds();
        uint32_t h[4];
        #pragma unroll
        for (int b = 0; b < 4; ++b) {                                    // lane owns four consecutive bins
            uint32_t t = 0;
            #pragma unroll
            for (int cpy = 0; cpy < PW_HCOPIES; ++cpy) t += S.hist[cpy][4 * lane + b];
            h[b] = t;
        }
        const uint32_t s4 = h[0] + h[1] + h[2] + h[3];
        const uint32_t incl = es_wave_incl_scan_u32(s4);
        const uint32_t excl = incl - s4;
        const bool hit = ((int)excl <= kk) && (kk < (int)incl);
        int bin = 0, nk = 0;
        if (hit) {
            uint32_t c = excl;
            #pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (kk >= (int)c && kk < (int)(c + h[b])) { bin = 4 * lane + b; nk = kk - (int)c; }
                c += h[b];
            }
        }
        const unsigned long long m = __ballot(hit);
        const int src = __ffsll((long long)m) - 1;
        bin = es_wave_read_lane(bin, src); kk = es_wave_read_lane(nk, src);
        prefix |= (uint32_t)bin << shift;
        wave_fence_lds();
    }
    return prefix;
}

// Exact order statistics k_lo <= k_hi of exact(i) from a screen with |screen - exact| <= d.  Results in r0, r1;
// false = the band does not fit (record must be flagged).
template <typename FS, typename FE>
__device__ bool pw_exact_stats(PwFixed& S, int n, int k_lo, int k_hi, double d, int lane, FS screen, FE exact,
                               double& r0, double& r1)
{
    auto key = [&](int i) { return f32_key((float)screen(i)); };
    const uint32_t key_lo = pw_select(S, n, k_lo, lane, key);
    const float m_lo = key_f32(key_lo);
    float m_hi = m_lo;
    if (k_hi != k_lo) {
        // k_hi = k_lo + 1: the next order statistic is m_lo itself if enough keys are <= m_lo, else the smallest
        // key above it -- one pass of counting instead of a second radix select
        int le = 0; uint32_t nxt = 0xffffffffu;
        for (int i = lane; i < n; i += 64) {
            const uint32_t kx = key(i);
            le += kx <= key_lo;
            if (kx > key_lo && kx < nxt) nxt = kx;
        }
        #pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            le += __shfl_xor(le, o);
            const uint32_t on = (uint32_t)__shfl_xor((int)nxt, o);
            nxt = on < nxt ? on : nxt;
        }
        m_hi = (le > k_hi) ? m_lo : key_f32(nxt);
    }
    const double lo = (double)m_lo - 2.0 * d - 1e-6 * __builtin_fabs((double)m_lo);
    const double hi = (double)m_hi + 2.0 * d + 1e-6 * __builtin_fabs((double)m_hi);
    int below = 0, nb = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const bool in = i < n;
        const double v = in ? screen(i) : 0.0;
        const bool isb = in && v < lo;
        const bool inb = in && !(v < lo) && v <= hi;
        below += __popcll(__ballot(isb));
        const unsigned long long mb = __ballot(inb);
        const int pos = nb + lanes_below(mb);
        if (inb && pos < PW_CAP) S.list[pos] = i;
        nb += __popcll(mb);
    }
    if (nb > PW_CAP) return false;
    wave_fence_lds();
    for (int idx = lane; idx < nb; idx += 64) S.val[idx] = exact(S.list[idx]);
    wave_fence_lds();
    const int r_lo = k_lo - below, r_hi = k_hi - below;
    double c0 = 0.0, c1 = 0.0; bool h0 = false, h1 = false;
    for (int idx = lane; idx < nb; idx += 64) {
        const double v = S.val[idx];
        int less = 0, eq = 0;
        for (int j = 0; j < nb; ++j) { const double vj = S.val[j]; less += vj < v; eq += vj == v; }
        if (less <= r_lo && r_lo < less + eq) { c0 = v; h0 = true; }
        if (less <= r_hi && r_hi < less + eq) { c1 = v; h1 = true; }
    }
    const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1);
    r0 = __shfl(c0, m0 ? __ffsll((long long)m0) - 1 : 0);
    r1 = __shfl(c1, m1 ? __ffsll((long long)m1) - 1 : 0);
    wave_fence_lds();
    return true;
}

// LDS of one fused-sync wave: a head region that holds the staged samples while the row is computed and the picker's
// PwFixed afterwards (never live together), then the record's screen row.
__host__ __device__ __forceinline__ int xf_head_floats(int ns2)
{
    const int a = (ns2 + 3) & ~3, b = (int)((sizeof(PwFixed) + 15) & ~(size_t)15) / 4;
    return a > b ? a : b;
}
__host__ __device__ __forceinline__ size_t xf_lds_per_wave(int ns2, int n_lags)
{
    return (size_t)xf_head_floats(ns2) * 4 + (((size_t)n_lags * 4 + 15) & ~(size_t)15);
}

// ------------------------------------------------------------------------------------ fused sync: the rare exact row
// A record the screen cannot settle (hundreds of exactly equal correlations: digital silence, constants, exact repeats;
// or a screen that overflowed float32) is settled HERE, by the same wave, entirely from float64 re-evaluations -- no
// workspace and no second launch: every pass recomputes corr64_at(i) (bit-identical to es_xcorr_kernel) for the lags it
// looks at.  Same rules and tie-breaks as es_pick_kernel: exact order statistics by 8-bit radix select on the monotone
// 64-bit image of the doubles, NMS window +-607, fallback = five largest, equal values -> higher index first.  Slow
// (16+ passes of ~200 float64 operations per lag) and meant to be: such records are degenerate.
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

template <typename F>
__device__ double pw_select64(PwFixed& S, int n, int k, int lane, F val)
{
    uint64_t prefix = 0;
    int kk = k;
    for (int shift = 56; shift >= 0; shift -= 8) {
        #pragma unroll
        for (int b = 0; b < 4 * PW_HCOPIES; ++b) (&S.hist[0][0])[lane + 64 * b] = 0;
        wave_fence_lds();
        const uint64_t himask = (shift == 56) ? 0ULL : (~0ULL << (shift + 8));
        uint32_t* const myh = S.hist[lane & (PW_HCOPIES - 1)];
        for (int i = lane; i < n; i += 64) {
            const uint64_t kx = f64_key(val(i));
            if ((kx & himask) == prefix) atomicAdd(&myh[(uint32_t)(kx >> shift) & 255u], 1u);
        }
        wave_fence_lds();
        uint32_t h[4];
        #pragma unroll
        for (int b = 0; b < 4; ++b) {
            uint32_t t = 0;
            #pragma unroll
            for (int cpy = 0; cpy < PW_HCOPIES; ++cpy) t += S.hist[cpy][4 * lane + b];
            h[b] = t;
        }
        const uint32_t s4 = h[0] + h[1] + h[2] + h[3];
        const uint32_t incl = es_wave_incl_scan_u32(s4);
        const uint32_t excl = incl - s4;
        const bool hit = ((int)excl <= kk) && (kk < (int)incl);
        int bin = 0, nk = 0;
        if (hit) {
            uint32_t c = excl;
            #pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (kk >= (int)c && kk < (int)(c + h[b])) { bin = 4 * lane + b; nk = kk - (int)c; }
                c += h[b];
            }
        }
        const unsigned long long m = __ballot(hit);
        const int src = __ffsll((long long)m) - 1;
        bin = es_wave_read_lane(bin, src); kk = es_wave_read_lane(nk, src);
        prefix |= (uint64_t)(uint32_t)bin << shift;
        wave_fence_lds();
    }
    return key_f64(prefix);
}

__device__ __noinline__ void sync_exact_row(PwFixed& S, int n, const double* yr, const double* tpl, long long rec, int lane,
                                            const FusedArgs& fo)
{
    const int min_distance = ES_FRAME_LEN / 2;
    auto ex = [&](int i) { return corr64_at(yr, i, tpl); };
    double med;
    if (n & 1) med = pw_select64(S, n, n / 2, lane, ex);
    else { const double lo = pw_select64(S, n, n / 2 - 1, lane, ex), hi = pw_select64(S, n, n / 2, lane, ex); med = (lo + hi) / 2.0; }
    auto dev = [&](int i) { return __builtin_fabs(corr64_at(yr, i, tpl) - med); };
    double mad;
    if (n & 1) mad = pw_select64(S, n, n / 2, lane, dev);
    else { const double lo = pw_select64(S, n, n / 2 - 1, lane, dev), hi = pw_select64(S, n, n / 2, lane, dev); mad = (lo + hi) / 2.0; }
    mad = mad + 1e-12;
    double thr = med + 4.5 * 1.4826 * mad;
    if (0.95 < thr) thr = 0.95;

    int total = 0;
    for (int base = 0; base < n; base += 64) {                          // ascending lags; a candidate is checked by the whole wave
        const int i = base + lane;
        const double cv_mine = (i < n) ? ex(i) : 0.0;
        unsigned long long m = __ballot((i < n) && !(cv_mine < thr));
        while (m) {
            const int bit = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int ci = base + bit;
            const double cv = __shfl(cv_mine, bit);
            int lo = ci - min_distance; if (lo < 0) lo = 0;
            int hi = ci + min_distance + 1; if (hi > n) hi = n;
            bool bigger = false;
            for (int j0 = lo; j0 < hi && !bigger; j0 += 64) {
                const int j = j0 + lane;
                if (__ballot((j < hi) && ex(j) > cv)) bigger = true;
            }
            if (bigger) continue;
            if (lane == 0 && total < ES_MAX_PEAKS) fo.peaks[rec * ES_MAX_PEAKS + total] = ci;
            ++total;
        }
    }
    if (total == 0) {
        const int kmax = n < 5 ? n : 5;                                 // five largest, descending; equal values -> higher index first
        int taken[5] = {-1, -1, -1, -1, -1};
        for (int r = 0; r < kmax; ++r) {
            double bv = 0.0; int bi = -1;
            for (int i = lane; i < n; i += 64) {
                bool used = false;
                #pragma unroll
                for (int qd = 0; qd < 5; ++qd) used |= (qd < r) && (taken[qd] == i);
                if (used) continue;
                const double v = ex(i);
                if (bi < 0 || v > bv || (v == bv && i > bi)) { bv = v; bi = i; }
            }
            #pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                const double ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
                if (oi >= 0 && (bi < 0 || ov > bv || (ov == bv && oi > bi))) { bv = ov; bi = oi; }
            }
            #pragma unroll
            for (int qd = 0; qd < 5; ++qd) if (qd == r) taken[qd] = bi;
            if (lane == 0) fo.peaks[rec * ES_MAX_PEAKS + r] = bi;
        }
        if (lane == 0) fo.npeaks[rec] = kmax | (1 << 30);
        total = kmax;
    } else if (lane == 0) {
        fo.npeaks[rec] = total;
    }
    if (lane >= total && lane < ES_MAX_PEAKS) fo.peaks[rec * ES_MAX_PEAKS + lane] = -1;
    if (lane == 0) fo.thr[rec] = thr;
}

// ------------------------------------------------------------------------------------ fused sync: threshold + peaks of one row
// One wave, the float32 screen row c[0..n) in LDS.  Same decisions as es_pick_exact_wave_kernel (hence as the float64
// path), reached with far fewer passes over the row in the common case:
//   * ONE linear 256-bin histogram of the row (bins of 1/128 over [-1, 1): correlation values spread over them, so the
//     LDS atomics hardly collide -- the radix select's first digit, sign + exponent, takes a handful of values).
//   * Saturation test.  thr = min(med + 4.5 * 1.4826 * MAD, 0.95).  The histogram brackets the median,
//     med >= lo(bl) - d (bl = bin of the lower middle order statistic, d = DELTA), and counts values certainly at least
//     r_j = j/128 - 2d away from ANY median in that bracket: the bins below bl - j and above bh + j.  If at least
//     n - k_lo values are that far out, the k_lo-th absolute deviation -- hence the MAD -- is >= r_j.  When
//     lo(bl) - d + 6.6717 * r_j >= 0.95 + 1e-6 the minimum is 0.95 whatever the exact median and MAD are, and neither
//     is computed.  (Band-passed noise, and clean frames too, have MAD ~0.2: the threshold saturates on every workload of
//     BASELINE.json; records where it does not take the exact order statistics of es_pick_exact_wave_kernel.)
//   * Threshold crossers are looked for only if the histogram has a value at or above thr - d.
//   * Fallback (no peak): the five largest exact correlations lie among the screen values >= lo(b*) - 2d, b* = the
//     highest bin with at least five values at or above it -- read off the histogram, no selection pass.
// Flags (record redone by the float64 kernels): 1 non-finite screen, 2/3 order-statistic band too wide, 4 too many rivals
// of a candidate within DELTA, 5 fallback list too long.
__device__ void sync_pick_row(PwFixed& S, float* c, int n, const double* yr, const double* tpl, long long rec, int lane,
                              const FusedArgs& fo)
{
    const int min_distance = ES_FRAME_LEN / 2;
    // a record the screen cannot settle: reason code to flags (informational), then the exact row by this same wave
    auto flag_out = [&](int code) {
        if (lane == 0) fo.flags[rec] = (uint8_t)code;
        wave_fence_lds();
        sync_exact_row(S, n, yr, tpl, rec, lane, fo);
    };
    auto exact_corr = [&](int i) { return corr64_at(yr, i, tpl); };
    const int k_hi = n / 2, k_lo = (n & 1) ? n / 2 : n / 2 - 1;
    constexpr double BINW = 1.0 / 128.0;
    constexpr double MARG = 2.5 * DELTA;                                // screen error both ways + float32 binning slack
    uint32_t* const cum = reinterpret_cast<uint32_t*>(S.val);           // cum[b] = number of values in bins < b, b = 0..256

    // ---- histogram
    #pragma unroll
    for (int b = 0; b < 4 * PW_HCOPIES; ++b) (&S.hist[0][0])[lane + 64 * b] = 0;
    wave_fence_lds();
    int bad = 0;
    {
        uint32_t* const myh = S.hist[lane & (PW_HCOPIES - 1)];
        for (int i0 = lane; i0 < n; i0 += 64 * 8) {                     // eight row values read before the first histogram update (LDS reads
            float v8[8];                                                //  and LDS atomics: in a plain loop each read waited behind the last update)
            #pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + 64 * u; v8[u] = c[i < n ? i : n - 1]; }
            #pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (i0 + 64 * u < n) {
                    const float v = v8[u];
                    bad |= !(__builtin_fabsf(v) < 1e30f);               // inf / nan / absurd: screen unusable
                    int b = (int)((v + 1.0f) * 128.0f);
                    b = b < 0 ? 0 : (b > 255 ? 255 : b);
                    atomicAdd(&myh[b], 1u);
                }
            }
        }
    }
    wave_fence_lds();
    if (__ballot(bad)) { flag_out(1); return; }
    uint32_t h[4];
    #pragma unroll
    for (int b = 0; b < 4; ++b) {                                       // lane owns four consecutive bins
        uint32_t t = 0;
        #pragma unroll
        for (int cpy = 0; cpy < PW_HCOPIES; ++cpy) t += S.hist[cpy][4 * lane + b];
        h[b] = t;
    }
    const uint32_t s4 = h[0] + h[1] + h[2] + h[3];
    const uint32_t incl = es_wave_incl_scan_u32(s4);
    {
        uint32_t e = incl - s4;
        #pragma unroll
        for (int b = 0; b < 4; ++b) { cum[4 * lane + b] = e; e += h[b]; }
        if (lane == 63) cum[256] = e;
    }
    wave_fence_lds();
    auto bin_of_rank = [&](int k) {                                     // the bin holding the k-th smallest screen value
        int mine = -1;
        uint32_t e = incl - s4;
        #pragma unroll
        for (int b = 0; b < 4; ++b) { if ((int)e <= k && k < (int)(e + h[b])) mine = 4 * lane + b; e += h[b]; }
        const unsigned long long m = __ballot(mine >= 0);
        return es_wave_read_lane(mine, __ffsll((long long)m) - 1);
    };
    const int bl = bin_of_rank(k_lo), bh = (k_hi == k_lo) ? bl : bin_of_rank(k_hi);

    // ---- threshold
    double thr = 0.0;
    bool have_cum = true;
    {
        const int j = lane;                                             // r_j = j/128 - MARG
        const uint32_t below = (bl - j > 0) ? cum[bl - j] : 0u;
        const uint32_t above = (uint32_t)n - cum[(bh + j + 1 < 256) ? bh + j + 1 : 256];
        const bool okj = (int)(below + above) >= n - k_lo;
        const unsigned long long mk = __ballot(okj);
        const int jmax = (~mk) ? (int)__builtin_ctzll(~mk) - 1 : 63;    // count_out is non-increasing in j: leading run of ok's
        const double mL = (-1.0 + bl * BINW) - MARG;
        const bool sat = bl >= 1 && bh <= 254 && jmax >= 0 && mL + (4.5 * 1.4826) * (jmax * BINW - MARG) >= 0.95 + 1e-6;
        if (sat) {
            thr = 0.95;
        } else {
            double r0, r1;
            if (!pw_exact_stats(S, n, k_lo, k_hi, DELTA, lane, [&](int i) { return (double)c[i]; }, exact_corr, r0, r1)) { flag_out(2); return; }
            const double med = (n & 1) ? r0 : (r0 + r1) / 2.0;
            if (!pw_exact_stats(S, n, k_lo, k_hi, DELTA, lane, [&](int i) { return __builtin_fabs((double)c[i] - med); },
                                [&](int i) { return __builtin_fabs(corr64_at(yr, i, tpl) - med); }, r0, r1)) { flag_out(3); return; }
            const double mad = ((n & 1) ? r0 : (r0 + r1) / 2.0) + 1e-12;
            thr = med + 4.5 * 1.4826 * mad;
            if (0.95 < thr) thr = 0.95;
            have_cum = false;                                           // the selects reused the histogram and cum[]
        }
    }

    // ---- threshold crossers in ascending order; each one is settled exactly (as in es_pick_exact_wave_kernel)
    int total = 0;
    bool overflow = false;
    bool any_cross = true;
    if (have_cum) {                                                     // is there any value at or above thr - DELTA at all?
        int bt = (int)(((thr - DELTA) + 1.0) * 128.0) - 1;              // one bin of slack for the float32 binning
        bt = bt < 0 ? 0 : (bt > 255 ? 255 : bt);
        any_cross = cum[256] - cum[bt] > 0;
    }
    double* const stage = reinterpret_cast<double*>(&S.hist[0][0]);    // the histogram copies are dead from here on (a later select rebuilds them): corr64_list's windows
    auto lane_f64 = [](double v, int src) {                             // v of lane `src` (wave-uniform)
        const uint64_t u = __builtin_bit_cast(uint64_t, v);
        const uint32_t lo = (uint32_t)es_wave_read_lane((int)(uint32_t)u, src), hi = (uint32_t)es_wave_read_lane((int)(uint32_t)(u >> 32), src);
        return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
    };
    for (int base = 0; any_cross && base < n && !overflow; base += 64) {
        const int i = base + lane;
        const bool cand = (i < n) && ((double)c[i] >= thr - DELTA);
        const unsigned long long m = __ballot(cand);
        if (!m) continue;
        const int nc = __popcll(m);
        auto nth = [&](int w) { unsigned long long t = m; for (int q = 0; q < w; ++q) t &= t - 1; return base + (int)__ffsll((long long)t) - 1; };   // lag of the chunk's w-th candidate
        // one candidate: exact value cv at lag ci -> a peak unless a bigger value sits within +-min_distance
        auto settle = [&](int ci, double cv) {
            if (cv < thr) return;
            int lo = ci - min_distance; if (lo < 0) lo = 0;
            int hi = ci + min_distance + 1; if (hi > n) hi = n;
            int namb = 0;
            for (int j0 = lo; j0 < hi; j0 += 64) {
                const int j = j0 + lane;
                const bool in = j < hi;
                const double s32 = in ? (double)c[j] : 0.0;
                const bool big = in && s32 > cv + DELTA;
                const bool amb = in && !big && s32 >= cv - DELTA && j != ci;   // rival within reach: settle exactly
                if (__ballot(big)) return;
                const unsigned long long ma = __ballot(amb);
                const int pos = namb + lanes_below(ma);
                if (amb && pos < 64) S.list[pos] = j;
                namb += __popcll(ma);
            }
            if (namb > 64) { overflow = true; return; }
            if (namb > 0) {
                wave_fence_lds();
                bool beats = false;
                corr64_list(stage, namb, yr, n, tpl, lane, [&](int w) { return S.list[w]; }, [&](int, double jv) { beats |= jv > cv; });
                if (__ballot(beats)) return;
            }
            if (lane == 0 && total < ES_MAX_PEAKS) fo.peaks[rec * ES_MAX_PEAKS + total] = ci;
            ++total;
        };
        for (int g0 = 0; g0 < nc && !overflow; g0 += XL_G) {            // the chunk's candidates, XL_G exact values at a time (value of candidate g0 + w in lane w)
            const int ng = nc - g0 < XL_G ? nc - g0 : XL_G;
            double cvg = 0.0;
            corr64_list(stage, ng, yr, n, tpl, lane, [&](int w) { return nth(g0 + w); }, [&](int, double v) { cvg = v; });
            for (int w = 0; w < ng && !overflow; ++w) settle(nth(g0 + w), lane_f64(cvg, w));
        }
    }
    if (overflow) { flag_out(4); return; }

    if (total == 0) {
        // ---- fallback: five largest exact correlations (descending; equal values -> higher index)
        const int kmax = n < 5 ? n : 5;
        double lo;
        if (have_cum) {
            int mine = -1;                                              // highest bin with >= kmax values at or above it
            #pragma unroll
            for (int b = 0; b < 4; ++b) if ((int)(cum[256] - cum[4 * lane + b]) >= kmax) mine = 4 * lane + b;
            const unsigned long long m = __ballot(mine >= 0);
            const int bstar = es_wave_read_lane(mine, 63 - (int)__builtin_clzll(m));      // bin 0 always qualifies (n >= kmax)
            lo = (bstar == 0) ? -1e300 : (-1.0 + bstar * BINW) - 2.0 * DELTA - 1e-6;
        } else {
            const float t5 = key_f32(pw_select(S, n, n - kmax, lane, [&](int i) { return f32_key(c[i]); }));
            lo = (double)t5 - 2.0 * DELTA;
        }
        int nb = 0;
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + lane;
            const bool in = (i < n) && ((double)c[i] >= lo);
            const unsigned long long mb = __ballot(in);
            const int pos = nb + lanes_below(mb);
            if (in && pos < PW_CAP) S.list[pos] = i;
            nb += __popcll(mb);
        }
        if (nb > PW_CAP) { flag_out(5); return; }
        wave_fence_lds();
        corr64_list(stage, nb, yr, n, tpl, lane, [&](int w) { return S.list[w]; }, [&](int w, double v) { S.val[w] = v; });
        wave_fence_lds();
        for (int idx = lane; idx < nb; idx += 64) {
            const double v = S.val[idx]; const int ii = S.list[idx];
            int before = 0;                                             // how many sort ahead of me
            for (int j = 0; j < nb; ++j) {
                const double vj = S.val[j]; const int ij = S.list[j];
                before += (vj > v) || (vj == v && ij > ii);
            }
            if (before < kmax) fo.peaks[rec * ES_MAX_PEAKS + before] = ii;
        }
        if (lane == 0) fo.npeaks[rec] = kmax | (1 << 30);
        total = kmax;
        wave_fence_lds();
    } else if (lane == 0) {
        fo.npeaks[rec] = total;
    }
    if (lane >= total && lane < ES_MAX_PEAKS) fo.peaks[rec * ES_MAX_PEAKS + lane] = -1;         // unused tail of the row
    if (lane == 0) { fo.thr[rec] = thr; fo.flags[rec] = 0; }
}

__global__ __launch_bounds__(64 * PW_WAVES) void es_pick_exact_wave_kernel(const float* __restrict__ corr32,
        const double* __restrict__ y, long long B, int T, const uint8_t* __restrict__ band,
        const es_band_tables* __restrict__ tabs, double* __restrict__ thr_out, int32_t* __restrict__ peaks,
        int32_t* __restrict__ npeaks, uint8_t* __restrict__ flags)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char pw_smem[];
    const int n = T - (ES_PRE_L - 1);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __builtin_amdgcn_s_setprio(2);      // front-end kernel: issue ahead of a resident list-decoder wave
    const size_t per_wave = sizeof(PwFixed) + (((size_t)n * 4 + 15) & ~(size_t)15);
    PwFixed& S = *reinterpret_cast<PwFixed*>(pw_smem + wv * per_wave);
    float* const c = reinterpret_cast<float*>(pw_smem + wv * per_wave + sizeof(PwFixed));
    const int min_distance = ES_FRAME_LEN / 2;
    const long long stride = (long long)gridDim.x * PW_WAVES;
    for (long long rec = (long long)blockIdx.x * PW_WAVES + wv; rec < B; rec += stride) {
        const float* cg = corr32 + rec * n;
        const double* yr = y + rec * T;
        const double* tpl = tabs->tpl[band[rec]];
        int bad = 0;
        for (int i = lane; i < n; i += 64) {
            const float v = cg[i];
            c[i] = v;
            bad |= !(__builtin_fabsf(v) < 1e30f);                       // inf / nan / absurd: screen unusable
        }
        wave_fence_lds();
        if (__ballot(bad)) { if (lane == 0) flags[rec] = 1; continue; }   // reason 1: non-finite screen
        auto exact_corr = [&](int i) { return corr64_at(yr, i, tpl); };
        const int k_hi = n / 2, k_lo = (n & 1) ? n / 2 : n / 2 - 1;

        // ---- median and MAD (exact)
        double r0, r1;
        if (!pw_exact_stats(S, n, k_lo, k_hi, DELTA, lane, [&](int i) { return (double)c[i]; }, exact_corr, r0, r1)) {
            if (lane == 0) flags[rec] = 2;                               // reason 2: median band too wide
            continue;
        }
        const double med = (n & 1) ? r0 : (r0 + r1) / 2.0;
        if (!pw_exact_stats(S, n, k_lo, k_hi, DELTA, lane, [&](int i) { return __builtin_fabs((double)c[i] - med); },
                            [&](int i) { return __builtin_fabs(corr64_at(yr, i, tpl) - med); }, r0, r1)) {
            if (lane == 0) flags[rec] = 3;                               // reason 3: MAD band too wide
            continue;
        }
        const double mad = ((n & 1) ? r0 : (r0 + r1) / 2.0) + 1e-12;
        double thr = med + 4.5 * 1.4826 * mad;
        if (0.95 < thr) thr = 0.95;

        // ---- threshold crossers in ascending order; each one is settled exactly
        int total = 0;
        bool overflow = false;
        for (int base = 0; base < n && !overflow; base += 64) {
            const int i = base + lane;
            const bool cand = (i < n) && ((double)c[i] >= thr - DELTA);
            unsigned long long m = __ballot(cand);
            if (!m) continue;
            const double cv_mine = cand ? corr64_at(yr, i, tpl) : 0.0;   // all candidates of the chunk at once
            while (m && !overflow) {
                const int bit = __ffsll((long long)m) - 1;
                m &= m - 1;
                const int ci = base + bit;
                const double cv = __shfl(cv_mine, bit);
                if (cv < thr) continue;
                int lo = ci - min_distance; if (lo < 0) lo = 0;
                int hi = ci + min_distance + 1; if (hi > n) hi = n;
                bool bigger = false;
                int namb = 0;
                for (int j0 = lo; j0 < hi && !bigger; j0 += 64) {
                    const int j = j0 + lane;
                    const bool in = j < hi;
                    const double s32 = in ? (double)c[j] : 0.0;
                    const bool big = in && s32 > cv + DELTA;
                    const bool amb = in && !big && s32 >= cv - DELTA && j != ci;   // rival within reach: settle exactly
                    if (__ballot(big)) { bigger = true; break; }
                    const unsigned long long ma = __ballot(amb);
                    const int pos = namb + lanes_below(ma);
                    if (amb && pos < 64) S.list[pos] = j;
                    namb += __popcll(ma);
                }
                if (bigger) continue;
                if (namb > 64) { overflow = true; break; }
                if (namb > 0) {
                    wave_fence_lds();
                    const bool have = lane < namb;
                    const double jv = have ? corr64_at(yr, S.list[lane], tpl) : 0.0;
                    const bool beats = have && jv > cv;
                    wave_fence_lds();
                    if (__ballot(beats)) continue;
                }
                if (lane == 0 && total < ES_MAX_PEAKS) peaks[rec * ES_MAX_PEAKS + total] = ci;
                ++total;
            }
        }
        if (overflow) { if (lane == 0) flags[rec] = 4; continue; }       // reason 4: too many rivals within DELTA

        if (total == 0) {
            // ---- fallback: five largest exact correlations (descending; equal values -> higher index)
            const int kmax = n < 5 ? n : 5;
            const float t5 = key_f32(pw_select(S, n, n - kmax, lane, [&](int i) { return f32_key(c[i]); }));
            const double lo = (double)t5 - 2.0 * DELTA;
            int nb = 0;
            for (int i0 = 0; i0 < n; i0 += 64) {
                const int i = i0 + lane;
                const bool in = (i < n) && ((double)c[i] >= lo);
                const unsigned long long mb = __ballot(in);
                const int pos = nb + lanes_below(mb);
                if (in && pos < PW_CAP) S.list[pos] = i;
                nb += __popcll(mb);
            }
            if (nb > PW_CAP) { if (lane == 0) flags[rec] = 5; continue; }   // reason 5: fallback list too long
            wave_fence_lds();
            for (int idx = lane; idx < nb; idx += 64) S.val[idx] = corr64_at(yr, S.list[idx], tpl);
            wave_fence_lds();
            for (int idx = lane; idx < nb; idx += 64) {
                const double v = S.val[idx]; const int ii = S.list[idx];
                int before = 0;                                          // how many sort ahead of me
                for (int j = 0; j < nb; ++j) {
                    const double vj = S.val[j]; const int ij = S.list[j];
                    before += (vj > v) || (vj == v && ij > ii);
                }
                if (before < kmax) peaks[rec * ES_MAX_PEAKS + before] = ii;
            }
            if (lane == 0) npeaks[rec] = kmax | (1 << 30);
            total = kmax;
            wave_fence_lds();
        } else if (lane == 0) {
            npeaks[rec] = total;
        }
        if (lane >= total && lane < ES_MAX_PEAKS) peaks[rec * ES_MAX_PEAKS + lane] = -1;        // unused tail of the row
        if (lane == 0) { thr_out[rec] = thr; flags[rec] = 0; }
    }
}

}  // namespace

int es_launch_xcorr32(es_ctx* ctx, const float* y32, int64_t B, int T, const uint8_t* band, float* corr32, hipStream_t st)
{
    const int n_lags = T - (ES_PRE_L - 1);
    const long long cap = (long long)ctx->num_cu * XC_GRID_PER_CU;
    // fewer single-segment items than two waves per SIMD: split records four ways
    const bool small = B * ((n_lags + XC_SEG - 1) / XC_SEG) < (long long)ctx->num_cu * 8;
    const bool win2k = !small && T == XC_T_WINDOW;              // config-3 windows: two waves per window, 17 lags per lane, compile-time bounds
    const int seg = small ? 64 * XC_R_SMALL : (win2k ? 64 * XC_R_WINDOW : XC_SEG);
    const long long nseg = (n_lags + seg - 1) / seg;
    long long blocks = (B * nseg + XC_WAVES - 1) / XC_WAVES;
    if (blocks > cap) blocks = cap;
    if (small)
        hipLaunchKernelGGL((es_xcorr32_kernel<XC_R_SMALL, 0, false>), dim3((unsigned)blocks), dim3(64 * XC_WAVES), 0, st, y32,
                           (long long)B, T, band, ctx->d_tables, corr32, FusedArgs{});
    else if (win2k)
        hipLaunchKernelGGL((es_xcorr32_kernel<XC_R_WINDOW, XC_T_WINDOW, false>), dim3((unsigned)blocks), dim3(64 * XC_WAVES), 0, st, y32,
                           (long long)B, T, band, ctx->d_tables, corr32, FusedArgs{});
    else if (T == ES_FRAME_LEN)
        hipLaunchKernelGGL((es_xcorr32_kernel<XC_R, ES_FRAME_LEN, false>), dim3((unsigned)blocks), dim3(64 * XC_WAVES), 0, st, y32,
                           (long long)B, T, band, ctx->d_tables, corr32, FusedArgs{});
    else
        hipLaunchKernelGGL((es_xcorr32_kernel<XC_R, 0, false>), dim3((unsigned)blocks), dim3(64 * XC_WAVES), 0, st, y32,
                           (long long)B, T, band, ctx->d_tables, corr32, FusedArgs{});
    ES_HIP_CHECK(ctx, hipGetLastError());
    return ES_OK;
}

int es_launch_pick_exact(es_ctx* ctx, const float* corr32, const double* y, int64_t B, int T, const uint8_t* band,
                         double* thr, int32_t* peaks, int32_t* npeaks, uint8_t* flags, hipStream_t st)
{
    if (T - (ES_PRE_L - 1) > PX_MAXN) { ctx->err = "es_pick_exact_batch: more than 4096 lags; use the float64 path"; return ES_EINVAL; }
    const int n = T - (ES_PRE_L - 1);
    const size_t per_wave = sizeof(PwFixed) + (((size_t)n * 4 + 15) & ~(size_t)15);
    const size_t lds = per_wave * PW_WAVES;
    if (!ctx->pick_attr_set) {           // per context (= per device): the attribute belongs to the device's copy of the kernel
        ES_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&es_pick_exact_wave_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(PW_WAVES * (sizeof(PwFixed) + PX_MAXN * 4))));
        ctx->pick_attr_set = true;
    }
    long long blocks = (B + PW_WAVES - 1) / PW_WAVES;
    const long long cap = (long long)ctx->num_cu * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(es_pick_exact_wave_kernel, dim3((unsigned)blocks), dim3(64 * PW_WAVES), lds, st, corr32, y, (long long)B, T,
                       band, ctx->d_tables, thr, peaks, npeaks, flags);
    ES_HIP_CHECK(ctx, hipGetLastError());
    return ES_OK;
}

int es_launch_sync_fused(es_ctx* ctx, const float* y32, const double* y, int64_t B, int T, const uint8_t* band, double* thr,
                         int32_t* peaks, int32_t* npeaks, uint8_t* flags, hipStream_t st)
{
    const int n_lags = T - (ES_PRE_L - 1);
    if (n_lags > PX_MAXN) { ctx->err = "es_sync_fused_batch: more than 4096 lags; use the float64 path"; return ES_EINVAL; }
    const bool win2k = T == XC_T_WINDOW;
    const int R = win2k ? XC_R_WINDOW : XC_R;
    const long long nseg = (n_lags + 64 * R - 1) / (64 * R);
    if ((long long)B * nseg >= (1LL << 31)) { ctx->err = "es_sync_fused_batch: batch too large for one launch"; return ES_EINVAL; }
    const size_t lds = XF_WAVES * xf_lds_per_wave(64 * R + ES_PRE_L - 1 + 2, n_lags);
    long long blocks = (B + XF_WAVES - 1) / XF_WAVES;
    const long long cap = (long long)ctx->num_cu * 32;
    if (blocks > cap) blocks = cap;
    const FusedArgs fo{y, thr, peaks, npeaks, flags};
    if (win2k)
        hipLaunchKernelGGL((es_xcorr32_kernel<XC_R_WINDOW, XC_T_WINDOW, true>), dim3((unsigned)blocks), dim3(64 * XF_WAVES), lds, st, y32,
                           (long long)B, T, band, ctx->d_tables, (float*)nullptr, fo);
    else if (T == ES_FRAME_LEN)
        hipLaunchKernelGGL((es_xcorr32_kernel<XC_R, ES_FRAME_LEN, true>), dim3((unsigned)blocks), dim3(64 * XF_WAVES), lds, st, y32,
                           (long long)B, T, band, ctx->d_tables, (float*)nullptr, fo);
    else
        hipLaunchKernelGGL((es_xcorr32_kernel<XC_R, 0, true>), dim3((unsigned)blocks), dim3(64 * XF_WAVES), lds, st, y32,
                           (long long)B, T, band, ctx->d_tables, (float*)nullptr, fo);
    ES_HIP_CHECK(ctx, hipGetLastError());
    return ES_OK;
}
