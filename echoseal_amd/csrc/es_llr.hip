// es_llr.hip -- per-frame soft demodulation (WatermarkDetector._llr, rtwm/detector.py:296-416)
// for gfx950: matched filter -> integer chip-shift search -> PN despread -> robust scaling.
//
// One WAVE per frame record, one-wave blocks (round 1 used a 256-thread block per record: ~14 block barriers and phases in
// which three of four waves waited for the fourth -- 2.3 ms per 65 536 records against 1.4 ms now).  LDS per wave 17.2 KB.
//
// Numerics follow the reference's NumPy float32 data flow step by step:
//   * matched filter: float64 accumulation of exact float32 products, ascending sample order,
//     rounded once to float32 (the reference's BLAS sdot order is machine dependent; see
//     oracle/c/eso_dsp.c for the rationale -- kernel and oracle use the same definition);
//   * every float32 sum (shift scores, mean, variance) reproduces NumPy's pairwise reduction:
//     blocks of <=128 elements, 8 strided accumulators combined as ((0+1)+(2+3))+((4+5)+(6+7)),
//     recursive halving above 128.  With <=1024 elements that tree has at most 16 leaves, which is
//     exactly one wave: lane = leaf*4 + j owns accumulators j and j+4 of its leaf, combined with
//     xor-shuffles 1,2 (inside a leaf) and 4,8,16,32 (across leaves).  Absent leaves contribute
//     +0.0, which is exact;
//   * medians are exact order statistics (block-wide 8-bit-digit radix select on the 32-bit keys);
//   * the matched filter is register tiled: 7 consecutive outputs per thread slide over the
//     samples they share (zero-padded in LDS, so the tap loop has no bounds).
//
// Build with -ffp-contract=off.
#include "es_internal.h"
#include <cstdio>

namespace {

constexpr int NPAY = ES_POLAR_N;
constexpr int PAYLOAD_START = ES_PRE_L + ES_HDR_L;   // 191
// Sizes that follow from the longest matched filter the kernel must hold.  Two instantiations: ES_MAX_TAPS_FAST (160: the taps of the
// default fs_target = 48 000 are 93..131 long) and ES_MAX_TAPS (576: any fs_target the reference's band plan admits down to 44 100 Hz, where
// the 18-22 kHz band sits at Nyquist and the cascade's impulse response needs 550 taps; rtwm/detector.py:260-294).
#ifndef ES_LLR_MF_R
#define ES_LLR_MF_R 7
#endif
constexpr int MF_R = ES_LLR_MF_R;                              // matched-filter outputs per thread (odd: a lane stride of 7 words is free of LDS bank conflicts; 6 was two-way)
template <int MAXT> struct LlrSizes {
    static constexpr int MAX_RX = NPAY + MAXT;                             // prefix + payload
    static constexpr int MAX_WIN = NPAY + 2 * MAXT + 8;                    // matched-filter window
    static constexpr int MF_PAD = (MAXT + MF_R - 1) / MF_R * MF_R + 15;    // >= MAXT rounded up to a multiple of MF_R, + slack (176 at MAXT = 160)
    static constexpr int LW_NSH = 2 * MAXT + 8;                            // shifts, at most
    static constexpr int HD_MAXWIN = ES_HDR_L + 2 * MAXT + 8;
};
static_assert(LlrSizes<ES_MAX_TAPS_FAST>::MF_PAD == 176, "the default instantiation keeps its layout");

struct PwPlan { int start[16]; int len[16]; };

// NumPy pairwise_sum split points for a vector of n (<= 1024) elements, laid out on a full
// depth-4 binary tree (slots 0..15).  A node longer than 128 is split at h = n/2 - (n/2)%8 into
// (slot, h) and (slot + span/2, n - h); a node that is already <= 128 long stays where it is and
// its unused sibling slots keep length 0.  Four levels suffice: a depth-4 piece is at most
// n/16 + 14 <= 78 elements.
__device__ __forceinline__ void pw_plan_build(PwPlan& p, int n)
{
    #pragma unroll
    for (int s = 0; s < 16; ++s) { p.start[s] = 0; p.len[s] = 0; }
    p.len[0] = n;
    #pragma unroll
    for (int span = 16; span >= 2; span >>= 1) {
        #pragma unroll
        for (int s = 0; s < 16; s += span) {
            if (p.len[s] > 128) {
                int h = p.len[s] / 2; h -= h % 8;
                p.start[s + span / 2] = p.start[s] + h;
                p.len[s + span / 2] = p.len[s] - h;
                p.len[s] = h;
            }
        }
    }
}

// One wave evaluates NumPy's pairwise float32 sum of get(i), i in [0, n), using the plan:
// lane = leaf*4 + jj owns accumulators jj and jj+4 of its leaf.  All lanes return the total.
template <typename F>
__device__ __forceinline__ float wave_pairwise_sum(const PwPlan& p, int lane, F get)
{
    const int leaf = lane >> 2, jj = lane & 3;
    int st = 0, ln = 0;
    #pragma unroll
    for (int s = 0; s < 16; ++s) if (s == leaf) { st = p.start[s]; ln = p.len[s]; }
    float r;
    if (ln >= 8) {
        float ra = get(st + jj), rb = get(st + jj + 4);
        const int full = ln - (ln % 8);
        for (int i = 8; i < full; i += 8) { ra = ra + get(st + i + jj); rb = rb + get(st + i + jj + 4); }
        ra = ra + __shfl_xor(ra, 1); rb = rb + __shfl_xor(rb, 1);      // (r0+r1) , (r4+r5)
        ra = ra + __shfl_xor(ra, 2); rb = rb + __shfl_xor(rb, 2);      // +(r2+r3), +(r6+r7)
        r = ra + rb;
        for (int i = full; i < ln; ++i) r = r + get(st + i);
    } else {
        r = 0.0f;
        for (int i = 0; i < ln; ++i) r = r + get(st + i);
    }
    r = r + __shfl_xor(r, 4);
    r = r + __shfl_xor(r, 8);
    r = r + __shfl_xor(r, 16);
    r = r + __shfl_xor(r, 32);
    return r;
}

// lane l <-> lane l ^ S through the data-parallel primitives (quad_perm for 1, 2; a row shift each way and a
// select for 4, 8); 16 and 32 go through ds_bpermute.
template <int S>
__device__ __forceinline__ float xor_lanes_f32(float x, int lane)
{
    int v; __builtin_memcpy(&v, &x, 4);
    int r;
    if constexpr (S == 1) r = __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true);
    else if constexpr (S == 2) r = __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true);
    else if constexpr (S == 4 || S == 8) {
        const int up = __builtin_amdgcn_mov_dpp(v, 0x100 + S, 0xf, 0xf, true);     // row_shl:S (lane l gets l+S)
        const int dn = __builtin_amdgcn_mov_dpp(v, 0x110 + S, 0xf, 0xf, true);     // row_shr:S (lane l gets l-S)
        r = (lane & S) ? dn : up;
    } else r = __shfl_xor(v, S);
    float o; __builtin_memcpy(&o, &r, 4); return o;
}

// The shift search evaluates the same pairwise tree hundreds of times per record on NON-NEGATIVE data (|win|), which
// allows a latency-free form: every lane issues all the LDS reads of its leaf up front (absent elements read as
// +0.0, and x + (+0.0) == x exactly for x >= +0), then adds in NumPy's order.  A leaf is at most 128 long (NumPy
// only splits nodes longer than that): 16 strided steps of 8 and at most 7 trailing elements.
struct PwLane { int st, full, ln; };
__device__ __forceinline__ PwLane pw_lane(const PwPlan& p, int lane)
{
    const int leaf = lane >> 2;
    int st = 0, ln = 0;
    #pragma unroll
    for (int s = 0; s < 16; ++s) if (s == leaf) { st = p.start[s]; ln = p.len[s]; }
    PwLane g; g.st = st; g.ln = ln; g.full = (ln >= 8) ? ln - (ln % 8) : 0;
    return g;
}
// (|a[i]| is taken on the fly: the kernel keeps no separate |win| array)
__device__ __forceinline__ float wave_pairwise_sum_nonneg_abs(const PwLane& g, int lane, const float* __restrict__ a)
{
    constexpr int NIT = 16;
    const int jj = lane & 3;
    float x0[NIT], x1[NIT], tl[7];
    #pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const bool ok = 8 * k < g.full;
        const int idx = ok ? g.st + 8 * k + jj : 0;
        const float u = __builtin_fabsf(a[idx]), v = __builtin_fabsf(a[idx + 4]);
        x0[k] = ok ? u : 0.0f; x1[k] = ok ? v : 0.0f;
    }
    #pragma unroll
    for (int k = 0; k < 7; ++k) {
        const bool ok = g.full + k < g.ln;
        const float u = __builtin_fabsf(a[ok ? g.st + g.full + k : 0]);
        tl[k] = ok ? u : 0.0f;
    }
    float ra = x0[0], rb = x1[0];
    #pragma unroll
    for (int k = 1; k < NIT; ++k) { ra = ra + x0[k]; rb = rb + x1[k]; }
    ra = ra + xor_lanes_f32<1>(ra, lane); rb = rb + xor_lanes_f32<1>(rb, lane);     // (r0+r1) , (r4+r5)
    ra = ra + xor_lanes_f32<2>(ra, lane); rb = rb + xor_lanes_f32<2>(rb, lane);     // +(r2+r3), +(r6+r7)
    float r = ra + rb;
    #pragma unroll
    for (int k = 0; k < 7; ++k) r = r + tl[k];
    r = r + xor_lanes_f32<4>(r, lane);
    r = r + xor_lanes_f32<8>(r, lane);
    r = r + xor_lanes_f32<16>(r, lane);
    r = r + xor_lanes_f32<32>(r, lane);
    return r;
}

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

// k-th smallest of the wave's keys by ONE wave: four 8-bit-digit passes on four private LDS histograms (lane & 3: the leading byte of a
// float takes few values, and 64 lanes adding to one word would serialise), wave scan.  The keys (at most SEL_KPL per lane, kx[u] = key of
// element lane + 64 u, `nmine` of them valid) live in registers for all four passes: read from LDS inside each pass they cost a
// dependent round trip per element and pass.
constexpr int SEL_KPL = NPAY / 64;
__device__ uint32_t wave_select_key32(uint32_t (*hist)[256], const uint32_t (&kx)[SEL_KPL], int nmine, int k, int lane)
{
    uint32_t prefix = 0;
    int kk = k;
    for (int shift = 24; shift >= 0; shift -= 8) {
        #pragma unroll
        for (int b = 0; b < 16; ++b) (&hist[0][0])[lane + 64 * b] = 0;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
        const uint32_t himask = (shift == 24) ? 0u : (~0u << (shift + 8));
        uint32_t* const myh = hist[lane & 3];
        #pragma unroll
        for (int u = 0; u < SEL_KPL; ++u)
            if (u < nmine && (kx[u] & himask) == prefix) atomicAdd(&myh[(kx[u] >> shift) & 255u], 1u);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
        uint32_t h[4];
        #pragma unroll
        for (int b = 0; b < 4; ++b) h[b] = hist[0][4 * lane + b] + hist[1][4 * lane + b] + hist[2][4 * lane + b] + hist[3][4 * lane + b];
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
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
    }
    return prefix;
}

// NumPy median of val(i), i in [0, n), n <= NPAY, by one wave: one select; for even n the upper middle element comes from one
// counting pass (it is the lower one again if enough keys are <= it, else the smallest key above it).
template <typename F>
__device__ float wave_median_hist_f32(uint32_t (*hist)[256], int n, int lane, F val)
{
    uint32_t kx[SEL_KPL];
    const int nmine = (n > lane) ? (n - lane + 63) / 64 : 0;            // elements lane, lane + 64, ... below n
    #pragma unroll
    for (int u = 0; u < SEL_KPL; ++u) kx[u] = (u < nmine) ? f32_key(val(lane + 64 * u)) : 0xffffffffu;
    if (n & 1) return key_f32(wave_select_key32(hist, kx, nmine, n / 2, lane));
    const int k_lo = n / 2 - 1, k_hi = n / 2;
    const uint32_t key_lo = wave_select_key32(hist, kx, nmine, k_lo, lane);
    int le = 0; uint32_t nxt = 0xffffffffu;
    #pragma unroll
    for (int u = 0; u < SEL_KPL; ++u) {
        if (u < nmine) {
            le += kx[u] <= key_lo;
            if (kx[u] > key_lo && kx[u] < nxt) nxt = kx[u];
        }
    }
    #pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        le += __shfl_xor(le, o);
        const uint32_t on = (uint32_t)__shfl_xor((int)nxt, o);
        nxt = on < nxt ? on : nxt;
    }
    const float lo = key_f32(key_lo), hi = (le > k_hi) ? lo : key_f32(nxt);
    return (lo + hi) / 2.0f;
}

// ---- the demodulator: one WAVE per record ----------------------------------------------------------------------------
// A record belongs to one wave from the first load to the last store: no block barriers.  LDS per wave is 17.2 KB -- the PN
// symbols stay packed, |win| is taken on the fly, only the float64 prefix sums that the 2 x (2 max_shift + 1) window ends need
// are kept, and regions are reused across phases -- and a block is one wave, so that it fits beside three resident
// list-decoder blocks (41 KB of LDS left) and nine fit an otherwise empty CU.
constexpr int LW_WAVES = 1;                           // one-wave blocks (17.2 KB of LDS): they fit beside three resident list-decoder blocks
template <int MAXT>
struct LlrWaveLds {
    static constexpr int MAX_RX = LlrSizes<MAXT>::MAX_RX, MAX_WIN = LlrSizes<MAXT>::MAX_WIN, MF_PAD = LlrSizes<MAXT>::MF_PAD, LW_NSH = LlrSizes<MAXT>::LW_NSH;
    union {
        struct { float rx[MF_PAD + MAX_RX + MF_PAD]; double h[MF_PAD]; } mf;    // matched-filter inputs, zero padded (taps already as float64: one conversion per tap, not per use)
        double pre[2][LW_NSH];                                               // prefix sums at the window ends of every shift
        float d[NPAY];                                                        // despread values
    } a;
    float win[MAX_WIN];
    union {
        struct { double A[LW_NSH]; int cand[LW_NSH]; } sh;                    // exact-sum score per shift, candidate shifts
        uint32_t hist[4][256];                                                // radix-select histograms of the robust statistics
    } b;
    uint32_t pnw[32];                                                         // payload PN bits, first = MSB of word 0
};

#ifdef ES_LLR_STAMPS
// Diagnostic build: cycles per phase, summed over the records of a launch.  The stamps are atomics, i.e. vector-memory operations: the first
// interval that waits on vmcnt (the sample loads) also waits for them, so "load" reads several times too large (the load phase alone, with
// the arithmetic compiled out, is 0.13 ms of the kernel's 1.2 ms per 65 536 records); the other phases compare with each other.
__device__ unsigned long long g_llr_dbg[16];
#define LLR_STAMP(k) do { __builtin_amdgcn_s_waitcnt(0xC07F); const unsigned long long _t = __builtin_amdgcn_s_memtime(); \
        if (lane == 0) atomicAdd(&g_llr_dbg[k], _t - t_last); t_last = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define LLR_STAMP(k) do { } while (0)
#endif

template <int MAXT>
__global__ __launch_bounds__(64 * LW_WAVES, 3) void es_llr_wave_kernel(const double* __restrict__ y, long long B,
        int T, const int32_t* __restrict__ start, const uint8_t* __restrict__ band,
        const uint8_t* __restrict__ pn_rows, int variant, const es_band_tables* __restrict__ tabs,
        float* __restrict__ llr, int32_t* __restrict__ best_s_out, float* __restrict__ score_out)
{
    constexpr int MAX_RX = LlrSizes<MAXT>::MAX_RX, MF_PAD = LlrSizes<MAXT>::MF_PAD;
    __shared__ __attribute__((aligned(16))) LlrWaveLds<MAXT> s_w[LW_WAVES];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    LlrWaveLds<MAXT>& W = s_w[wv];
    auto fence = [&]() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    __builtin_amdgcn_s_setprio(2);      // front-end kernel: issue ahead of a resident list-decoder wave
    const long long stride = (long long)gridDim.x * LW_WAVES;
    // Per-record metadata one record ahead (start: a scalar load; band), the four tap counts once: a record's first sample request then
    // waits for nothing -- read at the top of its own turn they were two to three dependent round trips in front of the samples'.
    const int nt0 = tabs->ntaps[0], nt1 = tabs->ntaps[1], nt2 = tabs->ntaps[2], nt3 = tabs->ntaps[3];
    static_assert(LW_WAVES == 1, "the record index below is the block index: wave-uniform by construction, so that the metadata loads are scalar");
    long long rec = (long long)blockIdx.x;
    int st_next = (rec < B && start) ? start[rec] : 0;
    int bi_next = (rec < B) ? band[rec] : 0;
    for (; rec < B; rec += stride) {
#ifdef ES_LLR_STAMPS
        unsigned long long t_last = __builtin_amdgcn_s_memtime();
#endif
        float* out = llr + rec * NPAY;
        const int st0 = st_next;
        const int bi = bi_next;
        int flen = T - st0; if (flen > ES_FRAME_LEN) flen = ES_FRAME_LEN;
        const int ntaps = bi == 0 ? nt0 : bi == 1 ? nt1 : bi == 2 ? nt2 : nt3;
        const int mem = ntaps - 1;
        const int npl = flen - PAYLOAD_START;                         // payload samples present
        if (st0 < 0 || npl <= 0) {                                    // detector.py:320-325
            for (int i = lane; i < NPAY; i += 64) out[i] = 0.0f;
            if (lane == 0) { if (best_s_out) best_s_out[rec] = 0; if (score_out) { score_out[2 * rec] = -1.0f; score_out[2 * rec + 1] = -1.0f; } }
            const long long nrec = rec + stride;                      // (the next record's metadata: see below)
            st_next = (nrec < B && start) ? start[nrec] : 0;
            bi_next = (nrec < B) ? band[nrec] : 0;
            continue;
        }
        const double* fr = y + rec * T + st0;
        const int prefix = mem < PAYLOAD_START ? mem : PAYLOAD_START; // :327
        const int nfull = prefix + npl;
        LLR_STAMP(8);
        {
            // All the record's samples are requested before the first one is used: unconditional loads from clamped addresses, no
            // control flow between a load and its use (a bounds test per element made the compiler wait for every load where it was
            // issued: 19 dependent round trips to HBM per record; 1.39 -> 1.21 ms per 65 536 records).  The pads are plain LDS stores.
            constexpr int NLD = (MAX_RX + 63) / 64;
            const double* src = fr + (PAYLOAD_START - prefix);
            double v[NLD];
            float tp[(MF_PAD + 63) / 64];
            #pragma unroll
            for (int k = 0; k < NLD; ++k) { const int ii = lane + 64 * k; v[k] = src[ii < nfull ? ii : nfull - 1]; }
            #pragma unroll
            for (int k = 0; k < (MF_PAD + 63) / 64; ++k) { const int i = lane + 64 * k; tp[k] = tabs->taps[bi][i < ntaps ? i : ntaps - 1]; }
            {   // the next record's metadata: scalar loads that complete under the wait for this record's samples
                const long long nrec = rec + stride;
                st_next = (nrec < B && start) ? start[nrec] : 0;
                bi_next = (nrec < B) ? band[nrec] : 0;
            }
            for (int i = lane; i < MF_PAD; i += 64) { W.a.mf.rx[i] = 0.0f; W.a.mf.rx[MF_PAD + MAX_RX + i] = 0.0f; }
            #pragma unroll
            for (int k = 0; k < NLD; ++k) { const int ii = lane + 64 * k; if (ii < MAX_RX) W.a.mf.rx[MF_PAD + ii] = (ii < nfull) ? (float)v[k] : 0.0f; }
            #pragma unroll
            for (int k = 0; k < (MF_PAD + 63) / 64; ++k) { const int i = lane + 64 * k; if (i < MF_PAD) W.a.mf.h[i] = (i < ntaps) ? (double)tp[k] : 0.0; }
        }
        const int n = NPAY < npl ? NPAY : npl;                        // :337
        if (lane < 32) {                                              // 32 PN bits per lane, MSB first, from the packed row (:306-312)
            const uint8_t* pnr = pn_rows + rec * ES_PN_BYTES;
            const int p0 = ((variant == 0) ? PAYLOAD_START : 0) + 32 * lane;
            const int b0 = p0 >> 3, sh = p0 & 7;
            uint64_t v = 0;
            #pragma unroll
            for (int k = 0; k < 5; ++k) v = (v << 8) | (uint64_t)((b0 + k < ES_PN_BYTES) ? pnr[b0 + k] : 0);
            W.pnw[lane] = (uint32_t)(v >> (8 - sh));
        }
        fence();
        LLR_STAMP(0);
        auto pn_sym = [&](int i) { return ((W.pnw[i >> 5] >> (31 - (i & 31))) & 1u) ? 1.0f : -1.0f; };

        // ---- geometry (:335-363)
        const int nmf = nfull + ntaps - 1;
        const int offset = prefix + mem;
        int raw_shift = n / 2;
        if (4 * ntaps < raw_shift) raw_shift = 4 * ntaps;
        if (ES_HDR_L < raw_shift) raw_shift = ES_HDR_L;
        const int max_shift = mem > raw_shift ? mem : raw_shift;
        const int wstart = offset - max_shift > 0 ? offset - max_shift : 0;
        const int wstop = nmf < offset + n + max_shift ? nmf : offset + n + max_shift;
        const int nwin = wstop - wstart;
        const int base = offset - wstart;
        int guard = ntaps / 2 > 24 ? ntaps / 2 : 24;
        if (n / 4 < guard) guard = n / 4;
        if (guard >= n) guard = n / 4 > 0 ? n / 4 : 0;

        // ---- matched filter window (:334): mf[j] = sum_i rx[i] h[j-i], i ascending (= tap k = j-i descending), float64
        // accumulation of exact float32 products, rounded once.  Each lane owns MF_R consecutive outputs per pass and slides
        // a register window over the samples they share; zero padding (samples and taps) only adds exact zeros.
        {
            const int k6 = ((ntaps + MF_R - 1) / MF_R) * MF_R;
            for (int w0 = lane * MF_R; w0 < nwin; w0 += 64 * MF_R) {
                const int jj0 = wstart + w0;
                double acc[MF_R];
                #pragma unroll
                for (int r = 0; r < MF_R; ++r) acc[r] = 0.0;
                const float* px = W.a.mf.rx + MF_PAD + jj0 - (k6 - 1);
                double x[2 * MF_R - 1];
                #pragma unroll
                for (int r = 0; r < MF_R - 1; ++r) x[r] = (double)px[r];
                for (int kb = k6 - 1; kb >= 0; kb -= MF_R) {
                    #pragma unroll
                    for (int u = 0; u < MF_R; ++u) x[MF_R - 1 + u] = (double)px[MF_R - 1 + u];
                    #pragma unroll
                    for (int u = 0; u < MF_R; ++u) {
                        const double hk = W.a.mf.h[kb - u];
                        #pragma unroll
                        for (int r = 0; r < MF_R; ++r) acc[r] = __builtin_fma(x[u + r], hk, acc[r]);
                    }
                    #pragma unroll
                    for (int r = 0; r < MF_R - 1; ++r) x[r] = x[MF_R + r];
                    px += MF_R;
                }
                #pragma unroll
                for (int r = 0; r < MF_R; ++r) if (w0 + r < nwin) W.win[w0 + r] = (float)acc[r];
            }
        }
        fence();
        LLR_STAMP(1);

        // ---- shift search (:366-379): score(s) = mean(|win[base+s+i] * pn[i]|, i >= guard).  pn[i] is +-1, so |win * pn| ==
        // |win| exactly: the scores are NumPy pairwise sums over sliding windows of |win|.  (The sums themselves cannot slide:
        // float32 addition order is part of the result.)  Screen: a float32 pairwise sum of non-negative terms is within
        // (1+2^-24)^30 - 1 < 1.8e-6 (relative) of the exact sum, and the exact window sums of all shifts come from ONE float64
        // prefix sum.  Only shifts whose exact-sum score is within SCREEN_R of the runner-up can be the winner or the runner-up
        // that the kernel reports; those -- usually two to four of the ~260 -- get the NumPy-order float32 evaluation.  Flat or
        // non-finite data simply leaves every shift a candidate.
        PwPlan plan;
        pw_plan_build(plan, n - guard);
        const float cnt_f = (float)(n - guard);
        const PwLane geo = pw_lane(plan, lane);
        constexpr double SCREEN_R = 4e-6;
        const int nshift = 2 * max_shift + 1;
        const int lo_a = base - max_shift + guard, lo_b = base - max_shift + n;      // pre[lo_a + u], pre[lo_b + u] for shift index u
        double total;
        {
            const int chunk = (nwin + 63) / 64;
            const int j0 = lane * chunk, j1 = (j0 + chunk < nwin) ? j0 + chunk : nwin;
            double ls = 0.0;
            for (int j = j0; j < j1; ++j) ls += (double)__builtin_fabsf(W.win[j]);
            double incl = ls;
            #pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const double up = __shfl_up(incl, o); if (lane >= o) incl += up; }
            total = __shfl(incl, 63);
            double run = incl - ls;
            auto put = [&](int p, double v) {                       // p = number of elements summed
                const int ua = p - lo_a, ub = p - lo_b;
                if (ua >= 0 && ua < nshift) W.a.pre[0][ua] = v;
                if (ub >= 0 && ub < nshift) W.a.pre[1][ub] = v;
            };
            fence();                                                 // (the matched-filter inputs in region A are dead)
            if (lane == 0) put(0, 0.0);
            for (int j = j0; j < j1; ++j) { run += (double)__builtin_fabsf(W.win[j]); put(j + 1, run); }
        }
        fence();
        LLR_STAMP(2);
        const bool finite_all = total < 1.0e37;                      // float32 sums cannot overflow below this
        double a1 = -1.0, a2 = -1.0;                                 // two largest exact-sum scores (with multiplicity)
        for (int u = lane; u < nshift; u += 64) {
            const int i0 = base + (u - max_shift);
            double A = -1.0;
            if (i0 >= 0 && i0 + n <= nwin) A = (W.a.pre[1][u] - W.a.pre[0][u]) / (double)(n - guard);
            W.b.sh.A[u] = A;
            if (A > a1) { a2 = a1; a1 = A; } else if (A > a2) a2 = A;
        }
        #pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const double o1 = __shfl_xor(a1, o), o2 = __shfl_xor(a2, o);
            const double hi = a1 > o1 ? a1 : o1, lo = a1 > o1 ? o1 : a1;
            const double m2 = a2 > o2 ? a2 : o2;
            a1 = hi; a2 = lo > m2 ? lo : m2;
        }
        fence();
        const double tau = (finite_all && a2 >= 1.0e-30) ? a2 * (1.0 - SCREEN_R) : -2.0;
        int ncand = 0;
        for (int u0 = 0; u0 < nshift; u0 += 64) {                    // ordered compaction, ascending shift
            const int u = u0 + lane;
            const bool isc = (u < nshift) && (W.b.sh.A[u] >= 0.0) && (W.b.sh.A[u] * (1.0 + SCREEN_R) >= tau);
            const unsigned long long mb = __ballot(isc);
            const int below = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mb, 0u));
            if (isc) W.b.sh.cand[ncand + below] = u - max_shift;
            ncand += __popcll(mb);
        }
        fence();
        LLR_STAMP(3);
        float best = -1.0f, second = -1.0f; int best_s = 0;
        for (int ci = 0; ci < ncand; ++ci) {                         // ascending shift: the first maximum wins (strict > in the reference)
            const int s = W.b.sh.cand[ci];
            const float sum = wave_pairwise_sum_nonneg_abs(geo, lane, W.win + base + s + guard);
            const float score = sum / cnt_f;
            if (score > best) { second = best; best = score; best_s = s; }
            else if (score > second) second = score;
        }

        LLR_STAMP(4);
        // ---- despread at the chosen shift (:382-385)
        const int a0 = base + best_s;
        fence();
        for (int i = lane; i < n; i += 64) W.a.d[i] = W.win[a0 + i] * pn_sym(i);
        fence();

        // ---- robust statistics on the tail (:395-404): median and MAD (exact selects), then mean and variance (NumPy order)
        const int toff = (n > guard + 8) ? guard : 0;
        const int nt = n - toff;
        const float* tail = W.a.d + toff;
        const float medv = wave_median_hist_f32(W.b.hist, nt, lane, [&](int i) { return tail[i]; });
        const float madv = wave_median_hist_f32(W.b.hist, nt, lane, [&](int i) { return __builtin_fabsf(tail[i] - medv); });
        LLR_STAMP(5);
        PwPlan tp; pw_plan_build(tp, nt);
        const float mu = wave_pairwise_sum(tp, lane, [&](int i) { return tail[i]; }) / (float)nt;
        const float var = wave_pairwise_sum(tp, lane, [&](int i) { const float c = tail[i] - mu; return c * c; }) / (float)nt;
        LLR_STAMP(6);
        const double mad = (double)madv + 1e-12;
        const double sigma_mad = 1.4826 * mad;
        const double sigma_std = (double)__builtin_sqrtf(var) + 1e-12;
        double sigma = sigma_mad > sigma_std ? sigma_mad : sigma_std;
        if (0.1 > sigma) sigma = 0.1;
        double scale = 2.0 / (sigma * sigma);
        if (scale < 0.5) scale = 0.5;
        if (scale > 30.0) scale = 30.0;
        const float scale32 = (float)scale;
        for (int i = lane; i < NPAY; i += 64) {
            float v = 0.0f;
            if (i < n) {
                v = (W.a.d[i] - mu) * scale32;                        // :397,405
                if (v < -12.0f) v = -12.0f;
                if (v > 12.0f) v = 12.0f;
            }
            out[i] = v;
        }
        if (lane == 0) {
            if (best_s_out) best_s_out[rec] = best_s;
            if (score_out) { score_out[2 * rec] = best; score_out[2 * rec + 1] = second; }
        }
        fence();
        LLR_STAMP(7);
    }
}

// ---- header decode (WatermarkDetector._decode_header, rtwm/detector.py:452-515) -----------------
// One wave per record.  Matched filter on prefix + 128 header chips, shift search on
// |sum(win * hdr_pn)[guard:]| (first maximum in ascending shift order), then 16 x 8 majority.
// Eight shifts are scored at a time: lane = slot*8 + accumulator reproduces NumPy's <=128-element
// pairwise leaf (8 strided accumulators, ((0+1)+(2+3))+((4+5)+(6+7)), tail added sequentially).

// NumPy pairwise leaf (8 <= n <= 128) evaluated by the 8 lanes of a slot; all 8 lanes return it.
template <typename F>
__device__ __forceinline__ float slot_leaf_sum(int n, int j, F get)
{
    float r = get(j);
    const int full = n - (n % 8);
    for (int i = 8; i < full; i += 8) r = r + get(i + j);
    r = r + __shfl_xor(r, 1);
    r = r + __shfl_xor(r, 2);
    r = r + __shfl_xor(r, 4);
    for (int i = full; i < n; ++i) r = r + get(i);
    return r;
}

template <int MAXT>
__global__ __launch_bounds__(64) void es_header_kernel(const double* __restrict__ y, long long B, int T,
        const int32_t* __restrict__ start, const uint8_t* __restrict__ band, const uint8_t* __restrict__ hdr_pn,
        const es_band_tables* __restrict__ tabs, uint8_t* __restrict__ ok_out, int32_t* __restrict__ val_out,
        float* __restrict__ score_out, int32_t* __restrict__ best_s_out)
{
    __shared__ float s_rx[ES_PRE_L + ES_HDR_L];
    constexpr int HD_MAXWIN = LlrSizes<MAXT>::HD_MAXWIN;
    __shared__ float s_h[MAXT];
    __shared__ float s_pn[ES_HDR_L];
    __shared__ float s_win[HD_MAXWIN];
    __shared__ float s_d[ES_HDR_L];
    __shared__ float s_sums[16];
    const int lane = threadIdx.x;
    for (long long rec = blockIdx.x; rec < B; rec += gridDim.x) {
        const int st0 = start ? start[rec] : 0;
        const int flen = T - st0;
        if (st0 < 0 || flen < ES_PRE_L + ES_HDR_L) {                   // :461-462
            if (lane == 0) { ok_out[rec] = 0; val_out[rec] = 0; score_out[rec] = 0.0f; if (best_s_out) best_s_out[rec] = 0; }
            continue;
        }
        const int bi = band[rec];
        const int ntaps = tabs->ntaps[bi];
        const int mem = ntaps - 1;
        const int prefix = mem < ES_PRE_L ? mem : ES_PRE_L;            // :466
        const int nfull = prefix + ES_HDR_L;
        const double* fr = y + rec * T + st0;
        for (int i = lane; i < nfull; i += 64) s_rx[i] = (float)fr[ES_PRE_L - prefix + i];
        for (int i = lane; i < ntaps; i += 64) s_h[i] = tabs->taps[bi][i];
        const uint8_t* pnr = hdr_pn + rec * (ES_HDR_L / 8);
        for (int i = lane; i < ES_HDR_L; i += 64)
            s_pn[i] = 2.0f * (float)((pnr[i >> 3] >> (7 - (i & 7))) & 1u) - 1.0f;
        __syncthreads();
        const int nmf = nfull + ntaps - 1;
        const int offset = mem + prefix;                               // :474
        int max_shift = ES_HDR_L / 2 + prefix;                         // :475-478
        if (4 * ntaps < max_shift) max_shift = 4 * ntaps;
        if (max_shift < mem) max_shift = mem;
        const int wstart = offset - max_shift > 0 ? offset - max_shift : 0;
        const int wstop = nmf < offset + ES_HDR_L + max_shift ? nmf : offset + ES_HDR_L + max_shift;
        const int nwin = wstop - wstart;
        const int base = offset - wstart;
        int guard = ntaps / 8 < 32 ? ntaps / 8 : 32;                   // :484
        if (guard < 8) guard = 8;
        for (int w = lane; w < nwin; w += 64) {                        // :473 (same MF definition as _llr)
            const int jj = wstart + w;
            int i0 = jj - (ntaps - 1); if (i0 < 0) i0 = 0;
            const int i1 = jj < nfull - 1 ? jj : nfull - 1;
            double acc = 0.0;
            for (int i = i0; i <= i1; ++i) acc += (double)s_rx[i] * (double)s_h[jj - i];
            s_win[w] = (float)acc;
        }
        __syncthreads();
        // shift search, 8 shifts per round
        const int slot = lane >> 3, j = lane & 7;
        const int n = ES_HDR_L - guard;
        float best = -1.0f; int best_s = 0;
        for (int s0 = -max_shift; s0 <= max_shift; s0 += 8) {
            const int s = s0 + slot;
            const int i0 = base + s;
            const bool valid = (s <= max_shift) && i0 >= 0 && i0 + ES_HDR_L <= nwin;
            const float* a = s_win + (valid ? i0 : base) + guard;
            const float* b = s_pn + guard;
            const float sum = slot_leaf_sum(n, j, [&](int i) { return a[i] * b[i]; });
            const float sc = valid ? __builtin_fabsf(sum) : -2.0f;
            #pragma unroll
            for (int q = 0; q < 8; ++q) {                              // ascending shift, strict >
                const float v = __shfl(sc, q * 8);
                if (v > best) { best = v; best_s = s0 + q; }
            }
        }
        // despread at the chosen shift, 16 x 8 majority (:498-513)
        const int a0 = base + best_s;
        for (int i = lane; i < ES_HDR_L; i += 64) s_d[i] = s_win[a0 + i] * s_pn[i];
        __syncthreads();
        if (lane < 16) {
            const float* d = s_d + 8 * lane;
            s_sums[lane] = ((d[0] + d[1]) + (d[2] + d[3])) + ((d[4] + d[5]) + (d[6] + d[7]));
        }
        __syncthreads();
        const float sum_d = slot_leaf_sum(ES_HDR_L, j, [&](int i) { return s_d[i]; });
        const float sum_dd = slot_leaf_sum(ES_HDR_L, j, [&](int i) { return s_d[i] * s_d[i]; });
        const float mu = sum_d / (float)ES_HDR_L;
        const float sum_cc = slot_leaf_sum(ES_HDR_L, j, [&](int i) { const float c = s_d[i] - mu; return c * c; });
        const float sum_abs = slot_leaf_sum(16, j, [&](int i) { return __builtin_fabsf(s_sums[i]); });
        if (lane == 0) {
            unsigned val = 0; int npos = 0;
            for (int b = 0; b < 16; ++b) { val = (val << 1) | (s_sums[b] < 0.0f ? 1u : 0u); npos += s_sums[b] > 0.0f; }
            const float mean_abs = sum_abs / 16.0f;
            const float rms = __builtin_sqrtf(sum_dd / (float)ES_HDR_L) + (float)1e-12;
            const float margin = mean_abs / rms;
            const float sd = __builtin_sqrtf(sum_cc / (float)ES_HDR_L) + (float)1e-12;
            ok_out[rec] = (uint8_t)((npos >= 10) && (margin > 0.5f));
            val_out[rec] = (int32_t)val;
            score_out[rec] = mean_abs / sd;
            if (best_s_out) best_s_out[rec] = best_s;
        }
        __syncthreads();
    }
}

}  // namespace

int es_launch_llr(es_ctx* ctx, const double* y, int64_t B, int T, const int32_t* start,
                  const uint8_t* band, const uint8_t* pn, int variant, float* llr, int32_t* best_s,
                  float* score, hipStream_t st)
{
    long long blocks = (B + LW_WAVES - 1) / LW_WAVES;
    const long long cap = (long long)ctx->num_cu * 64 / LW_WAVES;
    if (blocks > cap) blocks = cap;
    if (ctx->max_ntaps <= ES_MAX_TAPS_FAST)
        hipLaunchKernelGGL(es_llr_wave_kernel<ES_MAX_TAPS_FAST>, dim3((unsigned)blocks), dim3(64 * LW_WAVES), 0, st, y, (long long)B, T,
                           start, band, pn, variant, ctx->d_tables, llr, best_s, score);
    else                                                                      // long matched filters (fs_target other than 48 000): 41 KB of LDS per wave
        hipLaunchKernelGGL(es_llr_wave_kernel<ES_MAX_TAPS>, dim3((unsigned)blocks), dim3(64 * LW_WAVES), 0, st, y, (long long)B, T,
                           start, band, pn, variant, ctx->d_tables, llr, best_s, score);
#ifdef ES_LLR_STAMPS
    {   // diagnostic build only: share of the phases, summed over the records of this launch
        unsigned long long h[16]; static const unsigned long long z[16] = {0};
        (void)hipDeviceSynchronize(); (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_llr_dbg), sizeof h); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_llr_dbg), z, sizeof z);
        const char* nm[9] = {"load", "matched filter", "prefix sums", "scores+candidates", "exact candidate sums", "despread+medians", "mean/var", "scale+store", "meta"};
        unsigned long long tot = 0; for (int k = 0; k < 9; ++k) tot += h[k];
        fprintf(stderr, "[llr stamps B=%lld]", (long long)B);
        for (int k = 0; k < 9; ++k) fprintf(stderr, " %s %.1f%%", nm[k], 100.0 * h[k] / (tot ? tot : 1));
        fprintf(stderr, "\n");
    }
#endif
    ES_HIP_CHECK(ctx, hipGetLastError());
    return ES_OK;
}

int es_launch_header(es_ctx* ctx, const double* y, int64_t B, int T, const int32_t* start, const uint8_t* band,
                     const uint8_t* hdr_pn, uint8_t* ok, int32_t* val, float* score, int32_t* best_s, hipStream_t st)
{
    long long blocks = B;
    const long long cap = (long long)ctx->num_cu * 32;
    if (blocks > cap) blocks = cap;
    if (ctx->max_ntaps <= ES_MAX_TAPS_FAST)
        hipLaunchKernelGGL(es_header_kernel<ES_MAX_TAPS_FAST>, dim3((unsigned)blocks), dim3(64), 0, st, y, (long long)B, T, start, band,
                           hdr_pn, ctx->d_tables, ok, val, score, best_s);
    else
        hipLaunchKernelGGL(es_header_kernel<ES_MAX_TAPS>, dim3((unsigned)blocks), dim3(64), 0, st, y, (long long)B, T, start, band,
                           hdr_pn, ctx->d_tables, ok, val, score, best_s);
    ES_HIP_CHECK(ctx, hipGetLastError());
    return ES_OK;
}
