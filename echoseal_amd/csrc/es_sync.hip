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

// ---------------------------------------------------------------------------------------- BPF (row)
// Few, long records (verify() on a recording: four bands x 240 000 samples): the recursion of ONE record is the whole run time, and a
// wave on its own issues one vector instruction per ~5.5 cycles whatever it is -- so what counts is the number of instructions per
// sample.  Here a record owns a DPP row of 16 lanes, lanes 0..7 of which hold one delay element each (z[k] in lane k; lanes 8..15 are
// switched off during the recursion and only help with staging): per sample every lane forms b[k+1] x, b[0] x and z + b[0] x (lane 0's
// is y), y reaches the row with ONE v_mov_b64_dpp row_newbcast:0, z[k+1] comes from the right-hand neighbour with a row_shl:1 move
// per half, then (z[k+1] + b[k+1] x) - a[k+1] y: nine vector instructions per sample against seventeen with four lanes per record.
// Lane 7 has no upper neighbour: its source lane 8 is disabled, so the move leaves lane 7's register at the -0.0 it was given at the
// start and -0.0 + t == t for every t, signed zeros included (an exact "no neighbour").  Arithmetic and its order are SciPy's
// direct-form-II-transposed loop (separate multiply and add), as in the other two kernels; bit-exactness is pinned by the same tests.
#ifndef ES_BPF_ROW_TWO_WAVES
#define ES_BPF_ROW_TWO_WAVES 1                      /* the recursion's wave does no global memory traffic: a second wave of the block loads and stores */
#endif
#ifndef ES_BPF_ROW2_UNROLL
#define ES_BPF_ROW2_UNROLL 1
#endif
#ifndef ES_BPF_ROW_FMAC
#define ES_BPF_ROW_FMAC 0                          /* 1: a[k+1] y through v_fmac_f64_dpp (same bits, same speed: measured) */
#endif
constexpr int BR_TT = 32;                           // samples per tile
constexpr int BR_RECS = 4;                          // records per wave (one per DPP row)

template <bool I16>
__global__ __launch_bounds__(64) void es_bpf_row_kernel(const void* __restrict__ frames,
        long long B, int T, const uint8_t* __restrict__ band, const es_band_tables* __restrict__ tabs,
        double* __restrict__ y, float* __restrict__ y32)
{
    __shared__ double s_x[BR_RECS][BR_TT];
    __shared__ double s_y[BR_RECS][BR_TT];
    const int lane = threadIdx.x;
    __builtin_amdgcn_s_setprio(3);
    const long long rec0 = (long long)blockIdx.x * BR_RECS;
    const int row = lane >> 4, k = lane & 15;
    const long long rec = rec0 + row;
    const bool live = rec < B;
    const int bi = live ? band[rec] : 0;
    const bool owner = k < 8;                         // holds z[k]
    const double b0 = tabs->ba[bi][0];
    const double bk = owner ? tabs->ba[bi][k + 1] : 0.0, ak = owner ? tabs->ba[bi][9 + k + 1] : 0.0;
    double z = 0.0;
    // staging: the 16 lanes of a row cover 32 samples of their record, two each
    float2 pre;
    auto fetch = [&](int t0) {
        float v0 = 0.0f, v1 = 0.0f;
        const int t = t0 + 2 * k;
        if (live) {
            if (I16) {
                const int16_t* f = (const int16_t*)frames + rec * T;
                if (t < T) v0 = (float)f[t] * (1.0f / 32768.0f);          // PCM16 as soundfile.read hands it to the reference: exact in float32
                if (t + 1 < T) v1 = (float)f[t + 1] * (1.0f / 32768.0f);
            } else {
                const float* f = (const float*)frames + rec * T;
                if (t < T) v0 = f[t];
                if (t + 1 < T) v1 = f[t + 1];
            }
        }
        pre = make_float2(v0, v1);
    };
    fetch(0);
    // the neighbour register: written by the row_shl:1 moves in lanes 0..6, never in lane 7 (its source lane is off) -> stays -0.0 there
    uint32_t nb_lo = 0u, nb_hi = 0x80000000u;
    for (int t0 = 0; t0 < T; t0 += BR_TT) {
        s_x[row][2 * k] = (double)pre.x; s_x[row][2 * k + 1] = (double)pre.y;
        wave_fence_lds();
        if (t0 + BR_TT < T) fetch(t0 + BR_TT);          // in flight while this tile is filtered
        if (owner) {
            // samples past the end of the record are zeros (their outputs are never stored): the tile is always walked in full
            #pragma unroll 1
            for (int tb = 0; tb < BR_TT; tb += 8) {
                double xs[8], ys[8];
                #pragma unroll
                for (int u = 0; u < 8; ++u) xs[u] = s_x[row][tb + u];
                #pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const double xn = xs[u];
                    const double t = xn * bk;                                       // b[k+1] x
                    const double yn = z + b0 * xn;                                  // lane 0: y = z[0] + b[0] x
#if ES_BPF_ROW_FMAC
                    // a[k+1] y with y taken from lane 0 of the row INSIDE the multiply: v_fmac_f64_dpp (the one float64 arithmetic instruction
                    // that takes a DPP operand) onto -0.0 -- fma(y, a, -0.0) is the correctly rounded product, signed zeros included -- so that
                    // the broadcast is not a link of its own in the sample-to-sample dependency chain (add -> multiply -> subtract)
                    double va = -0.0;
                    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "+v"(va) : "v"(yn), "v"(ak));
#else
                    const double yb = __builtin_amdgcn_update_dpp(yn, yn, 0x150, 0xf, 0xf, false);      // row_newbcast:0 (every active lane is written: `old` is never kept)
                    const double va = yb * ak;
#endif
                    uint64_t zu; __builtin_memcpy(&zu, &z, 8);
                    nb_lo = (uint32_t)__builtin_amdgcn_update_dpp((int)nb_lo, (int)(uint32_t)zu, 0x101, 0xf, 0xf, false);          // row_shl:1: lane k <- lane k+1
                    nb_hi = (uint32_t)__builtin_amdgcn_update_dpp((int)nb_hi, (int)(uint32_t)(zu >> 32), 0x101, 0xf, 0xf, false);
                    const uint64_t nu = ((uint64_t)nb_hi << 32) | nb_lo;
                    double z_nb; __builtin_memcpy(&z_nb, &nu, 8);
                    z = (z_nb + t) - va;                                            // z[k] = (z[k+1] + b[k+1] x) - a[k+1] y
                    ys[u] = yn;
                }
                if (k == 0) {
                    #pragma unroll
                    for (int u = 0; u < 8; ++u) s_y[row][tb + u] = ys[u];
                }
            }
        }
        wave_fence_lds();
        if (live) {
            #pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int t = t0 + 2 * k + h;
                if (t < T) {
                    const double v = s_y[row][2 * k + h];
                    y[rec * T + t] = v;
                    if (y32) y32[rec * T + t] = (float)v;
                }
            }
        }
        wave_fence_lds();
    }
}

// ---------------------------------------------------------------------------------------- BPF (row, two waves)
// The same recursion with the memory traffic taken OUT of the recursion's wave: in the one-wave form the wait for the prefetched samples
// is an `s_waitcnt vmcnt(0)` (the stores of the previous tile sit behind conditional branches, so the compiler cannot count them), i.e. once
// per 32 samples the recursion stands still until the previous tile's stores are acknowledged -- a third of the time on long records.
// Here a block is two waves: wave 1 loads tile k+1 into LDS and stores tile k-1 from LDS while wave 0 filters tile k (LDS to LDS); one
// workgroup barrier per tile.  Arithmetic, lane roles and the -0.0 neighbour of lane 7 exactly as in es_bpf_row_kernel.
template <bool I16>
__global__ __launch_bounds__(128) void es_bpf_row2_kernel(const void* __restrict__ frames,
        long long B, int T, const uint8_t* __restrict__ band, const es_band_tables* __restrict__ tabs,
        double* __restrict__ y, float* __restrict__ y32)
{
    __shared__ double s_x[2][BR_RECS][BR_TT];
    __shared__ double s_y[2][BR_RECS][BR_TT];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __builtin_amdgcn_s_setprio(3);
    const long long rec0 = (long long)blockIdx.x * BR_RECS;
    const int row = lane >> 4, k = lane & 15;
    const long long rec = rec0 + row;
    const bool live = rec < B;
    const int ntiles = (T + BR_TT - 1) / BR_TT;
    if (wv == 1) {
        // ---- the I/O wave: tile kt+1 in, tile kt-1 out, while the other wave filters tile kt
        auto fetch = [&](int t0, float& v0, float& v1) {
            v0 = 0.0f; v1 = 0.0f;
            const int t = t0 + 2 * k;
            if (live) {
                if (I16) {
                    const int16_t* f = (const int16_t*)frames + rec * T;
                    if (t < T) v0 = (float)f[t] * (1.0f / 32768.0f);          // PCM16 as soundfile.read hands it to the reference: exact in float32
                    if (t + 1 < T) v1 = (float)f[t + 1] * (1.0f / 32768.0f);
                } else {
                    const float* f = (const float*)frames + rec * T;
                    if (t < T) v0 = f[t];
                    if (t + 1 < T) v1 = f[t + 1];
                }
            }
        };
        float a0, a1;
        fetch(0, a0, a1);
        s_x[0][row][2 * k] = (double)a0; s_x[0][row][2 * k + 1] = (double)a1;
        if (ntiles > 1) fetch(BR_TT, a0, a1);
        __syncthreads();                                           // tile 0 is staged
        for (int kt = 0; kt < ntiles; ++kt) {
            if (kt + 1 < ntiles) { s_x[(kt + 1) & 1][row][2 * k] = (double)a0; s_x[(kt + 1) & 1][row][2 * k + 1] = (double)a1; }
            if (kt + 2 < ntiles) fetch((kt + 2) * BR_TT, a0, a1);
            if (kt >= 1 && live) {                                 // tile kt-1 was finished before the last barrier
                #pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int t = (kt - 1) * BR_TT + 2 * k + h;
                    if (t < T) {
                        const double v = s_y[(kt - 1) & 1][row][2 * k + h];
                        y[rec * T + t] = v;
                        if (y32) y32[rec * T + t] = (float)v;
                    }
                }
            }
            __syncthreads();                                       // tile kt filtered, tile kt+1 staged
        }
        if (live) {
            #pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int t = (ntiles - 1) * BR_TT + 2 * k + h;
                if (t < T) {
                    const double v = s_y[(ntiles - 1) & 1][row][2 * k + h];
                    y[rec * T + t] = v;
                    if (y32) y32[rec * T + t] = (float)v;
                }
            }
        }
        return;
    }
    // ---- the recursion wave
    const int bi = live ? band[rec] : 0;
    const bool owner = k < 8;                         // holds z[k]
    const double b0 = tabs->ba[bi][0];
    const double bk = owner ? tabs->ba[bi][k + 1] : 0.0, ak = owner ? tabs->ba[bi][9 + k + 1] : 0.0;
    double z = 0.0;
    uint32_t nb_lo = 0u, nb_hi = 0x80000000u;         // -0.0: what lane 7 keeps as its "neighbour" (see es_bpf_row_kernel)
    __syncthreads();                                  // tile 0 is staged
    for (int kt = 0; kt < ntiles; ++kt) {
        if (owner) {
#if ES_BPF_ROW2_UNROLL
            #pragma unroll                                          // all of the tile's LDS reads can be issued ahead of the recursion
#else
            #pragma unroll 1
#endif
            for (int tb = 0; tb < BR_TT; tb += 8) {
                double xs[8], ys[8];
                #pragma unroll
                for (int u = 0; u < 8; ++u) xs[u] = s_x[kt & 1][row][tb + u];
                #pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const double xn = xs[u];
                    const double t = xn * bk;                                       // b[k+1] x
                    const double yn = z + b0 * xn;                                  // lane 0: y = z[0] + b[0] x
                    const double yb = __builtin_amdgcn_update_dpp(yn, yn, 0x150, 0xf, 0xf, false);      // row_newbcast:0
                    uint64_t zu; __builtin_memcpy(&zu, &z, 8);
                    nb_lo = (uint32_t)__builtin_amdgcn_update_dpp((int)nb_lo, (int)(uint32_t)zu, 0x101, 0xf, 0xf, false);          // row_shl:1
                    nb_hi = (uint32_t)__builtin_amdgcn_update_dpp((int)nb_hi, (int)(uint32_t)(zu >> 32), 0x101, 0xf, 0xf, false);
                    const uint64_t nu = ((uint64_t)nb_hi << 32) | nb_lo;
                    double z_nb; __builtin_memcpy(&z_nb, &nu, 8);
                    z = (z_nb + t) - yb * ak;                                       // z[k] = (z[k+1] + b[k+1] x) - a[k+1] y
                    ys[u] = yn;
                }
                if (k == 0) {
                    #pragma unroll
                    for (int u = 0; u < 8; ++u) s_y[kt & 1][row][tb + u] = ys[u];
                }
            }
        }
        __syncthreads();                                           // tile kt filtered, tile kt+1 staged
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
        double* __restrict__ corr, const uint8_t* __restrict__ only_flagged)
{
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

// Does the threshold saturate at 0.95?  (rtwm/detector.py:83-86: thr = min(med + 4.5 * 1.4826 * MAD, 0.95).)  One pass instead of the
// ~32 of the exact order statistics: a 256-bin histogram over [-1, 1) brackets the median (bins b1..b2 of the two middle ranks) and
// bounds the MAD from below -- if at most n/2 - 1 values lie in the bins that can hold a value closer than r = m/128 to ANY median in the
// bracket (bins b1-m-1 .. b2+m+1: one bin of slack for the rounding of the bin index), both middle deviations are >= r, so
// med + 6.6717 MAD >= lo(b1) + 6.6716 r; with m chosen so that this is >= 0.95 + 1e-9 the float64 evaluation is >= 0.95 too and the
// reference's min() returns exactly 0.95.  Anything else (no proof, NaNs, medians in an edge bin, short rows) -> false: the exact path.
// Rows of a long recording (240 000 lags per band for 5 s) saturate whenever a watermark or a band-limited host is present (MAD ~0.24).
template <int NT>
__device__ bool block_threshold_saturates(const double* v, int n, uint32_t* s_hist, int* s_k)
{
    if (n < 1024) return false;                                 // (block-uniform)
    if (threadIdx.x < 256) s_hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) *s_k = 0;
    __syncthreads();
    int bad = 0;
    for (int i = threadIdx.x; i < n; i += NT) {
        const double x = v[i];
        int b;
        if (!(x >= -1.0)) { b = 0; bad |= (x != x); }
        else if (!(x < 1.0)) b = 255;
        else b = (int)__builtin_floor((x + 1.0) * 128.0);
        b = b < 0 ? 0 : (b > 255 ? 255 : b);
        atomicAdd(&s_hist[b], 1u);
    }
    if (bad) atomicOr(s_k, 2);
    __syncthreads();
    if (threadIdx.x == 0) {
        int ok = 0;
        if (*s_k == 0) {
            const int k1 = (n & 1) ? n / 2 : n / 2 - 1, k2 = n / 2;
            int acc = 0, b1 = -1, b2 = -1;
            for (int b = 0; b < 256; ++b) {
                const int c = (int)s_hist[b];
                if (b1 < 0 && k1 < acc + c) b1 = b;
                if (b2 < 0 && k2 < acc + c) b2 = b;
                acc += c;
            }
            if (b1 > 0 && b2 >= b1 && b2 < 255) {
                const double med_lo = -1.0 + (double)b1 / 128.0 - 1e-9;
                int m = (int)__builtin_ceil((0.95 + 1e-9 - med_lo) * 128.0 / 6.6716);
                if (m < 0) m = 0;
                int lo = b1 - m - 1, hi = b2 + m + 1;
                if (lo < 0) lo = 0;
                if (hi > 255) hi = 255;
                long long inside = 0;
                for (int b = lo; b <= hi; ++b) inside += s_hist[b];
                ok = (inside <= (long long)(n / 2 - 1));
            }
        }
        *s_k = ok;
    }
    __syncthreads();
    const bool r = (*s_k != 0);
    __syncthreads();
    return r;
}

template <bool IN_LDS, int NT>
__global__ __launch_bounds__(NT) void es_pick_kernel(const double* __restrict__ corr, long long B,
        int n, double* __restrict__ thr_out, int32_t* __restrict__ peaks, int32_t* __restrict__ npeaks,
        const uint8_t* __restrict__ only_flagged)
{
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
        double thr = 0.95;
        if (!block_threshold_saturates<NT>(c, n, s_hist, &s_k)) {      // (usually proven in one pass; else the exact order statistics)
            const double med = block_median<false, NT>(c, n, 0.0, s_hist, &s_pref, &s_k);
            const double mad = block_median<true, NT>(c, n, med, s_hist, &s_pref, &s_k) + 1e-12;
            thr = med + 4.5 * 1.4826 * mad;
            if (0.95 < thr) thr = 0.95;
        }

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
    // sixteen lanes per record (nine instead of seventeen vector instructions per sample and wave) while the batch cannot fill the chip with
    // the four-lanes-per-record kernel anyway: up to four such waves per SIMD
    if (B <= (long long)ctx->num_cu * 4 * 4 * BR_RECS / 4) {
        const unsigned blocks = (unsigned)((B + BR_RECS - 1) / BR_RECS);
#if ES_BPF_ROW_TWO_WAVES
        if (dtype == ES_DTYPE_I16)
            hipLaunchKernelGGL(es_bpf_row2_kernel<true>, dim3(blocks), dim3(128), 0, st, frames, (long long)B, T, band, ctx->d_tables, y, y32);
        else
            hipLaunchKernelGGL(es_bpf_row2_kernel<false>, dim3(blocks), dim3(128), 0, st, frames, (long long)B, T, band, ctx->d_tables, y, y32);
#else
        if (dtype == ES_DTYPE_I16)
            hipLaunchKernelGGL(es_bpf_row_kernel<true>, dim3(blocks), dim3(64), 0, st, frames, (long long)B, T, band, ctx->d_tables, y, y32);
        else
            hipLaunchKernelGGL(es_bpf_row_kernel<false>, dim3(blocks), dim3(64), 0, st, frames, (long long)B, T, band, ctx->d_tables, y, y32);
#endif
        ES_HIP_CHECK(ctx, hipGetLastError());
        return ES_OK;
    }
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
    return es_launch_xcorr_flagged(ctx, y, B, T, band, corr, nullptr, st);
}

int es_launch_xcorr_flagged(es_ctx* ctx, const double* y, int64_t B, int T, const uint8_t* band, double* corr,
                            const uint8_t* flags, hipStream_t st)
{
    const int n_lags = T - (ES_PRE_L - 1);
    const long long nseg = (n_lags + XC_SEG - 1) / XC_SEG;
    long long blocks = (B * nseg + XC_WAVES - 1) / XC_WAVES;
    const long long cap = (long long)ctx->num_cu * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(es_xcorr_kernel, dim3((unsigned)blocks), dim3(64 * XC_WAVES), 0, st, y, (long long)B,
                       T, band, ctx->d_tables, corr, flags);
    ES_HIP_CHECK(ctx, hipGetLastError());
    return ES_OK;
}

int es_launch_pick(es_ctx* ctx, const double* corr, int64_t B, int n_lags, double* thr, int32_t* peaks,
                   int32_t* npeaks, hipStream_t st)
{
    return es_launch_pick_flagged(ctx, corr, B, n_lags, thr, peaks, npeaks, nullptr, st);
}

int es_launch_pick_flagged(es_ctx* ctx, const double* corr, int64_t B, int n_lags, double* thr, int32_t* peaks,
                           int32_t* npeaks, const uint8_t* flags, hipStream_t st)
{
    long long blocks = B;
    const long long cap = (long long)ctx->num_cu * 16;
    if (blocks > cap) blocks = cap;
    if (n_lags <= PK_LDS_N)
        hipLaunchKernelGGL((es_pick_kernel<true, PK_THREADS>), dim3((unsigned)blocks), dim3(PK_THREADS), 0, st, corr,
                           (long long)B, n_lags, thr, peaks, npeaks, flags);
    else                                              // long rows (recordings): one block per row, so make it a big one
        hipLaunchKernelGGL((es_pick_kernel<false, 1024>), dim3((unsigned)blocks), dim3(1024), 0, st, corr,
                           (long long)B, n_lags, thr, peaks, npeaks, flags);
    ES_HIP_CHECK(ctx, hipGetLastError());
    return ES_OK;
}
