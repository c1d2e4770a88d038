/* es_internal.h -- private declarations shared by the HIP translation units. */
#ifndef ES_INTERNAL_H
#define ES_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <utility>

#include "../../include/echoseal_hip.h"

struct es_frozen_mask { uint32_t w[32]; };            /* bit i set = index i frozen */

struct es_band_tables {
    double ba[ES_NBANDS][18];                         /* b[0..8], a[0..8] (already divided by a0) */
    double tpl[ES_NBANDS][64];                        /* 63 taps + pad */
    float  taps[ES_NBANDS][ES_MAX_TAPS];
    float  tpl32[ES_NBANDS][64];                      /* template rounded to float32 (screening kernel) */
    int32_t ntaps[ES_NBANDS];
};

struct es_ctx {
    int device = 0;
    int list_size_max = 8;
    int num_cu = 256;
    bool tables_ready = false;
    std::string err;

    es_frozen_mask frozen{};
    int n_info = 0;
    int max_ntaps = 0;                /* longest matched filter of the tables (picks the demodulator's instantiation) */

    /* device-resident tables */
    es_band_tables* d_tables = nullptr;
    uint16_t* d_data_pos = nullptr;                   /* [448] ascending information indices */
    uint64_t* d_exp_tab = nullptr;                    /* [256] */
    /* scratch */
    double* d_scl_scratch = nullptr;  size_t scl_scratch_bytes = 0;
    unsigned* d_slot_bits = nullptr;  /* bitmap of the slab slots of es_scl_multi_kernel (one bit per resident block) */
    double* d_ws_corr = nullptr;      size_t ws_corr_bytes = 0;
    void*   d_wide_scratch = nullptr; size_t wide_scratch_bytes = 0; int wide_slots = 0;   /* lane-per-path list decoder (es_scl_wide.hip): list sizes 64..256, and shorter lists with scl_lanes = 1 */
    bool    wide_enabled = false;
    unsigned* d_wide_slot_bits = nullptr;                 /* its slab slot bitmap (128 words: up to 3 072 one-wave blocks) */
    uint8_t* d_sbox = nullptr;        /* AES S-box (es_schedule_batch) */
    uint8_t* d_hdr_pn = nullptr;      /* packed header PN (es_tx_frames_batch) */
    /* Who is using a scratch slab (es_slab_enter).  Domain 0 = d_scl_scratch (es_scl.hip, es_scl_multi.hip),
       domain 1 = d_wide_scratch (es_scl_wide.hip).  `shape`: how the launch cuts the slab into slots; launches of one shareable shape on
       several streams share the slab through its slot bitmap, anything else is ordered behind the outstanding launches. */
    struct slab_use {
        struct user { hipStream_t st; hipEvent_t done; int shape; bool shareable; };   /* a stream's LAST launch on this slab: recorded after it (es_slab_leave) */
        std::vector<user> users;
    };
    slab_use slab[2];
    int* d_cursors = nullptr;         /* frame counters of the lane-per-path list decoder's launches (skip_if_hard_ok): a ring, one per launch */
    unsigned cursor_next = 0;         /* eager launches rotate over the first ES_CURSOR_RING - ES_CURSOR_CAPTURED counters */
    unsigned cursor_captured = 0;     /* launches recorded into a stream capture keep a counter of their own for the life of the context (the graph may replay at any time) */
    bool pick_attr_set = false;       /* per-device kernel attributes already raised for this context's device */
    unsigned wide_attr_mask = 0;      /* bit per instantiation of the lane-per-path list decoder (its list capacity 1 .. 256) */
    /* tuning (es_set_option) */
    int scl_lanes = 0;                /* lanes per path of the multi-frame list decoder: 4 (16 paths per wave), 2 (32 paths per wave), 0 = by batch size */
    int scl_prio = 0;                 /* wave priority of the lane-per-path list decoder's launches (0..3) */
    int scl_multi = -1;               /* several frames per wave for list sizes <= 8: -1 auto (large batches), 0 never, 1 always */
};

#define ES_HIP_CHECK(ctx, expr)                                                         \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(_e);             \
            return ES_EHIP;                                                             \
        }                                                                               \
    } while (0)

#if defined(__HIPCC__)
/* Inclusive prefix sum over the 64 lanes of a wave on the data-parallel primitives (row_shr 1, 2, 4, 8 inside the rows of 16, then the two
 * row broadcasts): six dependent vector adds of a few cycles each, where six __shfl_up were six LDS-crossbar round trips (~100 cycles each).
 * Lanes without a source keep the `old` operand, 0. */
__device__ __forceinline__ uint32_t es_wave_incl_scan_u32(uint32_t x)
{
    int v = (int)x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);     /* row_shr:1 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);     /* row_shr:2 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);     /* row_shr:4 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);     /* row_shr:8 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);     /* row_bcast:15 into rows 1 and 3 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);     /* row_bcast:31 into rows 2 and 3 */
    return (uint32_t)v;
}
/* value of lane `src` (wave-uniform) in every lane: v_readlane instead of a ds_bpermute round trip */
__device__ __forceinline__ int es_wave_read_lane(int v, int src) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src)); }
#endif

static inline int es_wide_lanes_max(const es_ctx*) { return 256; }   /* lanes of the largest block of es_scl_wide.hip: the slab is sized in such blocks */
/* kernels exist for power-of-two list sizes; a context created for list_size_max serves every size up to the next one */
static inline int es_list_cap(int lmax) { int c = 1; while (c < lmax) c <<= 1; return c; }

/* Slab ownership (es_api.hip): es_slab_enter before a launch that uses a scratch slab, es_slab_leave right after it (same arguments).
 * Launches on ONE stream are ordered by the stream.  Launches of one shareable shape (the kernels that claim slots from the slab's bitmap)
 * on several streams run concurrently.  A launch of another shape -- or of a kernel that indexes the slab by block -- is ordered behind the
 * last launch of every other user ON THE DEVICE (hipStreamWaitEvent on the event es_slab_leave recorded): the host never blocks, and no
 * stream handle is used after its owner may have destroyed it (the events belong to the context).  Inside a stream capture both are no-ops:
 * a captured graph orders its own nodes, and graphs of different slot geometry must not be replayed concurrently on one context (header). */
int es_slab_enter(es_ctx* ctx, int domain, int shape, bool shareable, hipStream_t st);
int es_slab_leave(es_ctx* ctx, int domain, int shape, bool shareable, hipStream_t st);

/* launchers implemented in the kernel translation units */
size_t es_scl_scratch_bytes(const es_ctx* ctx);
size_t es_scl_wide_scratch_bytes(const es_ctx* ctx, int* slots_out);
size_t es_scl_multi_scratch_bytes(const es_ctx* ctx);
#define ES_CURSOR_RING 1024
#define ES_CURSOR_CAPTURED 256                                   /* of them: set aside for launches recorded into stream captures (never reused) */
int es_cursor_next(es_ctx* ctx, hipStream_t st, int** cursor);   /* a frame counter for one launch on `st` (es_api.hip) */
int es_launch_scl_multi(es_ctx* ctx, const void* llr, int dtype, int64_t B, int L, int skip_if_hard_ok,
                        uint8_t* hard_info, uint8_t* hard_ok, uint8_t* cand_info, double* cand_metric,
                        uint8_t* cand_ok, int32_t* ncand, hipStream_t st);
int es_launch_scl_wide(es_ctx* ctx, const void* llr, int dtype, int64_t B, int L, int skip_if_hard_ok,
                       uint8_t* hard_info, uint8_t* hard_ok, uint8_t* cand_info, double* cand_metric,
                       uint8_t* cand_ok, int32_t* ncand, hipStream_t st);
int es_launch_scl(es_ctx* ctx, const void* llr, int dtype, int64_t B, int L, int skip_if_hard_ok,
                  uint8_t* hard_info, uint8_t* hard_ok, uint8_t* cand_info, double* cand_metric,
                  uint8_t* cand_ok, int32_t* ncand, hipStream_t st);
int es_launch_softplus(es_ctx* ctx, const double* t, int64_t n, double* out, hipStream_t st);
int es_launch_polar_encode(es_ctx* ctx, const uint8_t* info, int64_t B, uint8_t* code, hipStream_t st);
int es_launch_bpf(es_ctx* ctx, const void* frames, int dtype, int64_t B, int T, const uint8_t* band,
                  double* y, float* y32, hipStream_t st);
int es_launch_xcorr32(es_ctx* ctx, const float* y32, int64_t B, int T, const uint8_t* band, float* corr32, hipStream_t st);
int es_launch_pick_exact(es_ctx* ctx, const float* corr32, const double* y, int64_t B, int T, const uint8_t* band,
                         double* thr, int32_t* peaks, int32_t* npeaks, uint8_t* flags, hipStream_t st);
int es_launch_sync_fused(es_ctx* ctx, const float* y32, const double* y, int64_t B, int T, const uint8_t* band, double* thr,
                         int32_t* peaks, int32_t* npeaks, uint8_t* flags, hipStream_t st);
/* redo of flagged records by the float64 kernels */
int es_launch_xcorr_flagged(es_ctx* ctx, const double* y, int64_t B, int T, const uint8_t* band, double* corr,
                            const uint8_t* flags, hipStream_t st);
int es_launch_pick_flagged(es_ctx* ctx, const double* corr, int64_t B, int n_lags, double* thr, int32_t* peaks,
                           int32_t* npeaks, const uint8_t* flags, hipStream_t st);
int es_launch_xcorr(es_ctx* ctx, const double* y, int64_t B, int T, const uint8_t* band, double* corr,
                    hipStream_t st);
int es_launch_pick(es_ctx* ctx, const double* corr, int64_t B, int n_lags, double* thr, int32_t* peaks,
                   int32_t* npeaks, hipStream_t st);
int es_launch_llr(es_ctx* ctx, const double* y, int64_t B, int T, const int32_t* start,
                  const uint8_t* band, const uint8_t* pn, int variant, float* llr, int32_t* best_s,
                  float* score, hipStream_t st);

int es_launch_tx_frames(es_ctx* ctx, const uint8_t* code, const uint8_t* pn_rows, const uint8_t* band, const uint32_t* ctr,
                        unsigned long long pre_bits, const uint8_t* hdr_pn16, int64_t B, double* y_ws, float* frames, hipStream_t st);
int es_launch_resample(es_ctx* ctx, const void* x, int dtype, int64_t B, int64_t n_x, const void* h_tf, int hpp, int up, int down,
                       int64_t y0, int64_t n_out, void* out, hipStream_t st);
int es_launch_schedule(es_ctx* ctx, const uint8_t* aes_key16, const uint8_t* band_key32, const uint32_t* ctr_dev,
                       uint32_t ctr0, int64_t n, uint8_t* pn_rows, uint8_t* band, hipStream_t st);
int es_launch_aead_seal(es_ctx* ctx, const uint8_t* key32, const uint8_t* nonces, const uint8_t* plain, int64_t n, uint8_t* blobs,
                        hipStream_t st);
int es_launch_aead_check(es_ctx* ctx, const uint8_t* key32, const uint8_t* blobs, int64_t n, int group, const uint32_t* ctr,
                         uint8_t* ok, uint8_t* plain, hipStream_t st);
int es_launch_select(es_ctx* ctx, const uint8_t* key32, const uint32_t* ctr, int64_t B, int L, const uint8_t* hard_info,
                     const uint8_t* hard_ok, const uint8_t* cand_info, const double* cand_metric, const uint8_t* cand_ok,
                     const int32_t* ncand, uint8_t* payload, int8_t* ok, int32_t* which, hipStream_t st);
int es_launch_header(es_ctx* ctx, const double* y, int64_t B, int T, const int32_t* start, const uint8_t* band,
                     const uint8_t* hdr_pn, uint8_t* ok, int32_t* val, float* score, int32_t* best_s, hipStream_t st);

#endif
