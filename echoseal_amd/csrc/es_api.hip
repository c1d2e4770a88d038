// es_api.hip -- C ABI of libechoseal_hip.so (see include/echoseal_hip.h): context, tables,
// argument checking and dispatch to the kernel launchers.  No torch types, no global state.
#include "es_internal.h"
#include "es_exp_tab.h"

#include <cstring>
#include <new>

namespace {
std::string g_create_err;
const uint64_t kExpTab[ES_EXP_TAB_WORDS] = ES_EXP_TAB_INIT;

struct DeviceGuard {
    int prev = -1; bool ok = true;
    explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; ok = (hipSetDevice(dev) == hipSuccess); }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

int fail(es_ctx* ctx, int code, const char* msg) { ctx->err = msg; return code; }

bool capturing(hipStream_t st)
{
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
}
}  // namespace

int es_slab_enter(es_ctx* ctx, int domain, int shape, bool shareable, hipStream_t st)
{
    if (capturing(st)) return ES_OK;                            // a captured graph orders its own nodes
    es_ctx::slab_use& u = ctx->slab[domain];
    for (size_t k = 0; k < u.users.size();) {
        es_ctx::slab_use::user& w = u.users[k];
        const bool compatible = shareable && w.shareable && w.shape == shape;
        if (w.st != st && hipEventQuery(w.done) == hipSuccess) {               // that stream's last launch here has finished: forget it
            (void)hipEventDestroy(w.done);
            u.users[k] = u.users.back(); u.users.pop_back();
            continue;
        }
        // another slot geometry (or a kernel that indexes the slab by block): this launch is ordered behind that user's last one ON THE DEVICE.
        // (Also for an entry of `st` itself: normally a no-op, and right if the handle value belongs to a new stream by now.)
        if (!compatible) ES_HIP_CHECK(ctx, hipStreamWaitEvent(st, w.done, 0));
        ++k;
    }
    return ES_OK;
}

int es_slab_leave(es_ctx* ctx, int domain, int shape, bool shareable, hipStream_t st)
{
    if (capturing(st)) return ES_OK;
    es_ctx::slab_use& u = ctx->slab[domain];
    es_ctx::slab_use::user* mine = nullptr;
    for (es_ctx::slab_use::user& w : u.users) if (w.st == st) mine = &w;
    if (!mine) {
        hipEvent_t ev;
        ES_HIP_CHECK(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        u.users.push_back({st, ev, shape, shareable});
        mine = &u.users.back();
    }
    mine->shape = shape; mine->shareable = shareable;
    ES_HIP_CHECK(ctx, hipEventRecord(mine->done, st));
    return ES_OK;
}

// A launch that draws its frames from a counter gets a counter of its own.  Eager launches rotate over the first ES_CURSOR_RING -
// ES_CURSOR_CAPTURED counters (one is reused only after that many further launches of the context); a launch recorded into a stream
// capture takes one of the last ES_CURSOR_CAPTURED for good -- its graph may be replayed at any time, beside any eager launch.
int es_cursor_next(es_ctx* ctx, hipStream_t st, int** cursor)
{
    if (capturing(st)) {
        if (ctx->cursor_captured >= ES_CURSOR_CAPTURED) {
            ctx->err = "es_scl_batch: this context has recorded its 256 list-decoder launches with skip_if_hard_ok into stream captures; use another context";
            return ES_ENOMEM;
        }
        *cursor = ctx->d_cursors + (ES_CURSOR_RING - 1 - ctx->cursor_captured++);
        return ES_OK;
    }
    *cursor = ctx->d_cursors + (ctx->cursor_next++ % (ES_CURSOR_RING - ES_CURSOR_CAPTURED));
    return ES_OK;
}

extern "C" {

int es_abi_version(void) { return ES_ABI_VERSION; }

int es_info_bytes(const es_ctx* ctx) { return (ctx && ctx->n_info >= 9) ? (ctx->n_info - 8 + 7) / 8 : ES_INFO_BYTES; }

const char* es_last_error(const es_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

es_ctx* es_create(int device, int list_size_max)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_create_err = "no HIP device visible"; return nullptr; }
    if (device < 0 || device >= ndev) { g_create_err = "device index out of range"; return nullptr; }
    if (list_size_max < 0 || list_size_max > ES_MAX_LIST) { g_create_err = "list_size_max must be in [0, 256] (0: a front-end context without list-decoder scratch)"; return nullptr; }
    es_ctx* ctx = new (std::nothrow) es_ctx();
    if (!ctx) { g_create_err = "out of host memory"; return nullptr; }
    ctx->device = device;
    ctx->list_size_max = list_size_max;
    DeviceGuard g(device);
    hipDeviceProp_t prop;
    if (!g.ok || hipGetDeviceProperties(&prop, device) != hipSuccess) { g_create_err = "hipGetDeviceProperties failed"; delete ctx; return nullptr; }
    ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hipMalloc(&ctx->d_tables, sizeof(es_band_tables)) != hipSuccess ||
        hipMalloc(&ctx->d_data_pos, sizeof(uint16_t) * ES_POLAR_N) != hipSuccess ||
        hipMalloc(&ctx->d_exp_tab, sizeof(kExpTab)) != hipSuccess ||
        hipMemcpy(ctx->d_exp_tab, kExpTab, sizeof(kExpTab), hipMemcpyHostToDevice) != hipSuccess) {
        g_create_err = "device allocation failed in es_create";
        es_destroy(ctx);
        return nullptr;
    }
    if (list_size_max == 0) return ctx;                /* front-end context (band-pass, sync, demodulator, header, TX ...): no list-decoder slabs */
    /* every launch-time buffer the SCL kernel needs is allocated here, so es_scl_batch only
       enqueues work (hipGraph-capturable) */
    ctx->scl_scratch_bytes = es_scl_scratch_bytes(ctx);
    if (es_scl_multi_scratch_bytes(ctx) > ctx->scl_scratch_bytes) ctx->scl_scratch_bytes = es_scl_multi_scratch_bytes(ctx);
    if (hipMalloc(&ctx->d_scl_scratch, ctx->scl_scratch_bytes) != hipSuccess) {
        g_create_err = "device allocation of the SCL scratch slab failed";
        ctx->d_scl_scratch = nullptr;
        es_destroy(ctx);
        return nullptr;
    }
    if (hipMalloc(&ctx->d_slot_bits, 64 * sizeof(unsigned)) != hipSuccess || hipMemset(ctx->d_slot_bits, 0, 64 * sizeof(unsigned)) != hipSuccess) {
        g_create_err = "device allocation of the slab slot bitmap failed";
        es_destroy(ctx);
        return nullptr;
    }
    ctx->wide_enabled = list_size_max > 32;
    ctx->wide_scratch_bytes = es_scl_wide_scratch_bytes(ctx, &ctx->wide_slots);
    if (hipMalloc(&ctx->d_wide_slot_bits, 128 * sizeof(unsigned)) != hipSuccess || hipMemset(ctx->d_wide_slot_bits, 0, 128 * sizeof(unsigned)) != hipSuccess) {
        g_create_err = "device allocation of the slab slot bitmap failed";
        es_destroy(ctx);
        return nullptr;
    }
    if (hipMalloc(&ctx->d_cursors, ES_CURSOR_RING * sizeof(int)) != hipSuccess || hipMemset(ctx->d_cursors, 0, ES_CURSOR_RING * sizeof(int)) != hipSuccess) {
        g_create_err = "device allocation of the frame counters failed";
        es_destroy(ctx);
        return nullptr;
    }
    if (ctx->wide_scratch_bytes && hipMalloc(&ctx->d_wide_scratch, ctx->wide_scratch_bytes) != hipSuccess) {
        g_create_err = "device allocation of the wide-list SCL scratch slab failed";
        ctx->d_wide_scratch = nullptr;
        es_destroy(ctx);
        return nullptr;
    }
    return ctx;
}

void es_destroy(es_ctx* ctx)
{
    if (!ctx) return;
    DeviceGuard g(ctx->device);
    for (es_ctx::slab_use& u : ctx->slab) for (es_ctx::slab_use::user& w : u.users) (void)hipEventDestroy(w.done);
    if (ctx->d_tables) (void)hipFree(ctx->d_tables);
    if (ctx->d_data_pos) (void)hipFree(ctx->d_data_pos);
    if (ctx->d_exp_tab) (void)hipFree(ctx->d_exp_tab);
    if (ctx->d_scl_scratch) (void)hipFree(ctx->d_scl_scratch);
    if (ctx->d_slot_bits) (void)hipFree(ctx->d_slot_bits);
    if (ctx->d_ws_corr) (void)hipFree(ctx->d_ws_corr);
    if (ctx->d_wide_scratch) (void)hipFree(ctx->d_wide_scratch);
    if (ctx->d_wide_slot_bits) (void)hipFree(ctx->d_wide_slot_bits);
    if (ctx->d_sbox) (void)hipFree(ctx->d_sbox);
    if (ctx->d_hdr_pn) (void)hipFree(ctx->d_hdr_pn);
    if (ctx->d_cursors) (void)hipFree(ctx->d_cursors);
    delete ctx;
}

int es_set_tables(es_ctx* ctx, const double* ba, const double* tpl, const float* taps,
                  const int32_t* ntaps, const uint8_t* frozen)
{
    if (!ctx) return ES_EINVAL;
    if (!ba || !tpl || !taps || !ntaps || !frozen) return fail(ctx, ES_EINVAL, "es_set_tables: null table pointer");
    es_band_tables h;
    std::memset(&h, 0, sizeof h);
    for (int b = 0; b < ES_NBANDS; ++b) {
        const double a0 = ba[b * 18 + 9];
        if (a0 == 0.0) return fail(ctx, ES_EINVAL, "es_set_tables: a[0] is zero");
        for (int k = 0; k < 18; ++k) h.ba[b][k] = ba[b * 18 + k] / a0;    // SciPy normalises by a[0]
        for (int k = 0; k < ES_PRE_L; ++k) { h.tpl[b][k] = tpl[b * ES_PRE_L + k]; h.tpl32[b][k] = (float)tpl[b * ES_PRE_L + k]; }
        if (ntaps[b] < 1 || ntaps[b] > ES_MAX_TAPS) return fail(ctx, ES_EINVAL, "es_set_tables: ntaps out of range");
        h.ntaps[b] = ntaps[b];
        if (b == 0 || ntaps[b] > ctx->max_ntaps) ctx->max_ntaps = ntaps[b];
        for (int k = 0; k < ntaps[b]; ++k) h.taps[b][k] = taps[b * ES_MAX_TAPS + k];
    }
    uint16_t dpos[ES_POLAR_N];
    std::memset(dpos, 0, sizeof dpos);
    std::memset(&ctx->frozen, 0, sizeof ctx->frozen);
    int n = 0;
    for (int i = 0; i < ES_POLAR_N; ++i) {
        if (frozen[i]) ctx->frozen.w[i >> 5] |= (1u << (i & 31));
        else dpos[n++] = (uint16_t)i;
    }
    // 448 is the reference's own code (rtwm/polar_fast.py:8-9); its PolarCode class takes any K (rtwm/fastpolar.py:209-234), and so does
    // es_scl_batch (at least one information bit in front of the CRC-8).
    if (n < 9)
        return fail(ctx, ES_EINVAL, "es_set_tables: the frozen mask must leave K >= 9 information positions (448 for every entry point but es_scl_batch)");
    ctx->n_info = n;
    DeviceGuard g(ctx->device);
    ES_HIP_CHECK(ctx, hipMemcpy(ctx->d_tables, &h, sizeof h, hipMemcpyHostToDevice));
    ES_HIP_CHECK(ctx, hipMemcpy(ctx->d_data_pos, dpos, sizeof dpos, hipMemcpyHostToDevice));
    ctx->tables_ready = true;
    return ES_OK;
}

#define ES_REQUIRE_DEFAULT_CODE(ctx, who)                                                  \
    do {                                                                                   \
        if ((ctx)->n_info != ES_POLAR_K) return fail((ctx), ES_EINVAL, who ": this entry point serves the reference's own code only (448 information positions, 55-byte payloads)"); \
    } while (0)

#define ES_REQUIRE_READY(ctx)                                                              \
    do {                                                                                   \
        if (!(ctx)) return ES_EINVAL;                                                      \
        if (!(ctx)->tables_ready) return fail((ctx), ES_ENOTREADY, "es_set_tables has not been called"); \
    } while (0)

int es_bpf_batch(es_ctx* ctx, const void* frames_dev, int dtype, int64_t B, int T,
                 const uint8_t* band_dev, double* y_dev, void* stream)
{
    ES_REQUIRE_READY(ctx);
    if (B < 0 || T < 0) return fail(ctx, ES_EINVAL, "es_bpf_batch: negative size");
    if (dtype != ES_DTYPE_F32 && dtype != ES_DTYPE_I16) return fail(ctx, ES_EINVAL, "es_bpf_batch: dtype must be f32 or i16");
    if (B == 0 || T == 0) return ES_OK;
    if (!frames_dev || !band_dev || !y_dev) return fail(ctx, ES_EINVAL, "es_bpf_batch: null pointer");
    DeviceGuard g(ctx->device);
    return es_launch_bpf(ctx, frames_dev, dtype, B, T, band_dev, y_dev, nullptr, (hipStream_t)stream);
}

int es_bpf2_batch(es_ctx* ctx, const void* frames_dev, int dtype, int64_t B, int T,
                  const uint8_t* band_dev, double* y_dev, float* y32_dev, void* stream)
{
    ES_REQUIRE_READY(ctx);
    if (B < 0 || T < 0) return fail(ctx, ES_EINVAL, "es_bpf2_batch: negative size");
    if (dtype != ES_DTYPE_F32 && dtype != ES_DTYPE_I16) return fail(ctx, ES_EINVAL, "es_bpf2_batch: dtype must be f32 or i16");
    if (B == 0 || T == 0) return ES_OK;
    if (!frames_dev || !band_dev || !y_dev || !y32_dev) return fail(ctx, ES_EINVAL, "es_bpf2_batch: null pointer");
    DeviceGuard g(ctx->device);
    return es_launch_bpf(ctx, frames_dev, dtype, B, T, band_dev, y_dev, y32_dev, (hipStream_t)stream);
}

int es_xcorr32_batch(es_ctx* ctx, const float* y32_dev, int64_t B, int T, const uint8_t* band_dev,
                     float* corr32_dev, void* stream)
{
    ES_REQUIRE_READY(ctx);
    if (B < 0) return fail(ctx, ES_EINVAL, "es_xcorr32_batch: negative size");
    if (T < ES_PRE_L) return fail(ctx, ES_EINVAL, "es_xcorr32_batch: record shorter than the 63-chip template");
    if (B == 0) return ES_OK;
    if (!y32_dev || !band_dev || !corr32_dev) return fail(ctx, ES_EINVAL, "es_xcorr32_batch: null pointer");
    DeviceGuard g(ctx->device);
    return es_launch_xcorr32(ctx, y32_dev, B, T, band_dev, corr32_dev, (hipStream_t)stream);
}

int es_pick_exact_batch(es_ctx* ctx, const float* corr32_dev, const double* y_dev, int64_t B, int T,
                        const uint8_t* band_dev, double* thr_dev, int32_t* peaks_dev, int32_t* npeaks_dev,
                        uint8_t* flags_dev, void* stream)
{
    ES_REQUIRE_READY(ctx);
    if (B < 0) return fail(ctx, ES_EINVAL, "es_pick_exact_batch: negative size");
    if (T < ES_PRE_L) return fail(ctx, ES_EINVAL, "es_pick_exact_batch: record shorter than the 63-chip template");
    if (B == 0) return ES_OK;
    if (!corr32_dev || !y_dev || !band_dev || !thr_dev || !peaks_dev || !npeaks_dev || !flags_dev)
        return fail(ctx, ES_EINVAL, "es_pick_exact_batch: null pointer");
    DeviceGuard g(ctx->device);
    const int n_lags = T - (ES_PRE_L - 1);
    /* float64 workspace for the (rare) records the float32 screen cannot settle; grows monotonically,
       so after a warm-up call nothing is allocated on the launch path */
    const size_t need = (size_t)B * n_lags * sizeof(double);
    if (need > ctx->ws_corr_bytes) { const int rc0 = es_reserve(ctx, B, T); if (rc0) return rc0; }
    hipStream_t st = (hipStream_t)stream;
    int rc = es_launch_pick_exact(ctx, corr32_dev, y_dev, B, T, band_dev, thr_dev, peaks_dev, npeaks_dev, flags_dev, st);
    if (rc) return rc;
    rc = es_launch_xcorr_flagged(ctx, y_dev, B, T, band_dev, ctx->d_ws_corr, flags_dev, st);
    if (rc) return rc;
    return es_launch_pick_flagged(ctx, ctx->d_ws_corr, B, n_lags, thr_dev, peaks_dev, npeaks_dev, flags_dev, st);
}

int es_sync_fused_batch(es_ctx* ctx, const float* y32_dev, const double* y_dev, int64_t B, int T,
                        const uint8_t* band_dev, double* thr_dev, int32_t* peaks_dev, int32_t* npeaks_dev,
                        uint8_t* flags_dev, void* stream)
{
    ES_REQUIRE_READY(ctx);
    if (B < 0) return fail(ctx, ES_EINVAL, "es_sync_fused_batch: negative size");
    if (T < ES_PRE_L) return fail(ctx, ES_EINVAL, "es_sync_fused_batch: record shorter than the 63-chip template");
    if (B == 0) return ES_OK;
    if (!y32_dev || !y_dev || !band_dev || !thr_dev || !peaks_dev || !npeaks_dev || !flags_dev)
        return fail(ctx, ES_EINVAL, "es_sync_fused_batch: null pointer");
    DeviceGuard g(ctx->device);
    /* one launch: records the screen cannot settle are settled by the same wave from float64 re-evaluations (flags_dev
       then carries the reason code, for information) */
    return es_launch_sync_fused(ctx, y32_dev, y_dev, B, T, band_dev, thr_dev, peaks_dev, npeaks_dev, flags_dev, (hipStream_t)stream);
}

int es_front_batch(es_ctx* ctx, const void* frames_dev, int dtype, int64_t B, int T, const uint8_t* band_dev,
                   const uint8_t* pn_dev, const int32_t* start_dev, double* y_dev, float* y32_dev, double* thr_dev,
                   int32_t* peaks_dev, int32_t* npeaks_dev, uint8_t* flags_dev, float* llr_dev, void* stream)
{
    /* es_bpf2_batch -> es_sync_fused_batch -> es_llr_batch (variant 0) in one call: the same three launches, one trip through the binding.
       Every argument is checked before the first launch, so a bad call enqueues nothing. */
    ES_REQUIRE_READY(ctx);
    if (B < 0 || T < 0) return fail(ctx, ES_EINVAL, "es_front_batch: negative size");
    if (dtype != ES_DTYPE_F32 && dtype != ES_DTYPE_I16) return fail(ctx, ES_EINVAL, "es_front_batch: dtype must be f32 or i16");
    if (B == 0) return ES_OK;
    if (T < ES_PRE_L) return fail(ctx, ES_EINVAL, "es_front_batch: record shorter than the 63-chip template");
    if (!frames_dev || !band_dev || !pn_dev || !y_dev || !y32_dev || !thr_dev || !peaks_dev || !npeaks_dev || !flags_dev || !llr_dev)
        return fail(ctx, ES_EINVAL, "es_front_batch: null pointer");
    int rc = es_bpf2_batch(ctx, frames_dev, dtype, B, T, band_dev, y_dev, y32_dev, stream);
    if (rc != ES_OK) return rc;
    rc = es_sync_fused_batch(ctx, y32_dev, y_dev, B, T, band_dev, thr_dev, peaks_dev, npeaks_dev, flags_dev, stream);
    if (rc != ES_OK) return rc;
    return es_llr_batch(ctx, y_dev, B, T, start_dev, band_dev, pn_dev, 0, llr_dev, nullptr, nullptr, stream);
}

int es_reserve(es_ctx* ctx, int64_t B_max, int T_max)
{
    if (!ctx) return ES_EINVAL;
    if (B_max < 0 || T_max < 0) return fail(ctx, ES_EINVAL, "es_reserve: negative size");
    DeviceGuard g(ctx->device);
    const size_t need = (T_max >= ES_PRE_L) ? (size_t)B_max * (size_t)(T_max - (ES_PRE_L - 1)) * sizeof(double) : 0;
    if (need > ctx->ws_corr_bytes) {
        if (ctx->d_ws_corr) ES_HIP_CHECK(ctx, hipFree(ctx->d_ws_corr));
        ctx->d_ws_corr = nullptr; ctx->ws_corr_bytes = 0;
        if (hipMalloc(&ctx->d_ws_corr, need) != hipSuccess) return fail(ctx, ES_ENOMEM, "es_reserve: device allocation of the float64 correlation workspace failed");
        ctx->ws_corr_bytes = need;
    }
    return ES_OK;
}

int es_xcorr_batch(es_ctx* ctx, const double* y_dev, int64_t B, int T, const uint8_t* band_dev,
                   double* corr_dev, void* stream)
{
    ES_REQUIRE_READY(ctx);
    if (B < 0) return fail(ctx, ES_EINVAL, "es_xcorr_batch: negative size");
    if (T < ES_PRE_L) return fail(ctx, ES_EINVAL, "es_xcorr_batch: record shorter than the 63-chip template");
    if (B == 0) return ES_OK;
    if (!y_dev || !band_dev || !corr_dev) return fail(ctx, ES_EINVAL, "es_xcorr_batch: null pointer");
    DeviceGuard g(ctx->device);
    return es_launch_xcorr(ctx, y_dev, B, T, band_dev, corr_dev, (hipStream_t)stream);
}

int es_pick_batch(es_ctx* ctx, const double* corr_dev, int64_t B, int n_lags, double* thr_dev,
                  int32_t* peaks_dev, int32_t* npeaks_dev, void* stream)
{
    ES_REQUIRE_READY(ctx);
    if (B < 0 || n_lags < 1) return fail(ctx, ES_EINVAL, "es_pick_batch: bad size");
    if (B == 0) return ES_OK;
    if (!corr_dev || !thr_dev || !peaks_dev || !npeaks_dev) return fail(ctx, ES_EINVAL, "es_pick_batch: null pointer");
    DeviceGuard g(ctx->device);
    return es_launch_pick(ctx, corr_dev, B, n_lags, thr_dev, peaks_dev, npeaks_dev, (hipStream_t)stream);
}

int es_sync_batch(es_ctx* ctx, const void* frames_dev, int dtype, int64_t B, int T,
                  const uint8_t* band_dev, double* y_dev, double* corr_dev, double* thr_dev,
                  int32_t* peaks_dev, int32_t* npeaks_dev, void* stream)
{
    ES_REQUIRE_READY(ctx);
    if (T < ES_PRE_L) return fail(ctx, ES_EINVAL, "es_sync_batch: record shorter than the 63-chip template");
    if (B <= 0) return B == 0 ? ES_OK : fail(ctx, ES_EINVAL, "es_sync_batch: negative batch");
    DeviceGuard g(ctx->device);
    const int n_lags = T - (ES_PRE_L - 1);
    double* corr = corr_dev;
    if (!corr) {                                          // workspace grows monotonically; never freed in-call
        const size_t need = (size_t)B * n_lags * sizeof(double);
        if (need > ctx->ws_corr_bytes) { const int rc0 = es_reserve(ctx, B, T); if (rc0) return rc0; }
        corr = ctx->d_ws_corr;
    }
    int rc = es_bpf_batch(ctx, frames_dev, dtype, B, T, band_dev, y_dev, stream);
    if (rc) return rc;
    rc = es_xcorr_batch(ctx, y_dev, B, T, band_dev, corr, stream);
    if (rc) return rc;
    return es_pick_batch(ctx, corr, B, n_lags, thr_dev, peaks_dev, npeaks_dev, stream);
}

int es_llr_batch(es_ctx* ctx, const double* y_dev, int64_t B, int T, const int32_t* start_dev,
                 const uint8_t* band_dev, const uint8_t* pn_dev, int variant, float* llr_dev,
                 int32_t* best_s_dev, float* score_dev, void* stream)
{
    ES_REQUIRE_READY(ctx);
    if (B < 0 || T < 0) return fail(ctx, ES_EINVAL, "es_llr_batch: negative size");
    if (variant != 0 && variant != 1) return fail(ctx, ES_EINVAL, "es_llr_batch: variant must be 0 or 1");
    if (B == 0) return ES_OK;
    if (!y_dev || !band_dev || !pn_dev || !llr_dev) return fail(ctx, ES_EINVAL, "es_llr_batch: null pointer");
    DeviceGuard g(ctx->device);
    return es_launch_llr(ctx, y_dev, B, T, start_dev, band_dev, pn_dev, variant, llr_dev, best_s_dev,
                         score_dev, (hipStream_t)stream);
}

int es_header_batch(es_ctx* ctx, const double* y_dev, int64_t B, int T, const int32_t* start_dev,
                    const uint8_t* band_dev, const uint8_t* hdr_pn_dev, uint8_t* ok_dev, int32_t* val_dev,
                    float* score_dev, int32_t* best_s_dev, void* stream)
{
    ES_REQUIRE_READY(ctx);
    if (B < 0 || T < 0) return fail(ctx, ES_EINVAL, "es_header_batch: negative size");
    if (B == 0) return ES_OK;
    if (!y_dev || !band_dev || !hdr_pn_dev || !ok_dev || !val_dev || !score_dev)
        return fail(ctx, ES_EINVAL, "es_header_batch: null pointer");
    DeviceGuard g(ctx->device);
    return es_launch_header(ctx, y_dev, B, T, start_dev, band_dev, hdr_pn_dev, ok_dev, val_dev, score_dev,
                            best_s_dev, (hipStream_t)stream);
}

int es_scl_batch(es_ctx* ctx, const void* llr_dev, int dtype, int64_t B, int list_size,
                 int skip_if_hard_ok, uint8_t* hard_info_dev, uint8_t* hard_ok_dev,
                 uint8_t* cand_info_dev, double* cand_metric_dev, uint8_t* cand_ok_dev,
                 int32_t* ncand_dev, void* stream)
{
    ES_REQUIRE_READY(ctx);
    if (B < 0) return fail(ctx, ES_EINVAL, "es_scl_batch: negative batch");
    if (dtype != ES_DTYPE_F32 && dtype != ES_DTYPE_F64) return fail(ctx, ES_EINVAL, "es_scl_batch: dtype must be f32 or f64");
    if (ctx->list_size_max == 0) return fail(ctx, ES_EINVAL, "es_scl_batch: this is a front-end context (es_create with list_size_max = 0): it has no list-decoder scratch");
    if (list_size < 1 || list_size > ctx->list_size_max)
        return fail(ctx, ES_EINVAL, "es_scl_batch: list_size must be in [1, list_size_max]");
    if (B == 0) return ES_OK;
    if (!llr_dev || !hard_info_dev || !hard_ok_dev || !cand_info_dev || !cand_metric_dev || !cand_ok_dev || !ncand_dev)
        return fail(ctx, ES_EINVAL, "es_scl_batch: null pointer");
    DeviceGuard g(ctx->device);
    if (ctx->n_info != ES_POLAR_K && !ctx->d_wide_scratch)
        return fail(ctx, ES_EINVAL, "es_scl_batch: a code other than K = 448 runs on the lane-per-path kernel only, and this context has no scratch for it (list_size_max <= 32 and scl_lanes 1 never requested before es_reserve)");
    if (list_size > 32 || ctx->n_info != ES_POLAR_K)
        return es_launch_scl_wide(ctx, llr_dev, dtype, B, list_size, skip_if_hard_ok, hard_info_dev, hard_ok_dev,
                                  cand_info_dev, cand_metric_dev, cand_ok_dev, ncand_dev, (hipStream_t)stream);
    int lp = 1; while (lp < list_size) lp <<= 1;              // the kernels are built for powers of two; any size runs on the next one
    // Which mapping?  Measured on one MI355X (tools/mapping_sweep.py, round 3; milliseconds per launch at L = 8):
    //      B        one frame per wave   4 lanes per path   2 lanes per path   1 lane per path
    //    2 048            1.68                 2.02               2.89              3.46
    //    4 096            3.29                 2.58               3.01              3.72
    //    8 192            6.54                 4.77               4.09              4.03
    //   16 384           13.13                 8.19               8.14              6.46
    //   65 536           52.39                30.13              27.04             21.73
    // One lane per path (es_scl_wide.hip: fewest instructions per frame, but a wave carries 64/L frames through the whole decode) wins
    // once the batch yields about one such wave per SIMD -- earlier for long lists, never for L = 1; it needs that kernel's slab
    // (list_size_max > 32, or es_set_option "scl_lane_slab").
    const long long wide_waves = (long long)B * lp / 64;
    const long long wide_min = (lp <= 8 ? 1024LL : lp == 16 ? 768LL : 256LL) * ctx->num_cu / 256;      // (L = 2: 4.3 against 4.6 ms at 32 768 frames, 6.9 against 8.1 at 65 536)
    const bool lane_auto = ctx->scl_lanes == 0 && ctx->scl_multi < 0 && ctx->d_wide_scratch && lp >= 2 && wide_waves >= wide_min;
    if ((ctx->scl_lanes == 1 && ctx->scl_multi != 0) || lane_auto)
        return es_launch_scl_wide(ctx, llr_dev, dtype, B, list_size, skip_if_hard_ok, hard_info_dev, hard_ok_dev,
                                  cand_info_dev, cand_metric_dev, cand_ok_dev, ncand_dev, (hipStream_t)stream);
    if (lp <= 32) {
        // Several frames per wave (es_scl_multi.hip, 16/L frames per wave at four lanes per path) against one frame per wave: the
        // break-even is ~3 000 frames for L <= 8 (1.5 of the former's waves per SIMD at L = 8), ~1 500 frames for L = 16, ~512 for L = 32
        // (where both map one frame to a wave and the multi-frame kernel's smaller footprint -- three waves per SIMD, non-persistent
        // blocks -- wins as soon as the batch exceeds one wave per SIMD).
        const bool fits = lp <= 8 ? B >= 12LL * ctx->num_cu : B >= (lp == 16 ? 6LL : 2LL) * ctx->num_cu + (lp == 32);
        const bool multi = ctx->scl_multi == 1 || (ctx->scl_multi < 0 && fits);
        if (multi)
            return es_launch_scl_multi(ctx, llr_dev, dtype, B, list_size, skip_if_hard_ok, hard_info_dev, hard_ok_dev,
                                       cand_info_dev, cand_metric_dev, cand_ok_dev, ncand_dev, (hipStream_t)stream);
    }
    return es_launch_scl(ctx, llr_dev, dtype, B, list_size, skip_if_hard_ok, hard_info_dev, hard_ok_dev,
                         cand_info_dev, cand_metric_dev, cand_ok_dev, ncand_dev, (hipStream_t)stream);
}

int es_schedule_batch(es_ctx* ctx, const uint8_t* aes_key16_host, const uint8_t* band_key32_host, const uint32_t* ctr_dev,
                      uint32_t ctr0, int64_t n, uint8_t* pn_rows_dev, uint8_t* band_dev, void* stream)
{
    if (!ctx) return ES_EINVAL;
    if (n < 0) return fail(ctx, ES_EINVAL, "es_schedule_batch: negative count");
    if (n == 0) return ES_OK;
    if (!aes_key16_host || !band_key32_host || !pn_rows_dev || !band_dev) return fail(ctx, ES_EINVAL, "es_schedule_batch: null pointer");
    DeviceGuard g(ctx->device);
    return es_launch_schedule(ctx, aes_key16_host, band_key32_host, ctr_dev, ctr0, n, pn_rows_dev, band_dev, (hipStream_t)stream);
}

int es_tx_frames_batch(es_ctx* ctx, const uint8_t* code_dev, const uint8_t* pn_rows_dev, const uint8_t* band_dev,
                       const uint32_t* ctr_dev, const uint8_t* preamble8_host, const uint8_t* hdr_pn16_host, int64_t B,
                       double* y_ws_dev, float* frames_dev, void* stream)
{
    ES_REQUIRE_READY(ctx);
    if (B < 0) return fail(ctx, ES_EINVAL, "es_tx_frames_batch: negative batch");
    if (B == 0) return ES_OK;
    if (!code_dev || !pn_rows_dev || !band_dev || !ctr_dev || !preamble8_host || !hdr_pn16_host || !y_ws_dev || !frames_dev)
        return fail(ctx, ES_EINVAL, "es_tx_frames_batch: null pointer");
    unsigned long long pre = 0;
    for (int i = 0; i < 8; ++i) pre = (pre << 8) | preamble8_host[i];      // 63 MLS bits, MSB first, in the top 63 bits
    DeviceGuard g(ctx->device);
    return es_launch_tx_frames(ctx, code_dev, pn_rows_dev, band_dev, ctr_dev, pre, hdr_pn16_host, B, y_ws_dev, frames_dev,
                               (hipStream_t)stream);
}

int es_resample_batch(es_ctx* ctx, const void* x_dev, int dtype, int64_t B, int64_t n_in, const void* h_tf_dev, int h_per_phase,
                      int up, int down, int64_t y0, int64_t n_out, void* out_dev, void* stream)
{
    if (!ctx) return ES_EINVAL;
    if (B < 0 || n_in < 0 || n_out < 0) return fail(ctx, ES_EINVAL, "es_resample_batch: negative size");
    if (dtype != ES_DTYPE_F32 && dtype != ES_DTYPE_F64) return fail(ctx, ES_EINVAL, "es_resample_batch: dtype must be f32 or f64");
    if (up < 1 || down < 1 || h_per_phase < 1 || y0 < 0) return fail(ctx, ES_EINVAL, "es_resample_batch: bad rate / filter geometry");
    if (B == 0 || n_out == 0) return ES_OK;
    if (!x_dev || !h_tf_dev || !out_dev) return fail(ctx, ES_EINVAL, "es_resample_batch: null pointer");
    DeviceGuard g(ctx->device);
    return es_launch_resample(ctx, x_dev, dtype, B, n_in, h_tf_dev, h_per_phase, up, down, y0, n_out, out_dev, (hipStream_t)stream);
}

int es_set_option(es_ctx* ctx, const char* name, int value)
{
    if (!ctx || !name) return ES_EINVAL;
    if (std::strcmp(name, "scl_multi") == 0) {
        if (value < -1 || value > 1) return fail(ctx, ES_EINVAL, "es_set_option: scl_multi takes -1 (auto), 0 or 1");
        ctx->scl_multi = value;
        return ES_OK;
    }
    if (std::strcmp(name, "scl_prio") == 0) {
        if (value < 0 || value > 3) return fail(ctx, ES_EINVAL, "es_set_option: scl_prio takes 0..3");
        ctx->scl_prio = value;
        return ES_OK;
    }
    if (std::strcmp(name, "scl_lanes") == 0 || std::strcmp(name, "scl_lane_slab") == 0) {
        const bool slab_only = std::strcmp(name, "scl_lane_slab") == 0;
        if (slab_only ? (value != 1) : (value != 0 && value != 1 && value != 2 && value != 4))
            return fail(ctx, ES_EINVAL, slab_only ? "es_set_option: scl_lane_slab takes 1" : "es_set_option: scl_lanes takes 0 (by batch size), 1, 2 or 4");
        if (value == 1 && !ctx->d_wide_scratch) {             // one lane per path: the slab of es_scl_wide.hip (allocated here, never in an enqueue call)
            DeviceGuard g(ctx->device);
            ctx->wide_enabled = true;
            ctx->wide_scratch_bytes = es_scl_wide_scratch_bytes(ctx, &ctx->wide_slots);
            if (hipMalloc(&ctx->d_wide_scratch, ctx->wide_scratch_bytes) != hipSuccess) {
                ctx->d_wide_scratch = nullptr; ctx->wide_enabled = false;
                return fail(ctx, ES_ENOMEM, "es_set_option: device allocation of the lane-per-path scratch slab failed");
            }
        }
        if (!slab_only) ctx->scl_lanes = value;
        return ES_OK;
    }
    return fail(ctx, ES_EINVAL, "es_set_option: unknown option");
}

int es_polar_encode_batch(es_ctx* ctx, const uint8_t* info_dev, int64_t B, uint8_t* code_dev, void* stream)
{
    ES_REQUIRE_READY(ctx);
    ES_REQUIRE_DEFAULT_CODE(ctx, "es_polar_encode_batch");
    if (B < 0) return fail(ctx, ES_EINVAL, "es_polar_encode_batch: negative batch");
    if (B == 0) return ES_OK;
    if (!info_dev || !code_dev) return fail(ctx, ES_EINVAL, "es_polar_encode_batch: null pointer");
    DeviceGuard g(ctx->device);
    return es_launch_polar_encode(ctx, info_dev, B, code_dev, (hipStream_t)stream);
}

int es_softplus_batch(es_ctx* ctx, const double* t_dev, int64_t n, double* out_dev, void* stream)
{
    if (!ctx) return ES_EINVAL;
    if (n < 0) return fail(ctx, ES_EINVAL, "es_softplus_batch: negative count");
    if (n == 0) return ES_OK;
    if (!t_dev || !out_dev) return fail(ctx, ES_EINVAL, "es_softplus_batch: null pointer");
    DeviceGuard g(ctx->device);
    return es_launch_softplus(ctx, t_dev, n, out_dev, (hipStream_t)stream);
}

int es_aead_check_batch(es_ctx* ctx, const uint8_t* key32_host, const uint8_t* blobs_dev, int64_t n, int group,
                        const uint32_t* ctr_dev, uint8_t* ok_dev, uint8_t* plain_dev, void* stream)
{
    ES_REQUIRE_READY(ctx);
    if (n < 0 || group < 1) return fail(ctx, ES_EINVAL, "es_aead_check_batch: negative count or group < 1");
    if (n == 0) return ES_OK;
    if (!key32_host || !blobs_dev || !ctr_dev || !ok_dev) return fail(ctx, ES_EINVAL, "es_aead_check_batch: null pointer");
    DeviceGuard g(ctx->device);
    return es_launch_aead_check(ctx, key32_host, blobs_dev, n, group, ctr_dev, ok_dev, plain_dev, (hipStream_t)stream);
}

int es_aead_seal_batch(es_ctx* ctx, const uint8_t* key32_host, const uint8_t* nonces_dev, const uint8_t* plain_dev, int64_t n,
                       uint8_t* blobs_dev, void* stream)
{
    if (!ctx) return ES_EINVAL;
    if (n < 0) return fail(ctx, ES_EINVAL, "es_aead_seal_batch: negative count");
    if (n == 0) return ES_OK;
    if (!key32_host || !nonces_dev || !plain_dev || !blobs_dev) return fail(ctx, ES_EINVAL, "es_aead_seal_batch: null pointer");
    DeviceGuard g(ctx->device);
    return es_launch_aead_seal(ctx, key32_host, nonces_dev, plain_dev, n, blobs_dev, (hipStream_t)stream);
}

int es_select_batch(es_ctx* ctx, const uint8_t* key32_host, const uint32_t* ctr_dev, int64_t B, int L,
                    const uint8_t* hard_info_dev, const uint8_t* hard_ok_dev, const uint8_t* cand_info_dev,
                    const double* cand_metric_dev, const uint8_t* cand_ok_dev, const int32_t* ncand_dev,
                    uint8_t* payload_dev, int8_t* ok_dev, int32_t* which_dev, void* stream)
{
    ES_REQUIRE_READY(ctx);
    ES_REQUIRE_DEFAULT_CODE(ctx, "es_select_batch");
    if (B < 0 || L < 1) return fail(ctx, ES_EINVAL, "es_select_batch: negative batch or list size < 1");
    if (B == 0) return ES_OK;
    if (!hard_info_dev || !hard_ok_dev || !cand_info_dev || !cand_metric_dev || !cand_ok_dev || !ncand_dev ||
        !payload_dev || !ok_dev || !which_dev) return fail(ctx, ES_EINVAL, "es_select_batch: null pointer");
    if (key32_host && !ctr_dev) return fail(ctx, ES_EINVAL, "es_select_batch: a key needs the expected counters");
    DeviceGuard g(ctx->device);
    return es_launch_select(ctx, key32_host, ctr_dev, B, L, hard_info_dev, hard_ok_dev, cand_info_dev, cand_metric_dev,
                            cand_ok_dev, ncand_dev, payload_dev, ok_dev, which_dev, (hipStream_t)stream);
}

}  // extern "C"
