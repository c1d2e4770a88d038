/* echoseal_hip.h -- C ABI of libechoseal_hip.so, the MI355X (gfx950) implementation of the
 * EchoSeal receive hot path.
 *
 * The reference (PetarSt98/EchoSeal) is pure Python and has no FFI layer; the boundary a
 * maintainer would bind is therefore the set of NumPy/SciPy calls its detector makes on the hot
 * path.  Each entry point below names the reference code it replaces.  The Python host package
 * (echoseal_amd/, re-exported as `rtwm`) binds these with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - return value: 0 on success, negative ES_E* code on failure; es_last_error() gives text.
 *   - "dev" pointers are device (HBM) addresses owned by the caller (e.g. torch tensors);
 *     "host" pointers are ordinary host memory, copied during the call.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls only enqueue
 *     work; they never synchronise the device, so they may be captured into a hipGraph.
 *   - one es_ctx per host thread / Python object; a context is not re-entrant.
 */
#ifndef ECHOSEAL_HIP_H
#define ECHOSEAL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Version of this ABI.  2 (round 4): the tap table of es_set_tables has a row stride of ES_MAX_TAPS = 576 floats (1: 160), the
 * `nflag` parameters are gone, es_info_bytes / es_front_batch exist.  A binder must compare es_abi_version() with the ES_ABI_VERSION it
 * was written against and refuse a mismatch (echoseal_amd/_native.py:load does; the stub in INTEGRATION.md does). */
#define ES_ABI_VERSION   2

#define ES_OK            0
#define ES_EINVAL      (-1)   /* bad argument (shape, list size, null pointer) */
#define ES_ENOTREADY   (-2)   /* tables / schedule not set */
#define ES_EHIP        (-3)   /* a HIP runtime call failed */
#define ES_ENOMEM      (-4)

#define ES_FRAME_LEN   1215   /* 63 preamble + 128 header + 1024 payload chips (rtwm/detector.py:13-19) */
#define ES_PRE_L         63
#define ES_HDR_L        128
#define ES_POLAR_N     1024
#define ES_POLAR_K      448
#define ES_INFO_BYTES    55
#define ES_NBANDS         4
#define ES_MAX_TAPS     576   /* row stride of the tap table: the reference's taps are 93..131 long at fs_target = 48 000, 550 at 44 100 (rtwm/detector.py:260-294) */
#define ES_MAX_TAPS_FAST 160  /* up to here the demodulator runs its small-footprint instantiation */
#define ES_MAX_PEAKS     32   /* detector consumes at most 25 peaks per scan (rtwm/detector.py:108) */
#define ES_MAX_LIST     256   /* any list size 1..256; the mapping of paths to lanes is chosen per launch (es_set_option) */
#define ES_PN_BYTES     152   /* ceil(1215 / 8): packed PN row of one frame counter */

#define ES_DTYPE_F32      0
#define ES_DTYPE_I16      1
#define ES_DTYPE_F64      2

typedef struct es_ctx es_ctx;

/* Context: binds a device, owns table / scratch memory.  list_size_max 1..256: the largest list es_scl_batch will be asked for (sizes
 * the list decoder's scratch slabs, 0.4 GB; above 32 also the 1.6 GB lane-per-path slab).  list_size_max = 0: a FRONT-END context --
 * every entry point except es_scl_batch, no list-decoder scratch at all (what a pipeline's band-pass / sync / demodulator streams use). */
es_ctx*     es_create(int device, int list_size_max);
void        es_destroy(es_ctx* ctx);
const char* es_last_error(const es_ctx* ctx);          /* ctx may be NULL (creation errors) */
int         es_abi_version(void);                      /* == ES_ABI_VERSION of the header the library was built from */
int         es_info_bytes(const es_ctx* ctx);          /* bytes of one packed information row of es_scl_batch: 55, or ceil((K - 8) / 8) after es_set_tables with another K */

/* Static per-band tables (host pointers).
 *   ba      [4][18]  Butterworth b[9] then a[9], float64      <- rtwm/utils.py:52-55 butter_bandpass
 *   tpl     [4][63]  unit-norm cascaded preamble template      <- rtwm/detector.py:67-69
 *   taps    [4][ES_MAX_TAPS] float32 matched-filter taps, zero padded <- rtwm/detector.py:260-294 (any fs_target whose filters fit:
 *           93..131 taps at 48 000 Hz, 550 at 44 100; above ES_MAX_TAPS_FAST the demodulator runs its large-footprint instantiation)
 *   ntaps   [4]
 *   frozen  [1024]   1 = frozen bit                            <- rtwm/fastpolar.py:225-226
 * The mask leaves K information positions (information bits + CRC-8): 448 for the reference's own code (rtwm/polar_fast.py:8-9) and for
 * every entry point; es_scl_batch alone also serves any 9 <= K <= 1024 (what PolarCode(1024, K) of rtwm/fastpolar.py:209-234
 * builds), with rows of es_info_bytes(ctx) = ceil((K - 8) / 8) bytes in place of ES_INFO_BYTES: np.packbits of the K - 8
 * information bits, i.e. zero padding in the last byte when K is not a multiple of 8.                                           */
int es_set_tables(es_ctx* ctx, const double* ba, const double* tpl, const float* taps,
                  const int32_t* ntaps, const uint8_t* frozen);

/* Band-pass: y = lfilter(b, a, x.astype(float32)), zero initial state, float64 out.
 *   replaces rtwm/detector.py:59-60 (and :240-241)
 *   frames_dev [B][T] ES_DTYPE_F32 or ES_DTYPE_I16 (int16 is dequantised as x/32768, what soundfile.read hands the
 *              reference for PCM16 files, rx_app.py:26)
 *   band_dev   [B] uint8 index into the band tables
 *   y_dev      [B][T] float64                                                               */
int es_bpf_batch(es_ctx* ctx, const void* frames_dev, int dtype, int64_t B, int T,
                 const uint8_t* band_dev, double* y_dev, void* stream);

/* Normalised cross-correlation with the 63-chip template:
 *   corr[i] = sum_k y[i+k] tpl[k] / (sqrt(sum_k y[i+k]^2) + 1e-12),  i in [0, T-62)
 *   replaces rtwm/detector.py:76-79 (np.convolve + scipy.signal.correlate)
 *   corr_dev [B][T-62] float64                                                               */
int es_xcorr_batch(es_ctx* ctx, const double* y_dev, int64_t B, int T, const uint8_t* band_dev,
                   double* corr_dev, void* stream);

/* Median/MAD threshold + non-maximum suppression (+ top-5 fallback):
 *   replaces rtwm/detector.py:83-99
 *   thr_dev    [B] float64
 *   peaks_dev  [B][ES_MAX_PEAKS] int32 (ascending; fallback: descending correlation); whole rows are written, -1 = unused
 *   npeaks_dev [B] int32: number of valid entries; bit 30 set when the fallback branch ran      */
int es_pick_batch(es_ctx* ctx, const double* corr_dev, int64_t B, int n_lags, double* thr_dev,
                  int32_t* peaks_dev, int32_t* npeaks_dev, void* stream);

/* Float32 correlation screen + float64 exact fix-ups (same thr / peaks / npeaks as the float64
 * calls above, bit for bit; see echoseal_amd/csrc/es_sync32.hip):
 *   es_bpf2_batch        like es_bpf_batch, and also writes y32_dev [B][T] = (float)y
 *                        (what _llr casts to anyway, rtwm/detector.py:323,329)
 *   es_xcorr32_batch     corr32_dev [B][T-62] float32 from y32_dev: 4 860 B in + 4 612 B out per
 *                        1215-sample record -- the HBM-graded kernel
 *   es_pick_exact_batch  thr/peaks/npeaks from corr32 with float64 re-evaluation of every value
 *                        near a decision; flags_dev [B] = 1 where the record had to be redone by the
 *                        float64 kernels (done inside the call).  T - 62 <= 4096.                   */
int es_bpf2_batch(es_ctx* ctx, const void* frames_dev, int dtype, int64_t B, int T,
                  const uint8_t* band_dev, double* y_dev, float* y32_dev, void* stream);
int es_xcorr32_batch(es_ctx* ctx, const float* y32_dev, int64_t B, int T, const uint8_t* band_dev,
                     float* corr32_dev, void* stream);
int es_pick_exact_batch(es_ctx* ctx, const float* corr32_dev, const double* y_dev, int64_t B, int T,
                        const uint8_t* band_dev, double* thr_dev, int32_t* peaks_dev, int32_t* npeaks_dev,
                        uint8_t* flags_dev, void* stream);

/* The same result in ONE kernel (SURVEY.md section 8d "fused with threshold + NMS and writing only peaks"): the float32
 * correlation row of a record stays in LDS, threshold (median / MAD or the proof that it saturates at 0.95) and peaks are
 * settled from it with float64 re-evaluation of every value near a decision, and only thr / peaks / npeaks / flags reach
 * HBM: 4 T bytes of samples in (+ the few float64 samples the re-evaluations read), <= 150 bytes out per record.
 * thr / peaks / npeaks bit-identical to es_xcorr_batch + es_pick_batch.  Records the screen cannot settle (digital silence,
 * constants, exact repeats, a float32 overflow) are settled by the same wave from float64 re-evaluations alone -- slowly, with
 * no workspace and no second launch; flags_dev [B] then holds the reason code 1..5 (0 = settled from the screen), for
 * information.  T - 62 <= 4096.
 *   replaces rtwm/detector.py:76-99                                                                                  */
int es_sync_fused_batch(es_ctx* ctx, const float* y32_dev, const double* y_dev, int64_t B, int T,
                        const uint8_t* band_dev, double* thr_dev, int32_t* peaks_dev, int32_t* npeaks_dev,
                        uint8_t* flags_dev, void* stream);

/* The receive front end of one batch in one call -- es_bpf2_batch, es_sync_fused_batch and es_llr_batch (variant 0, start_dev as
 * there: NULL = 0) enqueued on `stream` in that order: what a streaming pipeline submits per batch (one trip through the
 * binding instead of three; same kernels, same results).  T - 62 <= 4096.
 *   replaces rtwm/detector.py:59-99 + 296-416 for a batch of records                                                     */
int es_front_batch(es_ctx* ctx, const void* frames_dev, int dtype, int64_t B, int T, const uint8_t* band_dev,
                   const uint8_t* pn_dev, const int32_t* start_dev, double* y_dev, float* y32_dev, double* thr_dev,
                   int32_t* peaks_dev, int32_t* npeaks_dev, uint8_t* flags_dev, float* llr_dev, void* stream);

/* Size the context's float64 correlation workspace (used by es_sync_batch without corr_dev, and by the redo pass of
 * es_pick_exact_batch) for batches of up to B_max records of T_max samples.  Allocation synchronises
 * the device: call this once, outside any stream capture; afterwards those entry points only enqueue.  Without it they
 * grow the workspace themselves the first time a larger batch arrives (same effect as calling es_reserve there).
 * es_reserve does not cover the list decoder: its slabs are allocated by es_create, and the 1.6 GB lane-per-path slab of contexts with
 * list_size_max <= 32 by es_set_option "scl_lane_slab" / "scl_lanes" = 1 -- set those before the first enqueue-only call as well.
 * Streams: the float64 workspace is shared by every call on the context (one stream at a time for the entry points that use it).  The list
 * decoder's slabs are guarded: es_scl_batch launches on one stream are ordered by the stream, launches of the same slot geometry on several
 * streams share the slab through its slot bitmap, and a launch of another geometry is ordered ON THE DEVICE behind the last launch of every
 * other stream that used the slab (hipStreamWaitEvent on an event the context records after each launch: the host does not block, the call
 * stays enqueue-only).  Stream capture: the guard is not part of a captured graph -- do not replay graphs of different slot geometry
 * (different list capacities / kernel mappings) of ONE context concurrently; launches with skip_if_hard_ok recorded into a capture keep a
 * frame counter of their own for the life of the context (at most 256 such launches per context, ES_ENOMEM beyond).                                 */
int es_reserve(es_ctx* ctx, int64_t B_max, int T_max);

/* Convenience: the three float64 calls above back to back (workspace owned by the context). */
int es_sync_batch(es_ctx* ctx, const void* frames_dev, int dtype, int64_t B, int T,
                  const uint8_t* band_dev, double* y_dev, double* corr_dev /* nullable */,
                  double* thr_dev, int32_t* peaks_dev, int32_t* npeaks_dev, void* stream);

/* Soft demodulation of the payload of one frame per record:
 *   replaces WatermarkDetector._llr (rtwm/detector.py:296-416)
 *   y_dev      [B][T] float64 band-passed records
 *   start_dev  [B] int32 frame start inside the record (frame = y[start : start+1215], may be short)
 *   pn_dev     [B][ES_PN_BYTES] packed PN bits of the frame counter (MSB first)
 *                                                       <- SecureChannel.pn_bits, rtwm/crypto.py:46-48
 *   variant    0: payload PN = bits [191, 1215); 1: bits [0, 1024)   (rtwm/detector.py:306-312)
 *   llr_dev    [B][1024] float32
 *   best_s_dev [B] int32 (nullable), score_dev [B][2] float32 best / runner-up (nullable)      */
int es_llr_batch(es_ctx* ctx, const double* y_dev, int64_t B, int T, const int32_t* start_dev,
                 const uint8_t* band_dev, const uint8_t* pn_dev, int variant, float* llr_dev,
                 int32_t* best_s_dev, float* score_dev, void* stream);

/* Header decode (16-bit counter, each bit repeated 8x, spread with the static header PN):
 *   replaces WatermarkDetector._decode_header (rtwm/detector.py:452-515)
 *   y_dev      [B][T] float64 band-passed records, start_dev [B] frame start (nullable = 0)
 *   hdr_pn_dev [B][16] packed header PN bits (pn_bits(0, 128), MSB first)
 *   ok_dev [B] uint8, val_dev [B] int32 (ctr & 0xFFFF estimate), score_dev [B] float32,
 *   best_s_dev [B] int32 (nullable)                                                          */
int es_header_batch(es_ctx* ctx, const double* y_dev, int64_t B, int T, const int32_t* start_dev,
                    const uint8_t* band_dev, const uint8_t* hdr_pn_dev, uint8_t* ok_dev, int32_t* val_dev,
                    float* score_dev, int32_t* best_s_dev, void* stream);

/* Polar(1024,448)+CRC-8 decode: hard-decision shortcut and successive-cancellation list.
 *   replaces PolarCode.decode (rtwm/fastpolar.py:254-359) up to validator selection
 *   llr_dev         [B][1024] ES_DTYPE_F32 or ES_DTYPE_F64
 *   list_size       1..256 (<= the context's list_size_max); any size, as in the reference: a size that is not a power of
 *                   two runs on the next power of two's kernel with the surplus paths switched off
 *   skip_if_hard_ok non-zero: records whose hard decision passes CRC skip the list loop
 *                   (the reference's behaviour when validator is None, fastpolar.py:268-276); the lane-per-path kernel then
 *                   fills its waves with the records that do not pass (drawn from a per-launch counter), so a batch of
 *                   mostly clean records costs what its noisy ones cost
 *   hard_info_dev   [B][55], hard_ok_dev [B]
 *   cand_info_dev   [B][L][55] candidates in ascending path-metric (stable) order
 *   cand_metric_dev [B][L] float64, cand_ok_dev [B][L] CRC flags
 *   ncand_dev       [B] int32: L, or 0 when the list loop was skipped (the record's candidate rows then read as zeros), or -1: the
 *                   kernel could not decode the record (its block found no free slot of the scratch slab -- not reachable while resident
 *                   blocks <= slots; reported, never silent: es_select_batch turns it into ok = -2 and the host layer raises) */
int es_scl_batch(es_ctx* ctx, const void* llr_dev, int dtype, int64_t B, int list_size,
                 int skip_if_hard_ok, uint8_t* hard_info_dev, uint8_t* hard_ok_dev,
                 uint8_t* cand_info_dev, double* cand_metric_dev, uint8_t* cand_ok_dev,
                 int32_t* ncand_dev, void* stream);

/* Polar encode (CRC-8 append, placement, butterfly): replaces PolarCode.encode
 * (rtwm/fastpolar.py:237-252).  info_dev [B][55] packed, code_dev [B][1024] uint8 {0,1}.      */
int es_polar_encode_batch(es_ctx* ctx, const uint8_t* info_dev, int64_t B, uint8_t* code_dev,
                          void* stream);

/* Key / PN / hop schedule on the device (schedule half of f-3): for each frame counter the 152
 * packed PN bytes of SecureChannel.pn_bits (rtwm/crypto.py:46-48 -> rtwm/utils.py:115-132: AES-128-ECB of
 * (ctr << 64 | j), j = 0..9, MSB-first bits) and the band index of choose_band (rtwm/utils.py:27-36:
 * HMAC-SHA256(band_key, ctr_be32)[0] % 4).  aes_key16_host = StreamPRNG's 16-byte sub-key, band_key32_host = the hop key
 * (the raw master key in the reference, rtwm/detector.py:31); counters are ctr_dev[i] (uint32) or, when ctr_dev is
 * NULL, ctr0 + i.  pn_rows_dev [n][152], band_dev [n]: the layout es_llr_batch / es_bpf_batch consume.            */
int es_schedule_batch(es_ctx* ctx, const uint8_t* aes_key16_host, const uint8_t* band_key32_host, const uint32_t* ctr_dev,
                      uint32_t ctr0, int64_t n, uint8_t* pn_rows_dev, uint8_t* band_dev, void* stream);

/* Frame generator: replaces WatermarkEmbedder._make_frame_chips for a batch
 * (rtwm/embedder.py:78-141): +-1 chips = 63 preamble | 128 header (lo16(ctr) MSB first, each bit 8 times, times the
 * header PN) | 1024 payload chips (code bit i times PN bit 191+i); band-pass of the frame's band with zero initial
 * state; if max|chips| + 1e-12 > 3 the frame is scaled by its reciprocal; float32 out.
 * code_dev [B][1024] {0,1} (es_polar_encode_batch), pn_rows_dev [B][152] and band_dev [B] (es_schedule_batch or the
 * broadcast schedule), ctr_dev [B] uint32; preamble8_host = np.packbits(mseq_63()) (8 bytes, last bit unused),
 * hdr_pn16_host = np.packbits(pn_bits(0, 128)); y_ws_dev = float64 workspace [B][1215]; frames_dev float32 [B][1215]. */
int es_tx_frames_batch(es_ctx* ctx, const uint8_t* code_dev, const uint8_t* pn_rows_dev, const uint8_t* band_dev,
                       const uint32_t* ctr_dev, const uint8_t* preamble8_host, const uint8_t* hdr_pn16_host, int64_t B,
                       double* y_ws_dev, float* frames_dev, void* stream);

/* Input conditioning: the polyphase FIR inside resample_to (rtwm/utils.py:58-66 =
 * scipy.signal.resample_poly(audio, up, down) -> upfirdn, zero extension).  The caller designs the filter exactly as
 * SciPy does (firwin, Kaiser 5.0, scaled by `up`, padded) and passes it in SciPy's transposed / flipped polyphase layout
 * (h_tf_dev, `up` phases of h_per_phase taps, element type = dtype); x_dev [B][n_in] and out_dev [B][n_out] have the same
 * element type (ES_DTYPE_F32 for float32 signals, ES_DTYPE_F64 otherwise: SciPy's output type).  Output k of a row is
 * sample y0 + k of the full upfirdn result (y0 = SciPy's n_pre_remove).  Bit-identical to SciPy 1.15's compiled loop. */
int es_resample_batch(es_ctx* ctx, const void* x_dev, int dtype, int64_t B, int64_t n_in, const void* h_tf_dev, int h_per_phase,
                      int up, int down, int64_t y0, int64_t n_out, void* out_dev, void* stream);

/* Diagnostic: out[i] = log1p(exp(t[i])) for t[i] <= 0 exactly as the list decoder's f / penalty evaluate it on the device
 * (np.logaddexp / np.log1p(np.exp(.)) of rtwm/fastpolar.py:18-23, 32-40 through the C library's exp and log1p) -- lets a
 * test compare the device arithmetic with the host C library bit for bit.  t_dev, out_dev float64 [n].                   */
int es_softplus_batch(es_ctx* ctx, const double* t_dev, int64_t n, double* out_dev, void* stream);

/* Tuning knobs; results never depend on them.  es_scl_batch has three mappings of list paths to lanes for lists of up to
 * 32 paths: one frame per wavefront (a path owns 64/L lanes: lowest latency, one wavefront per SIMD), several frames per
 * wavefront (a path owns 4 or 2 lanes: 16 or 32 paths per wavefront, three wavefronts per SIMD), and one lane per path (64/L
 * frames per wavefront, every lane busy at every tree depth: fewest instructions per frame, but a wavefront carries 64/L
 * frames through the whole decode, so it wants tens of thousands of frames per launch; the kernel that also serves lists of
 * 64..256 paths).  "scl_multi": -1 (default) chooses by batch size, 0 forces one frame per wavefront, 1 several.
 * "scl_lanes": lanes per path of the latter -- 4, 2, 1, or 0 (default: by batch size; 1 only when the context has that
 * kernel's scratch slab).  "scl_lane_slab" = 1 allocates that slab (1.6 GB; contexts created with list_size_max > 32 have
 * it from the start) without forcing anything: an allocation, so it belongs next to es_create / es_reserve, never between
 * enqueue calls that must not synchronise.  "scl_lanes" = 1 allocates it too.  "scl_prio" (0..3, default 0): wave priority
 * of the one-lane-per-path launches that follow -- a pipeline gives the later launch of a burst 1 so that it does not finish as
 * much later as it started (the short front-end kernels issue at 2 and 3).                                                                          */
int es_set_option(es_ctx* ctx, const char* name, int value);

/* ---- the step after the list decoder ------------------------------------------------
 * Payload validator: replaces the Python closure the reference passes to PolarCode.decode
 * (rtwm/detector.py:168-176): SecureChannel.open (rtwm/crypto.py:39-43: blob = nonce 12 | ciphertext 27 | tag 16,
 * ChaCha20-Poly1305 per RFC 8439, no AAD), plaintext starts with "ESAL", plaintext[4:8] big endian == counter.
 * blobs_dev [n][55]; blob i is checked against ctr_dev[i / group] (group = L for a [B][L][55] candidate array,
 * 1 for [B][55]); key32_host = the 32-byte AEAD key (host memory, passed by value to the kernel);
 * ok_dev [n] 1/0; plain_dev nullable [n][27] (plaintext when the tag verifies, zeros otherwise).               */
int es_aead_check_batch(es_ctx* ctx, const uint8_t* key32_host, const uint8_t* blobs_dev, int64_t n, int group,
                        const uint32_t* ctr_dev, uint8_t* ok_dev, uint8_t* plain_dev, void* stream);

/* The producer side of the same AEAD: SecureChannel.seal (rtwm/crypto.py:33-37) for 27-byte plaintexts with given
 * nonces: blobs_dev [n][55] = nonce 12 | ciphertext 27 | tag 16 (what the embedder puts into a frame, rtwm/embedder.py:153-168).
 * nonces_dev [n][12], plain_dev [n][27].                                                                            */
int es_aead_seal_batch(es_ctx* ctx, const uint8_t* key32_host, const uint8_t* nonces_dev, const uint8_t* plain_dev, int64_t n,
                       uint8_t* blobs_dev, void* stream);

/* Candidate selection: replaces the tail of PolarCode.decode (rtwm/fastpolar.py:268-276, 332-359) over the outputs
 * of es_scl_batch: the hard candidate if its CRC holds and the validator accepts it; else the first list candidate
 * (ascending metric) whose CRC holds and which the validator accepts (ok = 1); else the lowest-metric CRC-ok
 * candidate; else the lowest-metric candidate (ok = 0).  key32_host == NULL means validator=None; otherwise the
 * validator is the one of es_aead_check_batch with ctr_dev [B].  payload_dev [B][55]; ok_dev [B] int8 (1, 0, or
 * -1 when ncand is 0 although the shortcut did not return: run es_scl_batch with skip_if_hard_ok = 0 when a
 * validator is used; -2 when ncand is negative: es_scl_batch reported that it could not decode the record -- its payload row
 * is zeroed); which_dev [B] int32 (-1 = hard candidate, else list index).                              */
int es_select_batch(es_ctx* ctx, const uint8_t* key32_host, const uint32_t* ctr_dev, int64_t B, int L,
                    const uint8_t* hard_info_dev, const uint8_t* hard_ok_dev, const uint8_t* cand_info_dev,
                    const double* cand_metric_dev, const uint8_t* cand_ok_dev, const int32_t* ncand_dev,
                    uint8_t* payload_dev, int8_t* ok_dev, int32_t* which_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ECHOSEAL_HIP_H */
