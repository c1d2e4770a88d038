"""RxEngine: batched EchoSeal receive hot path on one MI355X.

Thin host layer over the C ABI (include/echoseal_hip.h).  torch is used only to own device
buffers and to name the HIP stream; every computation is a hand-written HIP kernel.  Each method
mirrors one stage of the reference detector:

    bpf / xcorr / pick / sync   rtwm/detector.py:59-99   (band-pass, NCC, threshold, NMS)
    llr                         rtwm/detector.py:296-416 (_llr)
    scl                         rtwm/fastpolar.py:254-359 (PolarCode.decode, pre-validator)
    polar_encode                rtwm/fastpolar.py:237-252

`decode_batch` strings them together for the throughput metric: one frame record ->
sync + LLR (variant 0, known start/counter) + SCL-L.
"""
from __future__ import annotations

import os
import sys
import warnings
from dataclasses import dataclass

import numpy as np
import torch

from . import _native as nat
from .tables import pack_tables

# The streaming pipelines run on up to eight HIP streams.  The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues
# (default 4) and streams that share a queue serialise: with the default, every arrangement with more than four streams is slower,
# silently (a third of the grouped pipeline's rate).  The variable is read when HIP initialises, so it is set here when that has not
# happened yet; DecodePipeline warns when more streams are alive than the budget this module could establish.
HW_QUEUES_WANTED = 8
if "GPU_MAX_HW_QUEUES" not in os.environ and not torch.cuda.is_initialized():
    os.environ["GPU_MAX_HW_QUEUES"] = str(HW_QUEUES_WANTED)
_HWQ_ENV_AT_IMPORT = os.environ.get("GPU_MAX_HW_QUEUES")      # None: HIP was already up, without the variable
_LIVE_STREAMS = [0]                                            # streams made through pipeline_streams and not yet given back (a plain count:
                                                               # weak references to torch.cuda.Stream objects crash the interpreter's final GC)


def hw_queue_budget() -> int:
    """Hardware queues the process's HIP streams are spread over, as far as this module can tell: GPU_MAX_HW_QUEUES as it stood
    when the module was imported (set by the module itself if HIP had not started), else the runtime's default of 4."""
    try:
        return max(1, int(_HWQ_ENV_AT_IMPORT)) if _HWQ_ENV_AT_IMPORT is not None else 4
    except ValueError:
        return 4


def pipeline_streams(device, n: int, priority: int = 0) -> list:
    """n new HIP streams, counted against the hardware-queue budget (DecodePipeline makes its own through this; a process that builds
    several pipelines makes the streams once and hands them to each: DecodePipeline(streams=...))."""
    out = [torch.cuda.Stream(device, priority=priority) for _ in range(n)]
    _LIVE_STREAMS[0] += n
    if _LIVE_STREAMS[0] > hw_queue_budget():
        import gc
        gc.collect()                                # (pipelines that are garbage but not yet collected still count their streams)
    if _LIVE_STREAMS[0] > hw_queue_budget():
        warnings.warn(f"DecodePipeline: {_LIVE_STREAMS[0]} pipeline streams are alive but the HIP runtime has {hw_queue_budget()} hardware queues "
                      f"(GPU_MAX_HW_QUEUES{'=' + _HWQ_ENV_AT_IMPORT if _HWQ_ENV_AT_IMPORT else ' was not set before HIP initialised'}): streams that share a "
                      "queue serialise.  Set GPU_MAX_HW_QUEUES=8 before the first GPU call (importing echoseal_amd.engine first does it), "
                      "and hand existing streams to further pipelines (streams=...).", RuntimeWarning, stacklevel=3)
    return out


def _ptr(t: torch.Tensor | None) -> int | None:
    return None if t is None else t.data_ptr()


@dataclass
class SclResult:
    hard_info: torch.Tensor    # [B,55] uint8 (engines of another code: [B, ceil((K - 8) / 8)])
    hard_ok: torch.Tensor      # [B] uint8
    cand_info: torch.Tensor    # [B,L,55] uint8, ascending metric
    cand_metric: torch.Tensor  # [B,L] float64
    cand_ok: torch.Tensor      # [B,L] uint8
    ncand: torch.Tensor        # [B] int32 (0 = list loop skipped; < 0 = the kernel could not decode the record: see check())

    def check(self) -> "SclResult":
        """Raise if the list decoder reported records it could not decode (ncand < 0: a block found no free slot of the scratch
        slab -- cannot happen while resident blocks <= slots, and must never pass silently).  Synchronises on the result."""
        bad = int((self.ncand < 0).sum().item())
        if bad:
            raise nat.NativeError(f"es_scl_batch: {bad} record(s) were not decoded (no free scratch-slab slot); the candidate rows of those records are undefined")
        return self


@dataclass
class SyncResult:
    y: torch.Tensor            # [B,T] float64 band-passed records
    corr: torch.Tensor | None  # [B,T-62] float64
    thr: torch.Tensor          # [B] float64
    peaks: torch.Tensor        # [B,32] int32
    npeaks: torch.Tensor       # [B] int32 (count; bit 30 = fallback branch)
    corr32: torch.Tensor | None = None   # float32 screen (sync_fast only)
    flags: torch.Tensor | None = None    # records redone in float64 (sync_fast only)
    y32: torch.Tensor | None = None


class RxEngine:
    def __init__(self, device: int | torch.device = 0, *, list_size_max: int = 32, fs: int = 48_000, code_k: int = 448):
        """code_k: information positions of the polar code (data bits + CRC-8).  448 is the reference's own code (rtwm/polar_fast.py:8-9);
        any other 9 <= K <= 1024 (PolarCode(1024, K), rtwm/fastpolar.py:209-234) makes an engine whose `scl` is the only FEC entry point
        (rows of ceil((K - 8) / 8) bytes = np.packbits of the information bits), on the lane-per-path kernel: it needs list_size_max > 32
        or the "scl_lane_slab" option.
        list_size_max: the largest list `scl` will be asked for (sizes the list decoder's scratch: 0.4 GB, above 32 another 1.6 GB);
        0 = a front-end engine (everything but `scl`, no list-decoder scratch): what a pipeline's band-pass / sync / demodulator streams use."""
        if not torch.cuda.is_available():
            raise nat.NativeError("RxEngine needs a ROCm GPU: torch.cuda.is_available() is False")
        self.device = torch.device("cuda", device if isinstance(device, int) else (device.index or 0))
        self._lib = nat.load()
        self._ctx = self._lib.es_create(self.device.index, int(list_size_max))
        if not self._ctx:
            raise nat.NativeError("es_create failed: " + (self._lib.es_last_error(None) or b"?").decode())
        self.list_size_max = int(list_size_max)
        self.fs = fs
        self.code_k = int(code_k)
        ba, tpl, taps, ntaps, frozen = pack_tables(fs, self.code_k)
        self._tables = (ba, tpl, taps, ntaps, frozen)       # keep host arrays alive
        nat.check(self._ctx, self._lib.es_set_tables(
            self._ctx, ba.ctypes.data, tpl.ctypes.data, taps.ctypes.data, ntaps.ctypes.data,
            frozen.ctypes.data), "es_set_tables")
        self.info_bytes = int(self._lib.es_info_bytes(self._ctx))

    def set_option(self, name: str, value: int) -> None:
        """Tuning knobs of the native library (results never depend on them); see include/echoseal_hip.h."""
        nat.check(self._ctx, self._lib.es_set_option(self._ctx, name.encode(), int(value)), "es_set_option")

    def close(self) -> None:
        if getattr(self, "_ctx", None):
            self._lib.es_destroy(self._ctx)
            self._ctx = None

    def __del__(self):  # pragma: no cover
        try:
            if not sys.is_finalizing():            # at interpreter shutdown the HIP runtime may already be gone; the OS reclaims the rest
                self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def _dev(self, x, dtype) -> torch.Tensor:
        t = torch.as_tensor(x)
        if t.dtype != dtype:
            t = t.to(dtype)
        return t.to(self.device, non_blocking=True).contiguous()

    # ------------------------------------------------------------------ sync stage
    def bpf(self, frames: torch.Tensor, band: torch.Tensor) -> torch.Tensor:
        if frames.dim() != 2:
            raise ValueError("frames must be [B, T]")
        if frames.dtype == torch.int16:
            dt = nat.ES_DTYPE_I16
        elif frames.dtype == torch.float32:
            dt = nat.ES_DTYPE_F32
        else:
            raise ValueError("frames must be float32 or int16")
        frames = frames.contiguous()
        B, T = frames.shape
        y = torch.empty((B, T), dtype=torch.float64, device=self.device)
        nat.check(self._ctx, self._lib.es_bpf_batch(self._ctx, _ptr(frames), dt, B, T, _ptr(band), _ptr(y),
                                                    self._stream()), "es_bpf_batch")
        return y

    def bpf2(self, frames: torch.Tensor, band: torch.Tensor):
        """Band-pass with both outputs: y (float64) and y32 = (float)y."""
        if frames.dim() != 2:
            raise ValueError("frames must be [B, T]")
        if frames.dtype == torch.int16:
            dt = nat.ES_DTYPE_I16
        elif frames.dtype == torch.float32:
            dt = nat.ES_DTYPE_F32
        else:
            raise ValueError("frames must be float32 or int16")
        frames = frames.contiguous()
        B, T = frames.shape
        y = torch.empty((B, T), dtype=torch.float64, device=self.device)
        y32 = torch.empty((B, T), dtype=torch.float32, device=self.device)
        nat.check(self._ctx, self._lib.es_bpf2_batch(self._ctx, _ptr(frames), dt, B, T, _ptr(band), _ptr(y), _ptr(y32),
                                                     self._stream()), "es_bpf2_batch")
        return y, y32

    def xcorr32(self, y32: torch.Tensor, band: torch.Tensor) -> torch.Tensor:
        B, T = y32.shape
        corr = torch.empty((B, T - 62), dtype=torch.float32, device=self.device)
        nat.check(self._ctx, self._lib.es_xcorr32_batch(self._ctx, _ptr(y32), B, T, _ptr(band), _ptr(corr),
                                                        self._stream()), "es_xcorr32_batch")
        return corr

    def pick_exact(self, corr32: torch.Tensor, y: torch.Tensor, band: torch.Tensor):
        """thr / peaks / npeaks identical to pick(xcorr(y)); also returns the per-record redo flags."""
        B, T = y.shape
        thr = torch.empty(B, dtype=torch.float64, device=self.device)
        peaks = torch.empty((B, nat.ES_MAX_PEAKS), dtype=torch.int32, device=self.device)     # the kernels write whole rows (-1 = unused)
        npeaks = torch.empty(B, dtype=torch.int32, device=self.device)
        flags = torch.empty(B, dtype=torch.uint8, device=self.device)
        nat.check(self._ctx, self._lib.es_pick_exact_batch(self._ctx, _ptr(corr32), _ptr(y), B, T, _ptr(band), _ptr(thr),
                                                           _ptr(peaks), _ptr(npeaks), _ptr(flags), self._stream()),
                  "es_pick_exact_batch")
        return thr, peaks, npeaks, flags

    def sync_fused(self, y: torch.Tensor, y32: torch.Tensor, band: torch.Tensor):
        """Correlation screen + exact threshold / peak picking in ONE kernel (the screen row never leaves LDS):
        -> (thr, peaks, npeaks, flags), identical to pick(xcorr(y))."""
        B, T = y.shape
        thr = torch.empty(B, dtype=torch.float64, device=self.device)
        peaks = torch.empty((B, nat.ES_MAX_PEAKS), dtype=torch.int32, device=self.device)
        npeaks = torch.empty(B, dtype=torch.int32, device=self.device)
        flags = torch.empty(B, dtype=torch.uint8, device=self.device)
        nat.check(self._ctx, self._lib.es_sync_fused_batch(self._ctx, _ptr(y32), _ptr(y), B, T, _ptr(band), _ptr(thr), _ptr(peaks),
                                                           _ptr(npeaks), _ptr(flags), self._stream()), "es_sync_fused_batch")
        return thr, peaks, npeaks, flags

    def front(self, frames: torch.Tensor, band: torch.Tensor, pn_rows: torch.Tensor, *, start: torch.Tensor | None = None,
              out: torch.Tensor | None = None):
        """bpf2 -> sync_fused -> llr (variant 0) in one library call (es_front_batch): -> (y, thr, peaks, npeaks, flags, llr)."""
        if frames.dim() != 2 or frames.dtype not in (torch.float32, torch.int16):
            raise ValueError("frames must be float32 or int16 [B, T]")
        frames = frames.contiguous()
        B, T = frames.shape
        dev = self.device
        y = torch.empty((B, T), dtype=torch.float64, device=dev)
        y32 = torch.empty((B, T), dtype=torch.float32, device=dev)
        thr = torch.empty(B, dtype=torch.float64, device=dev)
        peaks = torch.empty((B, nat.ES_MAX_PEAKS), dtype=torch.int32, device=dev)
        npeaks = torch.empty(B, dtype=torch.int32, device=dev)
        flags = torch.empty(B, dtype=torch.uint8, device=dev)
        if out is None:
            out = torch.empty((B, 1024), dtype=torch.float32, device=dev)
        elif out.shape != (B, 1024) or out.dtype != torch.float32 or not out.is_contiguous():
            raise ValueError("out must be a contiguous float32 [B, 1024] tensor")
        nat.check(self._ctx, self._lib.es_front_batch(self._ctx, _ptr(frames), nat.ES_DTYPE_I16 if frames.dtype == torch.int16 else nat.ES_DTYPE_F32,
                                                      B, T, _ptr(band), _ptr(pn_rows), _ptr(start), _ptr(y), _ptr(y32), _ptr(thr), _ptr(peaks),
                                                      _ptr(npeaks), _ptr(flags), _ptr(out), self._stream()), "es_front_batch")
        return y, thr, peaks, npeaks, flags, out

    def reserve(self, B_max: int, T_max: int) -> None:
        """Size the context's workspaces once (es_reserve): afterwards the sync entry points only enqueue."""
        nat.check(self._ctx, self._lib.es_reserve(self._ctx, int(B_max), int(T_max)), "es_reserve")

    FAST_MAX_LAGS = 4096

    def sync_fast(self, frames: torch.Tensor, band: torch.Tensor, *, fused: bool = True) -> SyncResult:
        """Band-pass + float32 correlation screen + exact peak picking (results identical to sync()).  fused: screen and
        picking in one kernel (the default); otherwise es_xcorr32_batch -> es_pick_exact_batch with the screen in HBM."""
        y, y32 = self.bpf2(frames, band)
        if fused:
            corr32 = None
            thr, peaks, npeaks, flags = self.sync_fused(y, y32, band)
        else:
            corr32 = self.xcorr32(y32, band)
            thr, peaks, npeaks, flags = self.pick_exact(corr32, y, band)
        res = SyncResult(y, None, thr, peaks, npeaks)
        res.corr32, res.flags, res.y32 = corr32, flags, y32
        return res

    def xcorr(self, y: torch.Tensor, band: torch.Tensor) -> torch.Tensor:
        B, T = y.shape
        corr = torch.empty((B, T - 62), dtype=torch.float64, device=self.device)
        nat.check(self._ctx, self._lib.es_xcorr_batch(self._ctx, _ptr(y), B, T, _ptr(band), _ptr(corr),
                                                      self._stream()), "es_xcorr_batch")
        return corr

    def pick(self, corr: torch.Tensor):
        B, n = corr.shape
        thr = torch.empty(B, dtype=torch.float64, device=self.device)
        peaks = torch.empty((B, nat.ES_MAX_PEAKS), dtype=torch.int32, device=self.device)     # the kernels write whole rows (-1 = unused)
        npeaks = torch.empty(B, dtype=torch.int32, device=self.device)
        nat.check(self._ctx, self._lib.es_pick_batch(self._ctx, _ptr(corr), B, n, _ptr(thr), _ptr(peaks),
                                                     _ptr(npeaks), self._stream()), "es_pick_batch")
        return thr, peaks, npeaks

    def sync(self, frames: torch.Tensor, band: torch.Tensor, *, keep_corr: bool = True) -> SyncResult:
        y = self.bpf(frames, band)
        corr = self.xcorr(y, band)
        thr, peaks, npeaks = self.pick(corr)
        return SyncResult(y, corr if keep_corr else None, thr, peaks, npeaks)

    # ------------------------------------------------------------------ soft demod
    def llr(self, y: torch.Tensor, band: torch.Tensor, pn_rows: torch.Tensor, *, start: torch.Tensor | None = None,
            variant: int = 0, want_diag: bool = False, out: torch.Tensor | None = None):
        B, T = y.shape
        if out is None:
            out = torch.empty((B, 1024), dtype=torch.float32, device=self.device)
        elif out.shape != (B, 1024) or out.dtype != torch.float32 or not out.is_contiguous():
            raise ValueError("out must be a contiguous float32 [B, 1024] tensor")
        best_s = torch.empty(B, dtype=torch.int32, device=self.device) if want_diag else None
        score = torch.empty((B, 2), dtype=torch.float32, device=self.device) if want_diag else None
        nat.check(self._ctx, self._lib.es_llr_batch(self._ctx, _ptr(y), B, T, _ptr(start), _ptr(band), _ptr(pn_rows),
                                                    int(variant), _ptr(out), _ptr(best_s), _ptr(score),
                                                    self._stream()), "es_llr_batch")
        return (out, best_s, score) if want_diag else out

    def header(self, y: torch.Tensor, band: torch.Tensor, hdr_pn: torch.Tensor, *, start: torch.Tensor | None = None):
        """Batched _decode_header (rtwm/detector.py:452-515) -> (ok uint8[B], val int32[B], score float32[B])."""
        B, T = y.shape
        if hdr_pn.shape[0] == 1 and B > 1:
            hdr_pn = hdr_pn.expand(B, 16)
        hdr_pn = hdr_pn.contiguous()
        ok = torch.empty(B, dtype=torch.uint8, device=self.device)
        val = torch.empty(B, dtype=torch.int32, device=self.device)
        score = torch.empty(B, dtype=torch.float32, device=self.device)
        nat.check(self._ctx, self._lib.es_header_batch(self._ctx, _ptr(y), B, T, _ptr(start), _ptr(band), _ptr(hdr_pn),
                                                       _ptr(ok), _ptr(val), _ptr(score), None, self._stream()),
                  "es_header_batch")
        return ok, val, score

    # ------------------------------------------------------------------ FEC
    def scl(self, llr: torch.Tensor, *, list_size: int = 8, skip_if_hard_ok: bool = True) -> SclResult:
        if llr.dim() != 2 or llr.shape[1] != 1024:
            raise ValueError("llr must be [B, 1024]")
        if llr.dtype == torch.float32:
            dt = nat.ES_DTYPE_F32
        elif llr.dtype == torch.float64:
            dt = nat.ES_DTYPE_F64
        else:
            raise ValueError("llr must be float32 or float64")
        llr = llr.contiguous()
        B, L = llr.shape[0], int(list_size)
        dev = self.device
        res = SclResult(
            torch.empty((B, self.info_bytes), dtype=torch.uint8, device=dev), torch.empty(B, dtype=torch.uint8, device=dev),
            torch.empty((B, L, self.info_bytes), dtype=torch.uint8, device=dev), torch.empty((B, L), dtype=torch.float64, device=dev),
            torch.empty((B, L), dtype=torch.uint8, device=dev), torch.empty(B, dtype=torch.int32, device=dev))   # every row is written by the kernel
        nat.check(self._ctx, self._lib.es_scl_batch(
            self._ctx, _ptr(llr), dt, B, L, int(bool(skip_if_hard_ok)), _ptr(res.hard_info), _ptr(res.hard_ok),
            _ptr(res.cand_info), _ptr(res.cand_metric), _ptr(res.cand_ok), _ptr(res.ncand), self._stream()),
            "es_scl_batch")
        return res

    def softplus(self, t: torch.Tensor) -> torch.Tensor:
        """Diagnostic: log1p(exp(t)) for t <= 0 as the list decoder evaluates it on the device (float64 in, float64 out)."""
        t = t.contiguous()
        out = torch.empty_like(t)
        nat.check(self._ctx, self._lib.es_softplus_batch(self._ctx, _ptr(t), t.numel(), _ptr(out), self._stream()), "es_softplus_batch")
        return out

    def polar_encode(self, info: torch.Tensor) -> torch.Tensor:
        info = info.contiguous()
        B = info.shape[0]
        code = torch.empty((B, 1024), dtype=torch.uint8, device=self.device)
        nat.check(self._ctx, self._lib.es_polar_encode_batch(self._ctx, _ptr(info), B, _ptr(code), self._stream()),
                  "es_polar_encode_batch")
        return code

    # ------------------------------------------------------------------ input conditioning
    def resample(self, audio, fs_orig: int, fs_target: int) -> torch.Tensor:
        """resample_to (rtwm/utils.py:58-66) on the device: 1-D or [B,n] signal -> the same values
        scipy.signal.resample_poly(audio, fs_target/g, fs_orig/g) returns (float32 for float32 input, else float64),
        bit for bit.  Equal rates return the input as a device tensor."""
        from .utils import resample_plan
        x = audio if torch.is_tensor(audio) else torch.as_tensor(np.asarray(audio))
        one_d = x.dim() == 1
        if one_d:
            x = x.reshape(1, -1)
        np_dtype = np.dtype(str(x.dtype).replace("torch.", ""))
        plan = resample_plan(x.shape[1], fs_target, fs_orig, np_dtype)
        if plan is None:
            out = x.to(self.device)
            return out[0] if one_d else out
        h_tf, hpp, up, down, y0, n_out, ctype = plan
        tdt = torch.float32 if ctype == np.float32 else torch.float64
        xd = self._dev(x, tdt)
        hd = torch.from_numpy(h_tf).to(self.device)
        out = torch.empty((x.shape[0], n_out), dtype=tdt, device=self.device)
        nat.check(self._ctx, self._lib.es_resample_batch(self._ctx, _ptr(xd), nat.ES_DTYPE_F32 if tdt == torch.float32 else nat.ES_DTYPE_F64,
                                                         x.shape[0], x.shape[1], _ptr(hd), hpp, up, down, y0, n_out, _ptr(out),
                                                         self._stream()), "es_resample_batch")
        return out[0] if one_d else out

    # ------------------------------------------------------------------ key / PN / hop schedule
    def schedule(self, aes_key16: bytes, band_key32: bytes, ctrs=None, *, ctr0: int = 0, n: int | None = None):
        """PN rows and band indices of frame counters, derived on the device: -> (pn [n,152] uint8, band [n] uint8).
        `ctrs` (any integer sequence / tensor) or the range ctr0 .. ctr0+n-1.  aes_key16 = StreamPRNG.sub_key,
        band_key32 = the hop key (rtwm/utils.py:27-36, 115-132)."""
        if len(aes_key16) != 16 or len(band_key32) != 32:
            raise ValueError("schedule needs a 16-byte AES key and a 32-byte band key")
        if ctrs is not None:
            cd = self._ctr_dev(ctrs); n = cd.numel()
        elif n is None:
            raise ValueError("give ctrs or (ctr0, n)")
        else:
            cd = None
        pn = torch.empty((n, 152), dtype=torch.uint8, device=self.device)
        band = torch.empty(n, dtype=torch.uint8, device=self.device)
        nat.check(self._ctx, self._lib.es_schedule_batch(self._ctx, bytes(aes_key16), bytes(band_key32), _ptr(cd),
                                                         int(ctr0) & 0xFFFFFFFF, n, _ptr(pn), _ptr(band), self._stream()),
                  "es_schedule_batch")
        return pn, band

    def make_frames(self, sec, band_key32: bytes, ctrs, payloads: torch.Tensor) -> torch.Tensor:
        """Batch of transmitted frames on the device (WatermarkEmbedder.make_frames, rtwm/embedder.py:78-141):
        payloads uint8 [B,55] (already sealed) under frame counters `ctrs` -> float32 [B,1215].  `sec` is the
        SecureChannel (PN sub-key, header PN), band_key32 the hop key."""
        from .utils import mseq_63
        cd = self._ctr_dev(ctrs)
        B = cd.numel()
        payloads = self._dev(payloads, torch.uint8)
        if payloads.shape != (B, 55):
            raise ValueError("payloads must be uint8 [B,55], one per counter")
        code = self.polar_encode(payloads)
        pn, band = self.schedule(sec._prng.sub_key, band_key32, cd.to(torch.int64) & 0xFFFFFFFF)
        pre8 = np.packbits(np.concatenate((mseq_63().astype(np.uint8), np.zeros(1, np.uint8)))).tobytes()
        hdr16 = np.packbits(sec.pn_bits(0, 128)).tobytes()
        y_ws = torch.empty((B, 1215), dtype=torch.float64, device=self.device)
        frames = torch.empty((B, 1215), dtype=torch.float32, device=self.device)
        nat.check(self._ctx, self._lib.es_tx_frames_batch(self._ctx, _ptr(code), _ptr(pn), _ptr(band), _ptr(cd), pre8, hdr16, B,
                                                          _ptr(y_ws), _ptr(frames), self._stream()), "es_tx_frames_batch")
        return frames

    def synthetic_frames(self, key32: bytes, ctr0: int, n: int, *, seed: int = 20260101):
        """The benchmark workloads' frames for counters ctr0 .. ctr0+n-1, made wholly on the device (SURVEY 8d:
        plaintext b"ESAL" | ctr_be32 | nonce8 | pad11 sealed with a 12-byte nonce, random bytes from a seeded torch
        generator, then `make_frames`).  -> (frames float32 [n,1215], payloads uint8 [n,55]).  Input synthesis for
        configs 3 and 4, where 65 536 .. 2^20 frames would take the host embedder minutes."""
        from .crypto import SecureChannel
        sec = SecureChannel(key32)
        ctr = torch.arange(ctr0, ctr0 + n, dtype=torch.int64, device=self.device)
        # 31 random bytes per frame from a counter-based hash of (seed, ctr, byte index), so that a frame does not
        # depend on how the counter range is cut into batches or shards (32-bit multiply-xorshift rounds in int64)
        h = (ctr[:, None] * 31 + torch.arange(31, dtype=torch.int64, device=self.device)[None, :] + (int(seed) & 0xFFFFFF) * 1_000_003) & 0xFFFFFFFF
        for _ in range(3):
            h = (h * 0x45D9F3B) & 0xFFFFFFFF
            h = h ^ (h >> 16)
        rnd = (h & 0xFF).to(torch.uint8)
        plain = torch.empty((n, 27), dtype=torch.uint8, device=self.device)
        plain[:, :4] = torch.tensor(list(b"ESAL"), dtype=torch.uint8, device=self.device)
        for k in range(4):
            plain[:, 4 + k] = ((ctr >> (8 * (3 - k))) & 0xFF).to(torch.uint8)
        plain[:, 8:27] = rnd[:, :19]
        payloads = self.aead_seal(sec._aead._key, rnd[:, 19:31].contiguous(), plain)
        return self.make_frames(sec, key32, ctr, payloads), payloads

    # ------------------------------------------------------------------ after the list decoder
    def _ctr_dev(self, ctrs) -> torch.Tensor:
        """Frame counters as the 32-bit words the kernels compare against (stored in an int32 tensor)."""
        t = self._dev(ctrs, torch.int64) & 0xFFFFFFFF
        return torch.where(t >= 2 ** 31, t - 2 ** 32, t).to(torch.int32).contiguous()

    def aead_check(self, key32: bytes, blobs: torch.Tensor, ctrs: torch.Tensor, *, want_plain: bool = False):
        """The detector's validator (rtwm/detector.py:168-176) on 55-byte blobs: [n,55] or [B,L,55] uint8 with
        one expected counter per row / per B.  -> ok uint8 (same leading shape), optionally plaintext [...,27]."""
        if len(key32) != 32:
            raise ValueError("AEAD key must be 32 bytes")
        if blobs.dtype != torch.uint8 or blobs.shape[-1] != 55 or blobs.dim() not in (2, 3):
            raise ValueError("blobs must be uint8 [n,55] or [B,L,55]")
        blobs = blobs.contiguous()
        group = blobs.shape[1] if blobs.dim() == 3 else 1
        n = blobs.numel() // 55
        ctrs = self._ctr_dev(ctrs)
        if ctrs.numel() * group != n:
            raise ValueError("one expected counter per blob row (or per frame for [B,L,55]) is required")
        ok = torch.empty(blobs.shape[:-1], dtype=torch.uint8, device=self.device)
        plain = torch.empty(blobs.shape[:-1] + (27,), dtype=torch.uint8, device=self.device) if want_plain else None
        nat.check(self._ctx, self._lib.es_aead_check_batch(self._ctx, bytes(key32), _ptr(blobs), n, group, _ptr(ctrs), _ptr(ok),
                                                           _ptr(plain), self._stream()), "es_aead_check_batch")
        return (ok, plain) if want_plain else ok

    def aead_seal(self, key32: bytes, nonces: torch.Tensor, plain: torch.Tensor) -> torch.Tensor:
        """SecureChannel.seal for a batch of 27-byte plaintexts (rtwm/crypto.py:33-37): nonces uint8 [n,12],
        plain uint8 [n,27] -> blobs uint8 [n,55] on the device."""
        if len(key32) != 32:
            raise ValueError("AEAD key must be 32 bytes")
        nonces = self._dev(nonces, torch.uint8); plain = self._dev(plain, torch.uint8)
        if nonces.dim() != 2 or nonces.shape[1] != 12 or plain.shape != (nonces.shape[0], 27):
            raise ValueError("nonces must be [n,12] and plain [n,27]")
        blobs = torch.empty((nonces.shape[0], 55), dtype=torch.uint8, device=self.device)
        nat.check(self._ctx, self._lib.es_aead_seal_batch(self._ctx, bytes(key32), _ptr(nonces), _ptr(plain), nonces.shape[0],
                                                          _ptr(blobs), self._stream()), "es_aead_seal_batch")
        return blobs

    def select(self, scl: SclResult, *, key32: bytes | None = None, ctrs: torch.Tensor | None = None):
        """Tail of PolarCode.decode (rtwm/fastpolar.py:268-276, 332-359) for every record of an SclResult, on the
        GPU: -> (payload [B,55] uint8, ok [B] int8, which [B] int32).  key32=None is validator=None; with a key the
        validator is `aead_check` against ctrs [B].  ok = -1 marks records whose list loop had been skipped."""
        B, L = scl.cand_metric.shape
        if key32 is not None:
            if len(key32) != 32:
                raise ValueError("AEAD key must be 32 bytes")
            if ctrs is None:
                raise ValueError("one expected counter per record is required with a key")
            ctrs = self._ctr_dev(ctrs)
            if ctrs.numel() != B:
                raise ValueError("one expected counter per record is required with a key")
        payload = torch.empty((B, 55), dtype=torch.uint8, device=self.device)
        ok = torch.empty(B, dtype=torch.int8, device=self.device)
        which = torch.empty(B, dtype=torch.int32, device=self.device)
        nat.check(self._ctx, self._lib.es_select_batch(
            self._ctx, None if key32 is None else bytes(key32), _ptr(ctrs) if key32 is not None else None, B, L,
            _ptr(scl.hard_info), _ptr(scl.hard_ok), _ptr(scl.cand_info), _ptr(scl.cand_metric), _ptr(scl.cand_ok),
            _ptr(scl.ncand), _ptr(payload), _ptr(ok), _ptr(which), self._stream()), "es_select_batch")
        return payload, ok, which

    # ------------------------------------------------------------------ metric unit
    def decode_batch(self, frames: torch.Tensor, band: torch.Tensor, pn_rows: torch.Tensor, *,
                     start: torch.Tensor | None = None, list_size: int = 8, keep_corr: bool = False):
        """sync + LLR(variant 0 at `start`, default 0) + SCL-L for every record.

        After the band-pass the chain forks: correlation + peak picking and the demodulator (frame
        start known) both only need `y`, so the LLR kernel runs on a side HIP stream beside them;
        the branches are joined before the list decoder starts."""
        main = torch.cuda.current_stream(self.device)
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(self.device)
        fast = (not keep_corr) and frames.shape[1] - 62 <= self.FAST_MAX_LAGS
        if fast:      # float32 correlation screen + exact float64 fix-ups (identical thr / peaks)
            y, y32 = self.bpf2(frames, band)
        else:
            y = self.bpf(frames, band)
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):      # the demodulator only needs y: it runs beside sync
            llr = self.llr(y, band, pn_rows, start=start, variant=0)
        if fast:
            corr = None
            thr, peaks, npeaks, _flags = self.sync_fused(y, y32, band)
        else:
            corr = self.xcorr(y, band)
            thr, peaks, npeaks = self.pick(corr)
        main.wait_stream(self._side)             # join before SCL, which wants the chip to itself
        llr.record_stream(main)
        scl = self.scl(llr, list_size=list_size, skip_if_hard_ok=True)
        return SyncResult(y, corr if keep_corr else None, thr, peaks, npeaks), llr, scl


class DecodePipeline:
    """Streaming form of `RxEngine.decode_batch`: several batches in flight.  Batches are independent; per batch the kernels'
    inputs and results are those of decode_batch in every arrangement.  Three arrangements:

      * group=G (the throughput arrangement, bench.py's headline): the front ends (band-pass, fused sync, demodulator) of G
        consecutive batches run on `lanes` high-priority HIP streams and fill one LLR buffer; ONE list-decoder launch -- one
        lane per path, 64/L frames per wave, the mapping with the fewest instructions per frame -- decodes the group on one
        of `scl_streams` further streams.  A batch's list-decoder rows are a slice of its group's result
        (`GroupTicket.result()`), complete when the group's launch is; `synchronize()` / `wait` decode an incomplete group.
      * lanes=K: K independent whole-chain lanes, each one HIP stream with its own context that runs band-pass .. list decoder
        of its batches (k, k+K, ...) in order; rows of a batch within a few milliseconds of its submission.
      * neither (round 1): the front end on one stream (LLR on a side stream), the list decoder on one of `scl_streams`
        further streams, at most `depth` batches in flight.

    Results are complete after `wait(result)` / `synchronize()`.  A process should keep at most GPU_MAX_HW_QUEUES (8) HIP
    streams alive: pass `streams=` (lanes: a list; grouped: (front-end streams, list-decoder streams)) to build a further pipeline on
    existing ones; more live pipeline streams than hardware queues raises a RuntimeWarning (hw_queue_budget)."""

    def __init__(self, eng: "RxEngine", *, list_size: int = 8, scl_streams: int = 2, depth: int | None = None, lanes: int = 0,
                 side_stream: bool = True, group: int = 0, streams=None):
        self.eng = eng
        self.list_size = int(list_size)
        dev = eng.device
        self._own_streams = 0                         # streams this pipeline made itself (given back to the budget when it is collected)
        self._build(eng, dev, scl_streams, depth, lanes, side_stream, group, streams)

    def _mk(self, device, n: int, priority: int = 0) -> list:
        self._own_streams += n
        return pipeline_streams(device, n, priority)

    def __del__(self):  # pragma: no cover
        try:
            _LIVE_STREAMS[0] -= self._own_streams
        except Exception:
            pass

    def _build(self, eng, dev, scl_streams, depth, lanes, side_stream, group, streams) -> None:
        # `group` > 0: the throughput arrangement -- the front ends (band-pass .. demodulator) of `group` consecutive batches
        # run on `lanes` front streams and write their LLR rows into ONE buffer, and ONE list-decoder launch (on one of
        # `scl_streams` streams, own context, one lane per path: es_set_option "scl_lanes" = 1) decodes the whole group:
        # that mapping needs tens of thousands of frames per launch to fill the chip, which a 1 024-frame batch cannot give it.
        # Per batch the kernels' inputs and results are those of decode_batch; a batch's list-decoder rows are complete when
        # its group's launch is (GroupTicket.result() / wait / synchronize flush an incomplete group).
        self.group = max(0, int(group))
        if self.group:
            self.lanes = max(1, int(lanes) or 4)
            nb = max(1, int(scl_streams))
            # the short front-end kernels get dispatch priority: they must slip in whenever list-decoder waves leave
            # `streams` = (front-end streams, list-decoder streams) to run on instead of new ones
            self.lane_streams = list(streams[0])[:self.lanes] if streams is not None else self._mk(dev, self.lanes, priority=-1)
            if len(self.lane_streams) != self.lanes:
                raise ValueError("streams: one front-end stream per lane")
            self.lane_engs = [eng] + [RxEngine(dev, list_size_max=0) for _ in range(self.lanes - 1)]     # front-end contexts: no list-decoder scratch
            self.backs = list(streams[1])[:nb] if streams is not None else self._mk(dev, nb)
            if len(self.backs) != nb:
                raise ValueError("streams: one list-decoder stream per scl_streams")
            self.scl_engs = [RxEngine(dev, list_size_max=max(8, self.list_size)) for _ in range(nb)]
            for e in self.scl_engs:                   # kernel by launch size: a full group runs one lane per path, a lone batch one frame per wave
                e.set_option("scl_multi", -1); e.set_option("scl_lanes", 0); e.set_option("scl_lane_slab", 1)
            self.front = self.side = None
            self._ring: list = [None] * (nb + 1)      # the most recent groups: a new group's front ends wait for the decoder of the group nb + 1 before it
            self._open = None
            self._k = self._g = 0
            self._flush_lanes = 0                     # (tests: 1 = one lane per path whatever the group's size)
            return
        # `lanes` > 0: the other arrangement -- K independent lanes, each one HIP stream (= one hardware queue) with its
        # own context that runs the WHOLE chain of its batches (k, k+K, ...) in order; no cross-stream events at all.
        self.lanes = max(0, int(lanes))
        if self.lanes:
            # (lane 0 on the caller's own stream was measured slower: 1.29 M against 1.43 M frames/s at 7 lanes)
            # `streams`: HIP streams to run the lanes on instead of new ones -- a process that builds several pipelines should hand
            # the same streams to each: beyond GPU_MAX_HW_QUEUES (8) live streams, streams share hardware queues and serialise
            self.lane_streams = list(streams)[:self.lanes] if streams is not None else self._mk(dev, self.lanes)
            if len(self.lane_streams) != self.lanes:
                raise ValueError("streams: one per lane")
            self.lane_engs = [eng] + [RxEngine(dev, list_size_max=max(8, self.list_size)) for _ in range(self.lanes - 1)]
            self.scl_engs = self.lane_engs
            self.backs = self.lane_streams
            self.front = self.side = None
            self._k = 0
            return
        # the short front-end kernels get dispatch priority over the long-running list decoders
        self.front = self._mk(dev, 1, priority=-1)[0]
        self.side = self._mk(dev, 1, priority=-1)[0] if side_stream else self.front   # one hardware queue less without it
        self.backs = self._mk(dev, max(1, int(scl_streams)))
        self.scl_engs = [eng] + [RxEngine(dev, list_size_max=max(8, self.list_size)) for _ in self.backs[1:]]
        # Batches in flight = list-decoder streams: the front end of batch k waits for batch k-2 to leave, so it
        # runs while only ONE list decoder is resident (two of them fill every SIMD's register file and would
        # starve the short front-end kernels of wave slots), and its own list decoder then starts beside the
        # one still running.
        self.depth = len(self.backs) if depth is None else max(1, int(depth))
        self._inflight: list = []            # `done` events of the most recent batches
        self._k = 0

    def submit(self, frames: torch.Tensor, band: torch.Tensor, pn_rows: torch.Tensor, *,
               start: torch.Tensor | None = None, xcorr_events=None, select: bool = False, inputs_ready: bool = False):
        """Enqueue one batch.  inputs_ready (grouped arrangement): skip the wait on the caller's stream -- for INPUTS known to be
        complete on the device (10 us of host time per batch; it covers the inputs only: the group's own buffers are ordered by the pipeline).  start: frame starts [B] (None = 0; "peak" = the first detected peak of each record, lanes only);
        select (lanes only): also run the candidate selection (es_select_batch, validator None) on the lane's stream --
        the result is attached to the returned SclResult as `.selected = (payload, ok, which)`."""
        eng = self.eng
        if frames.shape[1] - 62 > eng.FAST_MAX_LAGS:
            raise ValueError("DecodePipeline serves frame-sized records (use RxEngine.decode_batch for long captures)")
        if self.group:
            return self._submit_grouped(frames, band, pn_rows, start, xcorr_events, select, inputs_ready)
        if self.lanes:
            j = self._k % self.lanes
            self._k += 1
            st, e = self.lane_streams[j], self.lane_engs[j]
            st.wait_stream(torch.cuda.current_stream(eng.device))
            with torch.cuda.stream(st):
                y, y32 = e.bpf2(frames, band)
                if xcorr_events is not None:
                    xcorr_events[0].record()
                thr, peaks, npeaks, flags = e.sync_fused(y, y32, band)        # correlation screen + exact picking, one kernel
                if xcorr_events is not None:
                    xcorr_events[1].record()
                if isinstance(start, str):                                    # "peak": demodulate at the first detected peak
                    start = peaks[:, 0].clamp(min=0).contiguous()
                llr = e.llr(y, band, pn_rows, start=start, variant=0)
                scl = e.scl(llr, list_size=self.list_size, skip_if_hard_ok=True)
                if select:
                    scl.selected = e.select(scl)
                done = torch.cuda.Event()
                done.record()
            for t in (frames, band, pn_rows):
                t.record_stream(st)
            return SyncResult(y, None, thr, peaks, npeaks, flags=flags), llr, scl, done
        self.front.wait_stream(torch.cuda.current_stream(eng.device))   # inputs were produced on the caller's stream
        if len(self._inflight) >= self.depth:                           # at most `depth` batches in flight
            self.front.wait_event(self._inflight.pop(0))
        with torch.cuda.stream(self.front):
            y, y32 = eng.bpf2(frames, band)
            if self.side is not self.front:
                self.side.wait_stream(self.front)
            with torch.cuda.stream(self.side):
                llr = eng.llr(y, band, pn_rows, start=start, variant=0)
            if xcorr_events is not None:
                xcorr_events[0].record()
            thr, peaks, npeaks, flags = eng.sync_fused(y, y32, band)
            if xcorr_events is not None:
                xcorr_events[1].record()
            if self.side is not self.front:
                self.front.wait_stream(self.side)
            ready = torch.cuda.Event()
            ready.record()
        j = self._k % len(self.backs)
        self._k += 1
        back = self.backs[j]
        back.wait_event(ready)
        llr.record_stream(back)
        with torch.cuda.stream(back):
            scl = self.scl_engs[j].scl(llr, list_size=self.list_size, skip_if_hard_ok=True)
            done = torch.cuda.Event()
            done.record()
        self._inflight.append(done)
        return SyncResult(y, None, thr, peaks, npeaks, flags=flags), llr, scl, done

    # ---- grouped arrangement
    def _submit_grouped(self, frames, band, pn_rows, start, xcorr_events, select, inputs_ready=False):
        B = frames.shape[0]
        g = self._open
        if g is not None and (g.B != B or g.select != bool(select)):
            self._flush(g); g = None
        if g is None:
            r = self._g % len(self._ring)
            g = _Group(self, B, bool(select), self._ring[r], self.eng.device)
            self._ring[r] = g                         # (only its `done` event is looked at again: the throttle of the group opened R groups later)
            self._open = g
            self._g += 1
        j = self._k % self.lanes
        self._k += 1
        st, e = self.lane_streams[j], self.lane_engs[j]
        if not inputs_ready:                                                  # (inputs_ready: the caller vouches that frames / band / pn_rows are complete on the device)
            st.wait_stream(torch.cuda.current_stream(self.eng.device))
        if g.throttle is not None:
            st.wait_event(g.throttle)                                         # at most len(ring) groups in flight on the GPU
        if g.llr is None:
            g.allocate(st)                                                    # on THIS lane's stream: the block's earlier life is ordered before its writes
        if j not in g.lanes_used:
            g.lanes_used.add(j)
            if st is not g.alloc_stream:
                st.wait_event(g.alloc_ev)                                     # (a recycled block may still be in use by work queued on the allocating stream)
                g.llr.record_stream(st)
        slot = g.count
        rows = g.llr[slot * B:(slot + 1) * B]
        with torch.cuda.stream(st):
            if xcorr_events is None and not isinstance(start, str):              # the three launches through one library call
                y, thr, peaks, npeaks, flags, _ = e.front(frames, band, pn_rows, start=start, out=rows)
            else:
                y, y32 = e.bpf2(frames, band)
                if xcorr_events is not None:
                    xcorr_events[0].record()
                thr, peaks, npeaks, flags = e.sync_fused(y, y32, band)
                if xcorr_events is not None:
                    xcorr_events[1].record()
                if isinstance(start, str):
                    start = peaks[:, 0].clamp(min=0).contiguous()
                e.llr(y, band, pn_rows, start=start, variant=0, out=rows)
            ready = torch.cuda.Event()
            ready.record()
        for t in (frames, band, pn_rows):
            t.record_stream(st)
        g.ready.append(ready)
        g.count += 1
        ticket = GroupTicket(g, slot)
        if g.count == g.capacity:
            self._flush(g)
        return SyncResult(y, None, thr, peaks, npeaks, flags=flags), rows, ticket, ticket

    def _flush(self, g) -> None:
        if g.done is not None:
            return
        j = g.index % len(self.backs)
        back, e = self.backs[j], self.scl_engs[j]
        for ev in g.ready:
            back.wait_event(ev)
        lp = 1
        while lp < self.list_size:
            lp <<= 1
        # one lane per path once the group gives every SIMD a wave (launches of several groups overlap); a lone batch: the library's choice
        e.set_option("scl_lanes", 1 if (g.count * g.B * lp >= 64 * 1024 or self._flush_lanes == 1) else 0)
        g.llr.record_stream(back)
        with torch.cuda.stream(back):
            g.scl = e.scl(g.llr[:g.count * g.B], list_size=self.list_size, skip_if_hard_ok=True)
            if g.select:
                g.selected = e.select(g.scl)
            g.done = torch.cuda.Event()
            g.done.record()
        if self._open is g:
            self._open = None

    @staticmethod
    def wait(result) -> None:
        result[3].synchronize()

    def synchronize(self) -> None:
        if self.group:
            if self._open is not None:
                self._flush(self._open)
            for st in self.lane_streams:
                st.synchronize()
        if self.front is not None:
            self.front.synchronize(); self.side.synchronize()
        for b in self.backs:
            b.synchronize()


class _Group:
    """Batches that share one list-decoder launch (DecodePipeline, grouped arrangement)."""

    def __init__(self, pipe: DecodePipeline, B: int, select: bool, prev, device):
        self.pipe, self.B, self.select = pipe, B, select
        self.index = pipe._g
        self.capacity = pipe.group
        self.device = device
        # a buffer of its own (the rows handed out to the caller stay valid as long as the caller keeps them), allocated by the first
        # submit on that batch's lane stream (`allocate`); the caching allocator is told about every other stream that touches it
        self.llr = self.alloc_stream = self.alloc_ev = None
        self.throttle = prev.done if prev is not None else None
        self.lanes_used: set = set()
        self.ready: list = []
        self.count = 0
        self.scl = self.selected = self.done = None

    def allocate(self, stream) -> None:
        with torch.cuda.stream(stream):
            self.llr = torch.empty((self.capacity * self.B, 1024), dtype=torch.float32, device=self.device)
            self.alloc_ev = torch.cuda.Event()
            self.alloc_ev.record()
        self.alloc_stream = stream


class GroupTicket:
    """A batch's place in its group: `result()` -> SclResult rows of this batch (flushes / waits as needed)."""

    def __init__(self, g: _Group, slot: int):
        self.g, self.slot = g, slot

    def synchronize(self) -> None:
        self.g.pipe._flush(self.g)
        self.g.done.synchronize()

    def result(self) -> SclResult:
        self.synchronize()
        lo, hi = self.slot * self.g.B, (self.slot + 1) * self.g.B
        s = self.g.scl
        if not getattr(self.g, "checked", False):
            s.check(); self.g.checked = True                  # (the launch is complete: one small reduction per group)
        res = SclResult(s.hard_info[lo:hi], s.hard_ok[lo:hi], s.cand_info[lo:hi], s.cand_metric[lo:hi], s.cand_ok[lo:hi], s.ncand[lo:hi])
        if self.g.selected is not None:
            res.selected = tuple(t[lo:hi] for t in self.g.selected)
        return res


def select_payload(scl: SclResult, row: int = 0, validator=None):
    """Host-side tail of PolarCode.decode (rtwm/fastpolar.py:268-276, 332-359) for one record:
    apply CRC / validator rules to the hard candidate and the metric-ordered list."""
    hard = bytes(scl.hard_info[row].cpu().numpy().tobytes())
    hard_ok = bool(scl.hard_ok[row].item())

    def _valid(payload: bytes) -> bool:
        try:
            return bool(validator(payload))
        except Exception:
            return False

    if hard_ok and (validator is None or _valid(hard)):
        return hard, True
    n = int(scl.ncand[row].item())
    if n < 0:
        raise nat.NativeError("es_scl_batch: this record was not decoded (no free scratch-slab slot)")
    if n == 0:      # list loop skipped although the shortcut did not return: only when validator is set
        raise RuntimeError("list decode was skipped; call scl(..., skip_if_hard_ok=False) when using a validator")
    infos = scl.cand_info[row].cpu().numpy()
    oks = scl.cand_ok[row].cpu().numpy()
    metrics = scl.cand_metric[row].cpu().numpy()
    best_crc = None
    best_any = (np.inf, hard)
    for r in range(n):
        payload = infos[r].tobytes()
        if oks[r]:
            if validator is None or _valid(payload):
                return payload, True
            if best_crc is None or metrics[r] < best_crc[0]:
                best_crc = (metrics[r], payload)
        elif metrics[r] < best_any[0]:
            best_any = (metrics[r], payload)
    if best_crc is not None:
        return best_crc[1], False
    return best_any[1], False
