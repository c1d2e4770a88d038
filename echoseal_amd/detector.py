"""WatermarkDetector with the reference's interface (rtwm/detector.py:24-515), MI355X inside.

Host orchestration (band order, counter candidates, AEAD validation, anti-replay nonce) is plain
Python as in the reference; every numeric stage runs on the GPU through echoseal_amd.engine:

    _scan_band_multi_frame  -> es_bpf / es_xcorr / es_pick        (rtwm/detector.py:59-99)
    _decode_header          -> es_header_batch                    (rtwm/detector.py:452-515)
    _llr                    -> es_llr_batch                       (rtwm/detector.py:296-416)
    polar decode            -> es_scl_batch + host validator scan (rtwm/fastpolar.py:254-359)

Constructing a detector needs no GPU (keys, static sequences); the first numeric call creates the
engine and raises if the HIP library or the device is missing.  The reference's unconditional
print() debugging is not reproduced.
"""
from __future__ import annotations

import numpy as np

from ._native import NativeError
from .crypto import SecureChannel
from .polar_fast import N_DEFAULT
from .primitives import InvalidTag
from .tables import matched_filter_taps
from .utils import BAND_PLAN, BandHop, butter_bandpass, choose_band, mseq_63, resample_to  # noqa: F401 (re-exported)

PRE_BITS = mseq_63()
PRE_L = len(PRE_BITS)
HDR_BITS = 16
HDR_REPEAT = 8
HDR_L = 128
FRAME_LEN = PRE_L + HDR_L + N_DEFAULT
TIGHT_DELTA = 3
WIDE_DELTA = 200
EPS = 1e-12

MAX_TRIES = 400          # rtwm/detector.py:107
PEAK_LIMIT = 25          # rtwm/detector.py:108


class WatermarkDetector:
    """Recover an EchoSeal watermark from a recording (reference docstring: >= 3 s)."""

    def __init__(self, key32: bytes, *, fs_target: int = 48_000, list_size: int = 256, engine=None) -> None:
        self.sec = SecureChannel(key32)
        self.fs_target = fs_target
        self.session_nonce: bytes | None = None
        self._band_key = getattr(self.sec, "band_key", key32)
        self._hop = BandHop(self._band_key)                 # choose_band(self._band_key, .) memoised for the life of this detector
        self._mf_cache: dict = {}
        self._list_size = int(list_size)
        self._aead = getattr(self.sec, "_aead", None)
        self._pre_sy = 2.0 * PRE_BITS.astype(np.float32) - 1.0
        self._hdr_pn_sy = 2.0 * self.sec.pn_bits(0, HDR_L).astype(np.float32) - 1.0
        if self._hdr_pn_sy.size != HDR_L:
            raise RuntimeError(f"Header PN length {self._hdr_pn_sy.size} != expected {HDR_L}")
        self._engine = engine
        self._trace: list[tuple[int, int, int]] | None = None   # set to [] to record (band_lo, peak, ctr) tries
        self._hdr_trace: list[tuple[float, float, float]] | None = None   # set to [] to record every header decode of a scan

    # ------------------------------------------------------------------ engine plumbing
    @property
    def engine(self):
        if self._engine is None or self._engine.fs != self.fs_target:
            from .fastpolar import engine_for_fs
            try:
                self._engine = engine_for_fs(self.fs_target)       # tables (band-pass, template, taps) of this rate
            except ValueError as e:                                 # SciPy's own error (band above Nyquist) passes through, as in the reference
                if "taps" not in str(e):
                    raise
                raise NotImplementedError(f"fs_target = {self.fs_target}: {e} (DESIGN.md section 7)") from None
        return self._engine

    def _band_id(self, band) -> int:
        return BAND_PLAN.index((int(band[0]), int(band[1])))

    def _dev(self, arr: np.ndarray, dtype):
        import torch
        return torch.from_numpy(np.ascontiguousarray(arr, dtype=dtype)).to(self.engine.device)

    # ------------------------------------------------------------------ API
    def verify_wav(self, path: str) -> bool:
        """verify() for a WAV file (f-4 ingest): PCM16 samples stay int16 up to the band-pass kernel (x / 32768 there: the
        values soundfile.read hands the reference, rx_app.py:25-28)."""
        from .audiofile import read_wav
        samples, fs = read_wav(path)
        return self.verify(samples, fs)

    def _conditioned(self, audio, fs_in: int) -> np.ndarray:
        audio = np.asarray(audio)
        if audio.dtype == np.int16:
            if fs_in == self.fs_target:
                return audio                                 # ingested as int16 by the band-pass kernel
            audio = audio.astype(np.float32) / np.float32(32768.0)
        if fs_in != self.fs_target and audio.ndim == 1 and audio.size:
            # polyphase resampling on the device (es_resample_batch): the values scipy.signal.resample_poly returns
            return self.engine.resample(audio, fs_in, self.fs_target).cpu().numpy()
        signal, _ = resample_to(self.fs_target, audio, fs_in)
        return signal

    def _band_order(self):
        hop0 = self._hop.band(0)
        return [hop0] + [b for b in BAND_PLAN if b != hop0]                 # rtwm/detector.py:46-52

    def verify(self, audio: np.ndarray, fs_in: int) -> bool:
        return self.verify_batch([audio], fs_in)[0]

    def verify_batch(self, clips, fs_in) -> list[bool]:
        """verify() for several recordings ("full batched verify()"): the result, the order of tries
        and the evolution of `session_nonce` are those of calling the reference's verify() on the clips one after the other
        (rtwm/detector.py:44-53, 105-152); what is batched is the GPU work.  Per group of equally long clips: ONE sync launch
        sequence over (clips x 4 bands) records, ONE header decode over every peak that can hold a frame and ONE demodulate +
        list-decode + validate batch over the (peak, counter) candidates of all clips and bands; the host then walks clip by clip
        and band by band in the reference's order with its early returns."""
        fs_list = list(fs_in) if isinstance(fs_in, (list, tuple)) else [fs_in] * len(clips)
        signals = [np.asarray(self._conditioned(c, f)).reshape(-1) for c, f in zip(clips, fs_list)]
        signals = [sg if sg.dtype == np.int16 else sg.astype(np.float32, copy=False) for sg in signals]
        order = self._band_order()
        scans: list = [None] * len(signals)
        groups: dict[int, list[int]] = {}
        for i, sgl in enumerate(signals):
            groups.setdefault((sgl.size, sgl.dtype == np.int16), []).append(i)
        for (size, _is_i16), idx in groups.items():
            if size < PRE_L:                                                # rtwm/detector.py:71-73
                continue
            for i, sc in zip(idx, self._scan_prepare([signals[i] for i in idx], order)):
                scans[i] = sc
        # Decoding is stateless (the validator's verdict depends on blob and counter only; nonce bookkeeping happens on the host, in
        # _accept), so it is batched ahead of the walk; the WALK is clip by clip and band by band, in the reference's order with its
        # early returns.  When the walk needs a (clip, band) that is not decoded yet, that band and -- in walk order: the clip's
        # further bands, then the later clips of the group -- as many further ones as fit under a cap on the candidates per batch go
        # through ONE demodulate + list-decode + validate batch.  A lone clip (a few hundred candidates) is decoded in one batch,
        # as before; a hundred unwatermarked clips at list size 256 (4 x 400 candidates x 4 variants each) no longer ask for
        # gigabytes of candidate rows at once, and what an early return makes unnecessary is bounded by the cap.
        plans: dict[int, list] = {i: [self._scan_plan(scans[i], bi) for bi in range(len(order))] for i in range(len(signals)) if scans[i] is not None}
        cache: dict[tuple[int, int], list] = {}
        group_of = {i: idx for idx in groups.values() for i in idx}
        cap = self._pair_cap()

        def need(i: int, bi: int) -> list:
            if (i, bi) not in cache:
                todo = [(i, r) for r in range(bi, len(order))] + [(j, r) for j in group_of[i] if j > i and scans[j] is not None for r in range(len(order))]
                batch, n = [], 0
                for (j, r) in todo:
                    if (j, r) in cache:
                        continue
                    m = len(plans[j][r][0])
                    if batch and n + m > cap:
                        break
                    batch.append((j, r)); n += m
                flat = [(j, p) for (j, r) in batch for p in plans[j][r][0]]
                res = self._decode_pairs_grouped([(scans[j]["frames"], p[0], p[2]) for j, p in flat]) if flat else []
                at = 0
                for (j, r) in batch:
                    m = len(plans[j][r][0])
                    cache[(j, r)] = res[at:at + m]
                    at += m
            return cache[(i, bi)]

        out = []
        for i in range(len(signals)):
            ok = False
            if scans[i] is not None:
                for bi, (plan, hdr_log) in enumerate(plans[i]):
                    if self._scan_replay(scans[i], bi, plan, hdr_log, need(i, bi) if plan else []):
                        ok = True
                        break
            out.append(ok)
        return out

    # one scan = what _scan_band_multi_frame needs for every band of one clip, produced in batched launches
    def _scan_prepare(self, signals: list, bands: list) -> list:
        import torch
        eng = self.engine
        g, nb, M = len(signals), len(bands), signals[0].size
        x = self._dev(np.repeat(np.stack(signals), nb, axis=0), signals[0].dtype)      # row = clip * nb + band (float32, or int16 samples)
        bid = self._dev(np.tile(np.array([self._band_id(b) for b in bands]), g), np.uint8)
        sy = eng.sync_fast(x, bid) if M - (PRE_L - 1) <= eng.FAST_MAX_LAGS else eng.sync(x, bid, keep_corr=False)
        npk = (sy.npeaks.cpu().numpy() & 0xFFFF)
        pk = sy.peaks.cpu().numpy()
        rows, starts = [], []
        for r in range(g * nb):
            for st in pk[r, :min(int(npk[r]), pk.shape[1], PEAK_LIMIT)]:
                if st + FRAME_LEN <= M:                                     # rtwm/detector.py:112-113
                    rows.append(r); starts.append(int(st))
        frames = None
        hdr = (np.zeros(0, bool), np.zeros(0, np.int64), np.zeros(0))
        if rows:
            rt = torch.tensor(rows, device=eng.device)
            cols = torch.tensor(starts, device=eng.device)[:, None] + torch.arange(FRAME_LEN, device=eng.device)[None, :]
            frames = sy.y[rt[:, None], cols].contiguous()                  # [P, 1215] float64: y[start : start + 1215]
            okh, val, score = eng.header(frames, bid[rt].contiguous(),
                                         self._dev(np.packbits(self.sec.pn_bits(0, HDR_L)).reshape(1, -1), np.uint8))
            hdr = (okh.cpu().numpy().astype(bool), val.cpu().numpy().astype(np.int64), score.cpu().numpy().astype(np.float64))
        rows_a = np.array(rows, np.int64)
        return [{"bands": bands, "frames": frames, "rows": rows_a - c * nb, "sel": np.flatnonzero((rows_a // nb) == c) if rows else np.zeros(0, np.int64),
                 "starts": np.array(starts, np.int64), "hdr": hdr} for c in range(g)]

    def _scan_plan(self, scan, bi: int):
        """The candidate (peak, counter) pairs of one band in the reference's try order (rtwm/detector.py:105-140):
        -> (plan [(peak slot j, start, ctr, header-log index)], header log of the peaks looked at)."""
        band = scan["bands"][bi]
        sel = [int(j) for j in scan["sel"] if scan["rows"][j] == bi]       # this band's peaks, in peak order
        plan: list[tuple[int, int, int, int]] = []
        hdr_log = []
        tried = 0
        for j in sel:
            if tried >= MAX_TRIES:
                break
            start = int(scan["starts"][j])
            ctr_est = int(round(start / FRAME_LEN))
            hdr_ok, ctr_lo16, score = bool(scan["hdr"][0][j]), int(scan["hdr"][1][j]), float(scan["hdr"][2][j])
            hdr_log.append((float(hdr_ok), float(ctr_lo16), score))
            cands: list[int] = []
            if hdr_ok:                                                      # rtwm/detector.py:122-127
                for ctr in range(max(0, ctr_est - WIDE_DELTA), ctr_est + WIDE_DELTA + 1):
                    if (ctr & 0xFFFF) == ctr_lo16 and self._hop.band(ctr) == band:
                        cands.append(ctr)
            else:                                                           # :131-140
                for ctr in range(max(0, ctr_est - TIGHT_DELTA), ctr_est + TIGHT_DELTA + 1):
                    if self._hop.band(ctr) == band:
                        cands.append(ctr)
                if not cands:
                    for ctr in range(max(0, ctr_est - WIDE_DELTA), ctr_est + WIDE_DELTA + 1):
                        if self._hop.band(ctr) == band:
                            cands.append(ctr)
            for ctr in cands[:MAX_TRIES - tried]:
                plan.append((j, start, ctr, len(hdr_log) - 1))              # (.., index of this peak's header decode in hdr_log)
            tried += len(cands[:MAX_TRIES - tried])
        return plan, hdr_log

    def _scan_replay(self, scan, bi: int, plan, hdr_log, results) -> bool:
        """Walk one band's decoded candidates as the reference does (early return, traces, nonce bookkeeping in _accept)."""
        band = scan["bands"][bi]
        for (j, start, ctr, h), blobs in zip(plan, results):
            if self._trace is not None:
                self._trace.append((int(band[0]), int(start), int(ctr)))
            if self._accept(blobs, ctr):
                if self._hdr_trace is not None:                             # the reference decodes a peak's header when it reaches the peak:
                    self._hdr_trace.extend(hdr_log[:h + 1])                 # peaks after the accepted one were never looked at
                return True
        if self._hdr_trace is not None:
            self._hdr_trace.extend(hdr_log)
        return False

    def _scan_decide(self, scan, bi: int) -> bool:
        """The per-band loop of _scan_band_multi_frame (rtwm/detector.py:105-152) over prepared peaks / headers."""
        plan, hdr_log = self._scan_plan(scan, bi)
        if not plan:
            if self._hdr_trace is not None:
                self._hdr_trace.extend(hdr_log)
            return False
        results = self._decode_pairs(scan["frames"], [p[0] for p in plan], [p[2] for p in plan])
        return self._scan_replay(scan, bi, plan, hdr_log, results)

    def verify_raw_frame(self, signal: np.ndarray) -> bool:
        signal = np.asarray(signal)
        if len(signal) == FRAME_LEN:
            for ctr in range(4):
                band = self._hop.band(ctr)
                y = self._bandpass(signal, band)
                if self._try_decode_frame(y, ctr):
                    return True
        return self._scan_band_multi_frame(signal, self._hop.band(0))

    def _scan_band(self, signal: np.ndarray, band, skip_filtering: bool = False) -> bool:
        return self._scan_band_multi_frame(signal, band)

    def _try_window(self, frame: np.ndarray, ctr0: int, delta: int) -> bool:
        for ctr in range(max(0, ctr0 - delta), ctr0 + delta + 1):
            if self._try_decode_frame(frame, ctr):
                return True
        return False

    # ------------------------------------------------------------------ sync (GPU)
    def _bandpass(self, signal: np.ndarray, band) -> np.ndarray:
        x = self._dev(np.asarray(signal).astype(np.float32, copy=False).reshape(1, -1), np.float32)
        b = self._dev(np.array([self._band_id(band)]), np.uint8)
        return self.engine.bpf(x, b)[0].cpu().numpy()

    def _sync(self, signal: np.ndarray, band):
        """-> (y float64[M], thr, peaks list) or None when the record is shorter than the template."""
        sig = np.asarray(signal).astype(np.float32, copy=False).reshape(-1)
        if sig.size < PRE_L:                               # rtwm/detector.py:71-73
            return None
        x = self._dev(sig.reshape(1, -1), np.float32)
        b = self._dev(np.array([self._band_id(band)]), np.uint8)
        sy = self.engine.sync(x, b, keep_corr=False)
        n = int(sy.npeaks[0].item()) & 0xFFFF
        peaks = [int(p) for p in sy.peaks[0, : min(n, sy.peaks.shape[1])].cpu().numpy()]
        return sy.y[0].cpu().numpy(), float(sy.thr[0].item()), peaks

    def _scan_band_multi_frame(self, signal: np.ndarray, band) -> bool:
        sig = np.asarray(signal).reshape(-1)
        if sig.dtype != np.int16:
            sig = sig.astype(np.float32, copy=False)
        if sig.size < PRE_L:                                   # rtwm/detector.py:71-73
            return False
        return self._scan_decide(self._scan_prepare([sig], [band])[0], 0)

    # ------------------------------------------------------------------ demod + FEC (GPU)
    def _matched_filter_taps(self, band):
        key = (band[0], band[1], self.fs_target)
        h = self._mf_cache.get(key)
        if h is None:
            h = self._mf_cache[key] = matched_filter_taps((int(band[0]), int(band[1])), self.fs_target)
        return h

    def _pn_rows(self, ctrs) -> np.ndarray:
        return self.sec.pn_bytes_batch(list(ctrs), 152)

    def _llr(self, frame: np.ndarray, frame_id: int, pn_variant: int = 0) -> np.ndarray:
        frame = np.asarray(frame, dtype=np.float64).reshape(-1)
        if frame.size == 0:
            return np.zeros(N_DEFAULT, dtype=np.float32)
        band = self._hop.band(frame_id)
        y = self._dev(frame.reshape(1, -1), np.float64)
        out = self.engine.llr(y, self._dev(np.array([self._band_id(band)]), np.uint8),
                              self._dev(self._pn_rows([frame_id]), np.uint8), variant=int(pn_variant))
        return out[0].cpu().numpy()

    def _decode_header(self, frame: np.ndarray, band) -> tuple[bool, int, float]:
        frame = np.asarray(frame, dtype=np.float64).reshape(-1)
        if frame.size < PRE_L + HDR_L:                     # rtwm/detector.py:461-462
            return False, 0, 0.0
        y = self._dev(frame.reshape(1, -1), np.float64)
        ok, val, score = self.engine.header(y, self._dev(np.array([self._band_id(band)]), np.uint8),
                                            self._dev(np.packbits(self.sec.pn_bits(0, HDR_L)).reshape(1, -1), np.uint8))
        return bool(ok[0].item()), int(val[0].item()), float(score[0].item())

    def _validator(self, frame_ctr: int):
        def check(payload: bytes) -> bool:
            try:
                pt = self.sec.open(payload)
            except Exception:
                return False
            return pt.startswith(b"ESAL") and int.from_bytes(pt[4:8], "big") == frame_ctr
        return check

    def _decode_candidates(self, frame: np.ndarray, ctrs) -> list[list[bytes | None]]:
        """For every counter: the four polar decodes the reference tries in order
        (+llr0, -llr0, +llr1, -llr1; rtwm/detector.py:161-190), each None or a 55-byte blob."""
        ctrs = list(ctrs)
        if not ctrs:
            return []
        frames = self._dev(np.asarray(frame, dtype=np.float64).reshape(1, -1), np.float64)
        return self._decode_pairs(frames, [0] * len(ctrs), ctrs)

    def _decode_pairs_grouped(self, triples) -> list[list[bytes | None]]:
        """_decode_pairs for (frames tensor, row, ctr) triples that may come from several clips (each clip has its own
        frames tensor): consecutive triples of one tensor go through one call."""
        out: list = []
        k = 0
        while k < len(triples):
            fr = triples[k][0]
            m = k
            while m < len(triples) and triples[m][0] is fr:
                m += 1
            out += self._decode_pairs(fr, [t[1] for t in triples[k:m]], [t[2] for t in triples[k:m]])
            k = m
        return out

    def _pair_cap(self) -> int:
        """(frame, counter) pairs per decode launch: at most 2^18 list paths per sign / PN variant (1 024 pairs at the default list
        size 256 -- the candidate rows of one launch are then 58 MB --, 32 768 at list size 8)."""
        return max(64, (1 << 18) // max(1, self._list_size))

    def _decode_pairs(self, frames, rows, ctrs) -> list[list[bytes | None]]:
        """Chunked front of _decode_pairs_chunk, so that a long candidate list never asks for gigabytes at once."""
        cap = self._pair_cap()
        if len(ctrs) <= cap:
            return self._decode_pairs_chunk(frames, rows, ctrs)
        out: list = []
        for k in range(0, len(ctrs), cap):
            out += self._decode_pairs_chunk(frames, rows[k:k + cap], ctrs[k:k + cap])
        return out

    def _decode_pairs_chunk(self, frames, rows, ctrs) -> list[list[bytes | None]]:
        """The same for (frame, counter) pairs: frames = device tensor [P, <=1215] float64, pair i = (frames[rows[i]], ctrs[i]).
        One batch: two demodulations (PN variants 0 / 1), one list decode of the 4 B sign / variant combinations, one
        validation + selection (es_select_batch with the AEAD key) -- no host round trip per candidate."""
        from .engine import select_payload
        import torch
        eng = self.engine
        L = self._list_size
        if L > eng.list_size_max:
            raise NotImplementedError(f"list_size={L}: the HIP decoder supports list sizes up to {eng.list_size_max}")
        B = len(ctrs)
        y = frames[torch.tensor(rows, device=eng.device)].contiguous()
        # PN rows and band indices of the candidate counters straight from the device schedule (es_schedule_batch)
        pn, bands = eng.schedule(self.sec._prng.sub_key, self._band_key, ctrs=torch.tensor(ctrs, dtype=torch.int64))
        l0 = eng.llr(y, bands, pn, variant=0)
        l1 = eng.llr(y, bands, pn, variant=1)
        res = eng.scl(torch.cat((l0, -l0, l1, -l1), dim=0), list_size=L, skip_if_hard_ok=False)
        key = getattr(self._aead, "_key", None)
        if key is not None:
            # validator + candidate selection on the GPU (es_select_batch): same rules as select_payload with
            # self._validator(ctr), without a Python callback per candidate
            payload, ok, _which = eng.select(res, key32=key, ctrs=torch.tensor(ctrs * 4, dtype=torch.int64))
            payload = payload.cpu().numpy(); ok = ok.cpu().numpy()
            if (ok == -2).any():
                raise NativeError("es_scl_batch: some candidate records were not decoded (no free scratch-slab slot)")
            return [[payload[v * B + i].tobytes() if ok[v * B + i] == 1 else None for v in range(4)] for i in range(B)]
        out = []
        for i, ctr in enumerate(ctrs):
            val = self._validator(ctr)
            row = []
            for v in range(4):
                payload, ok = select_payload(res, v * B + i, val)
                row.append(payload if ok else None)
            out.append(row)
        return out

    def _accept(self, blobs, frame_ctr: int) -> bool:
        """Tail of _try_decode_frame (rtwm/detector.py:182-233): first non-None blob, AEAD open,
        magic, counter, session-nonce bookkeeping."""
        blob = next((b for b in blobs if b is not None), None)
        if blob is None:
            return False
        try:
            plain = self.sec.open(blob)
        except Exception:
            fb, _layout = self._decrypt_blob_fallback(blob)
            if fb is not None:
                plain = fb
            elif len(blob) >= 4 and blob[:4] == b"ESAL":
                plain = blob
            else:
                return False
        if not plain.startswith(b"ESAL"):
            return False
        if int.from_bytes(plain[4:8], "big") != frame_ctr:
            return False
        nonce = plain[8:16]
        if self.session_nonce and nonce == self.session_nonce:
            return True
        if self.session_nonce is None:
            self.session_nonce = nonce
            return True
        return False

    def _try_decode_frame(self, frame: np.ndarray, frame_ctr: int) -> bool:
        return self._accept(self._decode_candidates(frame, [frame_ctr])[0], frame_ctr)

    def _decrypt_blob_fallback(self, blob: bytes):
        if self._aead is None:
            return None, None
        if len(blob) >= 12:
            for nonce, body, name in ((blob[:12], blob[12:], "nonce-front"), (blob[-12:], blob[:-12], "nonce-tail")):
                try:
                    return self._aead.decrypt(nonce, body, b""), name
                except InvalidTag:
                    pass
        return None, None
