"""Watermark embedder (TX side) with the reference's interface (rtwm/embedder.py:20-168).

Host NumPy/SciPy code: it only exists here to synthesise the 48 kHz frames the receive path is
tested and benchmarked on.  `make_frames` is a batched variant of
`_make_frame_chips` with caller-supplied payloads, so fixtures and benchmark inputs are
reproducible (the reference draws payload randomness from `secrets`).
"""
from __future__ import annotations

import secrets
from dataclasses import dataclass, field

import numpy as np
from scipy.signal import lfilter

from .crypto import SecureChannel
from .polar_fast import K_DEFAULT, N_DEFAULT, encode as polar_enc
from .tables import band_coeffs
from .utils import BAND_PLAN, band_index, db_to_lin, mseq_63

EPS = 1e-12
MIN_RMS_SILENCE = 1e-4
MIX_HEADROOM = 0.98
HDR_BITS = 16
HDR_REPEAT = 8
HDR_L = 128


@dataclass
class TxParams:
    fs: int = 48_000
    target_rel_db: float = -10.0
    floor_rel_dbfs: float = -35.0
    N: int = N_DEFAULT
    K: int = K_DEFAULT
    preamble: np.ndarray = field(default_factory=mseq_63)


class WatermarkEmbedder:
    def __init__(self, key32: bytes, params: TxParams | None = None) -> None:
        self.p = params or TxParams()
        self.sec = SecureChannel(key32)
        self._band_key = getattr(self.sec, "band_key", key32)
        self.frame_ctr = 0
        self._chip_buf: np.ndarray | None = None
        self._session_nonce = secrets.token_bytes(8)
        self._preamble_sy = 2.0 * self.p.preamble.astype(np.float32) - 1.0
        self._hdr_pn_sy = 2.0 * self.sec.pn_bits(0, HDR_L).astype(np.float32) - 1.0

    # ------------------------------------------------------------------ API
    def process(self, samples: np.ndarray) -> np.ndarray:
        if self._chip_buf is None:
            self._chip_buf = np.empty(0, dtype=np.float32)
        x = samples.astype(np.float32, copy=False)
        in_rms = float(np.sqrt(np.mean(x * x)) + EPS)
        need = samples.size
        while self._chip_buf.size < need:
            self._chip_buf = np.concatenate((self._chip_buf, self._make_frame_chips()))
            self.frame_ctr = (self.frame_ctr + 1) % (2 ** 32)
        chips = self._chip_buf[:need].astype(np.float32, copy=False)
        self._chip_buf = self._chip_buf[need:]
        scale = max(db_to_lin(self.p.target_rel_db) * in_rms, db_to_lin(self.p.floor_rel_dbfs))
        headroom = max(MIX_HEADROOM - float(np.max(np.abs(x))), 0.0)
        peak = float(np.max(np.abs(chips))) + EPS
        scale = min(scale, headroom / peak) if peak > 0.0 else 0.0
        return x + chips * scale

    # ------------------------------------------------------------------ internals
    def _frame_symbols(self, ctr: int, payload: bytes) -> np.ndarray:
        """+-1 chips before filtering: 63 preamble | 128 header | 1024 spread payload."""
        data_sy = 2.0 * polar_enc(payload, N=self.p.N, K=self.p.K).astype(np.float32) - 1.0
        lo16 = ctr & 0xFFFF
        hdr_bits = np.unpackbits(np.array([lo16 >> 8, lo16 & 0xFF], dtype=np.uint8))
        hdr_sy = (2.0 * np.repeat(hdr_bits, HDR_REPEAT).astype(np.float32) - 1.0) * self._hdr_pn_sy
        frame_len = self.p.preamble.size + HDR_L + data_sy.size
        pn = self.sec.pn_bits(ctr, frame_len)[self.p.preamble.size + HDR_L:]
        if pn.size != data_sy.size:
            raise RuntimeError(f"PN payload length {pn.size} != encoded payload {data_sy.size}")
        spread = data_sy * (2.0 * pn.astype(np.float32) - 1.0)
        return np.concatenate((self._preamble_sy, hdr_sy, spread)).astype(np.float32, copy=False)

    def _filter_frame(self, ctr: int, symbols: np.ndarray) -> np.ndarray:
        b, a = band_coeffs(BAND_PLAN[band_index(self._band_key, ctr)], self.p.fs)
        npre = self.p.preamble.size
        zi0 = np.zeros(max(len(a), len(b)) - 1, dtype=np.float64)
        y_pre, zi1 = lfilter(b, a, symbols[:npre], zi=zi0)        # zero state on the preamble ...
        y_rest, _ = lfilter(b, a, symbols[npre:], zi=zi1)         # ... carried into header + payload
        chips = np.concatenate((y_pre, y_rest))
        peak = float(np.max(np.abs(chips))) + EPS
        if peak > 3.0:
            chips = chips * (1.0 / peak)
        return chips.astype(np.float32, copy=False)

    def _make_frame_chips(self) -> np.ndarray:
        return self._filter_frame(self.frame_ctr, self._frame_symbols(self.frame_ctr, self._build_payload()))

    def _build_payload(self) -> bytes:
        meta = b"ESAL" + self.frame_ctr.to_bytes(4, "big") + self._session_nonce + secrets.token_bytes(11)
        blob = self.sec.seal(meta)
        assert len(meta) == 27 and len(blob) == 55
        return blob

    # ------------------------------------------------------------------ reproducible batches
    def make_frames(self, ctrs, payloads) -> np.ndarray:
        """float32 [len(ctrs), 1215]: frame i carries `payloads[i]` (55 bytes) under counter ctrs[i]."""
        out = np.empty((len(ctrs), self.p.preamble.size + HDR_L + self.p.N), dtype=np.float32)
        for i, (c, p) in enumerate(zip(ctrs, payloads)):
            out[i] = self._filter_frame(int(c), self._frame_symbols(int(c), bytes(p)))
        return out


def synthetic_payloads(sec: SecureChannel, ctrs, seed: int = 20260101) -> list[bytes]:
    """Sealed 55-byte payloads for the benchmark workloads: plaintext
    b"ESAL" | ctr_be32 | nonce8 | pad11 with nonce8 / pad11 / AEAD nonce from default_rng(seed)."""
    rng = np.random.default_rng(seed)
    out = []
    for c in ctrs:
        rnd = rng.integers(0, 256, size=8 + 11 + 12, dtype=np.uint8).tobytes()
        meta = b"ESAL" + int(c).to_bytes(4, "big") + rnd[:8] + rnd[8:19]
        out.append(sec.seal(meta, nonce=rnd[19:31]))
    return out
