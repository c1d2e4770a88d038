"""Host-side helpers with the reference's names (mirror of rtwm/utils.py).

Band plan and keyed hop (rtwm/utils.py:19-36), dB helpers (:40-48), Butterworth design and
resampling wrappers (:52-66), the AES-based PN stream (:83-132) and the 63-chip MLS (:135-145).
None of this is hot-path arithmetic; it produces the tables and the key/PN schedule the HIP
kernels consume.  The AES / HMAC code is our own (echoseal_amd.primitives) because neither
`cryptography` nor PyCryptodome is available where this runs.
"""
from __future__ import annotations

import hashlib
import hmac
import math
from typing import Tuple

import numpy as np
from scipy.signal import butter, resample_poly

from .primitives import aes128_encrypt_blocks

BAND_PLAN: list[Tuple[int, int]] = [
    (4_000, 6_000),
    (8_000, 10_000),
    (16_000, 18_000),
    (18_000, 22_000),
]


def band_index(key: bytes, frame_ctr: int) -> int:
    """Index into BAND_PLAN for a frame counter: HMAC-SHA256(key, ctr_be32)[0] mod 4 (rtwm/utils.py:27-36).  Not memoised here: a
    process-wide cache would keep the secret hop key alive after its detector is gone (see BandHop)."""
    tag = hmac.new(bytes(key), int(frame_ctr).to_bytes(4, "big"), hashlib.sha256).digest()
    return tag[0] % len(BAND_PLAN)


class BandHop:
    """The hop schedule of ONE key, memoised per owner: the counter search of one verify() asks for several hundred counters, the same
    ones clip after clip (2 ms of HMACs per call on the host otherwise).  Lives and dies with the object that owns it (a
    WatermarkDetector), holds at most `limit` counters, and never leaves key material in module state."""

    def __init__(self, key: bytes, limit: int = 1 << 16) -> None:
        self._key, self._limit, self._memo = bytes(key), int(limit), {}

    def index(self, frame_ctr: int) -> int:
        c = int(frame_ctr)
        v = self._memo.get(c)
        if v is None:
            if len(self._memo) >= self._limit:
                self._memo.clear()
            v = self._memo[c] = band_index(self._key, c)
        return v

    def band(self, frame_ctr: int) -> tuple[int, int]:
        return BAND_PLAN[self.index(frame_ctr)]


def choose_band(key: bytes, frame_ctr: int) -> tuple[int, int]:
    return BAND_PLAN[band_index(key, frame_ctr)]


def db_to_lin(db: float) -> float:
    return 10.0 ** (db / 20.0)


def lin_to_db(lin: float) -> float:
    return 20.0 * np.log10(lin + 1e-12)


def butter_bandpass(lo: float, hi: float, fs: int, *, order: int = 4):
    nyq = 0.5 * fs
    return butter(order, [lo / nyq, hi / nyq], "band")


def resample_to(fs_target: int, audio: np.ndarray, fs_orig: int) -> tuple[np.ndarray, int]:
    if fs_orig == fs_target:
        return audio, fs_orig
    g = math.gcd(fs_orig, fs_target)
    return resample_poly(audio, fs_target // g, fs_orig // g), fs_target


def resample_plan(n_in: int, up: int, down: int, dtype):
    """Everything scipy.signal.resample_poly(x, up, down) does in Python before its compiled loop (SciPy 1.15: Kaiser-5.0
    `firwin` design of 20*max(up,down)+1 taps scaled by `up`, zero padding that centres the output, the kept output
    range) plus upfirdn's transposed / flipped polyphase layout.  -> (h_tf, taps per phase, up, down, first kept output,
    number of outputs, compute dtype), or None when the rates are equal.  The loop itself is es_resample_batch."""
    from scipy.signal import firwin
    g = math.gcd(int(up), int(down))
    up, down = int(up) // g, int(down) // g
    if up == down == 1:
        return None
    n_out = n_in * up
    n_out = n_out // down + bool(n_out % down)
    max_rate = max(up, down)
    half_len = 10 * max_rate
    h = firwin(2 * half_len + 1, 1.0 / max_rate, window=("kaiser", 5.0))
    if np.issubdtype(np.dtype(dtype), np.floating):
        h = h.astype(dtype)
    h *= up
    n_pre_pad = down - half_len % down
    n_post_pad = 0
    n_pre_remove = (half_len + n_pre_pad) // down

    def _output_len(len_h: int) -> int:
        nt = (n_in + (len_h + (-len_h % up)) // up - 1) * up
        return nt // down + (1 if nt % down > 0 else 0)

    while _output_len(len(h) + n_pre_pad + n_post_pad) < n_out + n_pre_remove:
        n_post_pad += 1
    h = np.concatenate((np.zeros(n_pre_pad, h.dtype), h, np.zeros(n_post_pad, h.dtype)))
    ctype = np.result_type(h.dtype, np.dtype(dtype), np.float32)
    padlen = len(h) + (-len(h) % up)
    hf = np.zeros(padlen, ctype)
    hf[: len(h)] = h
    h_tf = np.ascontiguousarray(hf.reshape(-1, up).T[:, ::-1].ravel())
    return h_tf, padlen // up, up, down, n_pre_remove, n_out, np.dtype(ctype)


class StreamPRNG:
    """AES-128 block stream: block j of frame c is AES(sub_key, (c << 64 | j) big-endian)."""

    def __init__(self, master_key: bytes):
        self._sub_key = hashlib.blake2s(master_key, digest_size=16, person=b"EchoSeal").digest()

    @property
    def sub_key(self) -> bytes:
        return self._sub_key

    def blocks(self, frame_ctrs, n_blocks: int) -> np.ndarray:
        """Keystream for many counters at once -> uint8 [len(frame_ctrs), 16 * n_blocks]."""
        ctrs = np.asarray(frame_ctrs, dtype=np.uint64).reshape(-1)
        inp = np.zeros((ctrs.size, n_blocks, 16), dtype=np.uint8)
        inp[:, :, 0:8] = ctrs.astype(">u8").view(np.uint8).reshape(-1, 1, 8)
        inp[:, :, 8:16] = np.arange(n_blocks, dtype=">u8").view(np.uint8).reshape(1, n_blocks, 8)
        return aes128_encrypt_blocks(self._sub_key, inp).reshape(ctrs.size, 16 * n_blocks)

    def bytes(self, frame_ctr: int, n: int = 64) -> bytes:
        return self.blocks([frame_ctr], (n + 15) // 16)[0, :n].tobytes()


def pn_bits(prng: StreamPRNG, frame_ctr: int, n_bits: int) -> np.ndarray:
    data = prng.bytes(frame_ctr, (n_bits + 7) // 8)
    return np.unpackbits(np.frombuffer(data, dtype="u1"))[:n_bits]


def mseq_63() -> np.ndarray:
    """63-chip maximal-length sequence: 6-stage LFSR, feedback bit5 ^ bit4, seed 0b111111."""
    state = 0b111111
    out = np.empty(63, dtype=np.uint8)
    for i in range(63):
        out[i] = state & 1
        fb = ((state >> 5) ^ (state >> 4)) & 1
        state = ((state << 1) & 0b111111) | fb
    return out
