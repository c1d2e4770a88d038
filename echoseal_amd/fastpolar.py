"""PolarCode with the reference's interface (rtwm/fastpolar.py:193-389).

* `encode` (TX side, 1024 XORs) is host NumPy -- it belongs to the frame generator, which is
  outside the accelerated path.
* `decode` is the hot path and runs ONLY on the HIP kernel (es_scl_batch): hard-decision shortcut
  and the SCL list are produced on the GPU, the validator callback is then applied on the host to
  the finished candidates in the reference's order (rtwm/fastpolar.py:268-276, 332-359).  Without
  the HIP library or a GPU it raises; there is no CPU decoder in this package.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Optional, Tuple

import numpy as np

from .reliability import Q_NMAX_1024

_engine = None


def default_engine():
    """Process-wide RxEngine on the current CUDA/HIP device (created on first use)."""
    global _engine
    if _engine is None:
        import torch
        from .engine import RxEngine
        # 256 = the detector's default list size (rtwm/detector.py:27); costs ~1.1 GB of scratch
        _engine = RxEngine(torch.cuda.current_device() if torch.cuda.is_available() else 0, list_size_max=256)
    return _engine


_other = None          # (K, engine) of the last code other than the reference's own that `decode` was asked for


def engine_for(K: int):
    """The engine `PolarCode(1024, K).decode` runs on: the process-wide one for K = 448, else one engine built for that K (one at a time:
    each holds ~1.1 GB of list-decoder scratch; asking for another K closes it)."""
    global _other
    if K == 448:
        return default_engine()
    if _other is None or _other[0] != K:
        import torch
        from .engine import RxEngine
        if _other is not None:
            _other[1].close()
        _other = (K, RxEngine(torch.cuda.current_device() if torch.cuda.is_available() else 0, list_size_max=256, code_k=K))
    return _other[1]


_other_fs = None       # (fs, engine) of the last fs_target other than 48 000 a detector was built for


def engine_for_fs(fs: int):
    """The engine a `WatermarkDetector(fs_target=fs)` runs on: the process-wide one for 48 000, else one engine whose tables (band-pass,
    preamble template, matched-filter taps) are built for that rate (one at a time, like `engine_for`)."""
    global _other_fs
    if fs == 48_000:
        return default_engine()
    if _other_fs is None or _other_fs[0] != fs:
        import torch
        from .engine import RxEngine
        if _other_fs is not None:
            _other_fs[1].close()
        _other_fs = (fs, RxEngine(torch.cuda.current_device() if torch.cuda.is_available() else 0, list_size_max=256, fs=fs))
    return _other_fs[1]


def _reliability_order(N: int) -> np.ndarray:
    rel = np.asarray(Q_NMAX_1024, dtype=np.int64)
    if rel.size != N:
        raise ValueError(f"Q_Nmax must have {N} entries (has {rel.size})")
    return rel


@dataclass
class PolarCode:
    N: int
    K: int
    list_size: int = 8
    crc_size: int = 8
    debug: bool = False

    frozen: np.ndarray = field(init=False, repr=False, default=None)
    _data_pos: np.ndarray = field(init=False, repr=False, default=None)
    _info_len: int = field(init=False, repr=False, default=0)
    _crc_poly: int = field(init=False, repr=False, default=0x07)

    def __post_init__(self) -> None:
        if self.N <= 0 or (self.N & (self.N - 1)) != 0:
            raise ValueError("N must be a power of 2 and > 0")
        if not (0 < self.K <= self.N):
            raise ValueError("0 < K <= N must hold")
        if self.list_size < 1:
            raise ValueError("list_size must be >= 1")
        if not (0 < self.crc_size < self.K):
            raise ValueError("0 < crc_size < K must hold")
        rel = _reliability_order(self.N)
        self.frozen = np.ones(self.N, dtype=bool)
        self.frozen[rel[: self.K]] = False           # reference quirk: the K LEAST reliable indices
        self._data_pos = np.flatnonzero(~self.frozen)
        self._info_len = self.K - self.crc_size

    # ------------------------------------------------------------------ TX side (host)
    def _crc8(self, bits: np.ndarray) -> np.ndarray:
        reg = 0
        for bit in np.asarray(bits, dtype=np.uint8):
            reg ^= (int(bit) & 1) << 7
            reg = ((reg << 1) ^ self._crc_poly) & 0xFF if reg & 0x80 else (reg << 1) & 0xFF
        return np.unpackbits(np.array([reg], dtype=np.uint8))

    def _crc_ok(self, info: np.ndarray, crc_bits: np.ndarray) -> bool:
        return bool(np.array_equal(self._crc8(info), np.asarray(crc_bits, dtype=np.uint8)))

    @staticmethod
    def _polar_transform(u: np.ndarray) -> np.ndarray:
        x = np.array(u, dtype=np.uint8, copy=True)
        n = x.size
        half = 1
        while half < n:
            blk = x.reshape(-1, 2, half)
            blk[:, 0, :] ^= blk[:, 1, :]
            half *= 2
        return x

    def encode(self, info_bits: np.ndarray) -> np.ndarray:
        info_bits = np.asarray(info_bits)
        if info_bits.dtype != np.uint8:
            info_bits = info_bits.astype(np.uint8)
        if info_bits.ndim != 1:
            raise ValueError("info_bits must be a 1D array")
        if info_bits.size != self._info_len:
            raise ValueError(f"info_bits must have length {self._info_len}")
        u = np.zeros(self.N, dtype=np.uint8)
        u[self._data_pos] = np.concatenate((info_bits, self._crc8(info_bits)))
        return self._polar_transform(u)

    # ------------------------------------------------------------------ RX side (GPU only)
    def decode(self, llr: np.ndarray, validator: Optional[Callable[[bytes], bool]] = None) -> Tuple[np.ndarray, bool]:
        llr = np.asarray(llr)
        if llr.ndim != 1 or llr.size != self.N:
            raise ValueError(f"llr must be 1D length {self.N}")
        if self.N != 1024 or self.crc_size != 8:
            raise NotImplementedError("the HIP decoder serves Polar(1024, K) + CRC-8 (DESIGN.md section 7)")
        import torch
        from .engine import select_payload
        eng = engine_for(self.K)
        if self.list_size > eng.list_size_max:
            raise NotImplementedError(f"list_size={self.list_size}: the HIP decoder supports list sizes up to {eng.list_size_max}")
        host = np.ascontiguousarray(llr, dtype=np.float32 if llr.dtype == np.float32 else np.float64)
        dev = torch.from_numpy(host).to(eng.device).reshape(1, self.N)
        res = eng.scl(dev, list_size=self.list_size, skip_if_hard_ok=(validator is None))
        payload, ok = select_payload(res, 0, validator)
        return np.unpackbits(np.frombuffer(payload, dtype=np.uint8))[: self._info_len], ok
