"""polar_fast: convenience wrappers with the reference's signatures (rtwm/polar_fast.py:26-87)."""
from __future__ import annotations

import logging
from typing import Callable, Optional, Tuple

import numpy as np

from .fastpolar import PolarCode

N_DEFAULT = 1024
K_DEFAULT = 448

_cache: dict[tuple[int, int, int, int], PolarCode] = {}


def _pc(N: int, K: int, list_size: int, crc_size: int) -> PolarCode:
    key = (N, K, list_size, crc_size)
    pc = _cache.get(key)
    if pc is None:
        pc = _cache[key] = PolarCode(N, K, list_size=list_size, crc_size=crc_size)
    return pc


def encode(payload: bytes, *, N: int = N_DEFAULT, K: int = K_DEFAULT, list_size: int = 8, crc_size: int = 8,
           debug: bool = False) -> np.ndarray:
    pc = _pc(N, K, list_size, crc_size)
    want = (pc.K - pc.crc_size) // 8
    if len(payload) != want:
        raise ValueError(f"payload must be {want} bytes (got {len(payload)})")
    bits = np.unpackbits(np.frombuffer(payload, dtype="u1"))
    if debug:
        logging.debug("[ENCODE] payload_hex=%s", payload.hex())
    return pc.encode(bits)


def decode(llr: np.ndarray, *, N: int = N_DEFAULT, K: int = K_DEFAULT, list_size: int = 8, crc_size: int = 8,
           return_ok: bool = False, debug: bool = False,
           validator: Optional[Callable[[bytes], bool]] = None) -> Optional[bytes] | Tuple[bytes, bool]:
    pc = _pc(N, K, list_size, crc_size)
    llr = np.asarray(llr)
    if llr.ndim != 1 or llr.size != pc.N:
        raise ValueError(f"LLR length {llr.size} != N {pc.N}")
    bits, ok = pc.decode(llr, validator=validator)
    if debug:
        logging.debug("[DECODE] ok=%s", ok)
    payload = np.packbits(bits).tobytes()
    if return_ok:
        return payload, ok
    return payload if ok else None
