"""Static tables the kernels consume: Butterworth coefficients, the cascaded preamble template,
matched-filter taps and the polar frozen mask.

These are one-off host computations (microseconds, cached per (band, fs)), done with the same
SciPy/NumPy calls as the reference so the numbers agree with it:
  butter_bandpass           rtwm/utils.py:52-55
  preamble template         rtwm/detector.py:67-69
  matched-filter taps       rtwm/detector.py:260-294
  frozen mask / data order  rtwm/fastpolar.py:220-230
"""
from __future__ import annotations

from functools import lru_cache

import numpy as np
from scipy.signal import lfilter

from .reliability import Q_NMAX_1024
from .utils import BAND_PLAN, butter_bandpass, mseq_63

MAX_TAPS = 576          # ES_MAX_TAPS: row stride of the tap table (the kernels' small-footprint instantiation serves up to 160)


@lru_cache(maxsize=None)
def band_coeffs(band: tuple[int, int], fs: int) -> tuple[np.ndarray, np.ndarray]:
    b, a = butter_bandpass(band[0], band[1], fs, order=4)
    return np.asarray(b, dtype=np.float64), np.asarray(a, dtype=np.float64)


@lru_cache(maxsize=None)
def preamble_template(band: tuple[int, int], fs: int) -> np.ndarray:
    """Unit-norm response of TX filter followed by RX filter to the +-1 MLS preamble."""
    b, a = band_coeffs(band, fs)
    chips = 2.0 * mseq_63().astype(np.float32) - 1.0
    twice = lfilter(b, a, lfilter(b, a, chips))
    return twice / (float(np.sqrt(np.sum(twice * twice))) + 1e-12)


@lru_cache(maxsize=None)
def matched_filter_taps(band: tuple[int, int], fs: int) -> np.ndarray:
    """float32 taps: time-reversed TX*RX impulse response cut at 99.9 % energy, unit energy."""
    b, a = band_coeffs(band, fs)
    span = max(256, 64 * max(len(a), len(b)))
    impulse = np.zeros(span, dtype=np.float32)
    impulse[0] = 1.0
    one_pass = lfilter(b, a, impulse).astype(np.float32)
    cascade = np.convolve(one_pass, one_pass).astype(np.float32)
    energy = np.cumsum(cascade * cascade)
    cut = int(np.searchsorted(energy, 0.999 * (float(energy[-1]) + 1e-20)))
    if cut + 1 < cascade.size:
        cascade = cascade[: cut + 1]
    taps = cascade[::-1]
    taps = taps / (np.sqrt(float(np.sum(taps * taps))) + 1e-12)
    taps = np.ascontiguousarray(taps, dtype=np.float32)
    if taps.size > MAX_TAPS:
        raise ValueError(f"matched filter needs {taps.size} taps; the kernels hold at most {MAX_TAPS}")
    return taps


def frozen_mask(N: int = 1024, K: int = 448) -> np.ndarray:
    """bool[N]: True = frozen.  The reference unfreezes the FIRST K entries of the reliability
    order, i.e. the K least reliable indices (rtwm/fastpolar.py:225-226) -- reproduced as is."""
    if N != 1024:
        raise ValueError("reliability table is defined for N = 1024 only")
    mask = np.ones(N, dtype=bool)
    mask[np.asarray(Q_NMAX_1024[:K], dtype=np.int64)] = False
    return mask


def pack_tables(fs: int = 48_000, K: int = 448):
    """Arrays in the layout es_set_tables() expects."""
    ba = np.zeros((4, 18), dtype=np.float64)
    tpl = np.zeros((4, 63), dtype=np.float64)
    taps = np.zeros((4, MAX_TAPS), dtype=np.float32)
    ntaps = np.zeros(4, dtype=np.int32)
    for i, band in enumerate(BAND_PLAN):
        b, a = band_coeffs(band, fs)
        ba[i, :9], ba[i, 9:] = b, a
        tpl[i] = preamble_template(band, fs)
        h = matched_filter_taps(band, fs)
        taps[i, : h.size] = h
        ntaps[i] = h.size
    return ba, tpl, taps, ntaps, frozen_mask(1024, K).astype(np.uint8)
