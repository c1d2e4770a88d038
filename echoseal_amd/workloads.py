"""Synthetic workloads of BASELINE.json's configurations, shared by bench.py, the tests and
the fixture generators under oracle/refshim/ so that all of them see the same inputs.

    C2  B clean 1215-sample float32 frames, key 0xAA*32, ctr = i, payload RNG seed 20260101
    C3  one frame per W = 2048-sample window: resampled by a factor uniform in [0.95, 1.05] (LINEAR interpolation,
        output sample k = frame(k * fac)), placed at a uniform offset, plus white Gaussian noise at -15 dB SNR
        relative to the resampled frame's RMS (sigma = rms * 10^(15/20)); factor / offset from default_rng(3),
        noise from default_rng(4)
    C4  = C2 generator over ctr 0 .. 2^20-1, sharded
    C5  needs an MP3 codec (absent from the image); `lossy_channel` below is the documented SURROGATE -- it is NOT
        MP3: band-limit to 16 kHz (the 128 kbps encoders' default low-pass), 1 % of full-scale quantisation noise
        shaped like the signal's short-time level (codec delay compensated by default).

Host NumPy here; `c3_windows_device` builds the same kind of windows from device frames with torch ops (bench input
synthesis at 65 536 windows -- tensor plumbing, not part of the decoded path; it draws its own random numbers).
"""
from __future__ import annotations

import numpy as np

KEY = b"\xAA" * 32
FRAME_LEN = 1215
C3_WINDOW = 2048
C3_SNR_DB = -15.0


def c2_frames(ctrs, key: bytes = KEY, seed: int = 20260101):
    """-> (frames float32 [n,1215], band uint8 [n], pn uint8 [n,152], payloads list[bytes])."""
    from .embedder import WatermarkEmbedder, synthetic_payloads
    from .utils import band_index
    tx = WatermarkEmbedder(key)
    ctrs = [int(c) for c in ctrs]
    payloads = synthetic_payloads(tx.sec, ctrs, seed)
    frames = tx.make_frames(ctrs, payloads)
    band = np.array([band_index(key, c) for c in ctrs], np.uint8)
    pn = tx.sec.pn_bytes_batch(ctrs, 152)
    return frames, band, pn, payloads


def c3_windows(frames: np.ndarray, *, W: int = C3_WINDOW, snr_db: float = C3_SNR_DB, seed_geom: int = 3, seed_noise: int = 4):
    """C3 windows from clean frames [n,1215] -> (win float32 [n,W], offset int32 [n], factor float64 [n], m int32 [n])."""
    n = frames.shape[0]
    rng3, rng4 = np.random.default_rng(seed_geom), np.random.default_rng(seed_noise)
    win = np.zeros((n, W), np.float32)
    offs = np.zeros(n, np.int32); facs = np.zeros(n, np.float64); lens = np.zeros(n, np.int32)
    T = frames.shape[1]
    for i in range(n):
        fac = rng3.uniform(0.95, 1.05)
        m = int(np.floor((T - 1) / fac)) + 1
        res = np.interp(np.arange(m) * fac, np.arange(T), frames[i]).astype(np.float32)
        off = int(rng3.integers(0, W - m + 1))
        win[i, off:off + m] = res
        rms = float(np.sqrt(np.mean(res.astype(np.float64) ** 2)))
        win[i] += rng4.normal(0.0, rms * 10 ** (-snr_db / 20), W).astype(np.float32)
        offs[i], facs[i], lens[i] = off, fac, m
    return win, offs, facs, lens


def c3_windows_device(frames_d, *, W: int = C3_WINDOW, snr_db: float = C3_SNR_DB, seed: int = 34):
    """Same construction with torch ops on the frames' device: frames_d float32 [n,1215] -> (win [n,W], offset [n])."""
    import torch
    n, T = frames_d.shape
    dev = frames_d.device
    g = torch.Generator(device=dev); g.manual_seed(seed)
    fac = 0.95 + 0.10 * torch.rand(n, device=dev, dtype=torch.float64, generator=g)
    m = torch.floor((T - 1) / fac).to(torch.int64) + 1                           # resampled length
    off = torch.floor(torch.rand(n, device=dev, dtype=torch.float64, generator=g) * (W - m + 1).to(torch.float64)).to(torch.int64)
    k = torch.arange(W, device=dev, dtype=torch.int64)[None, :] - off[:, None]      # index inside the resampled frame
    inside = (k >= 0) & (k < m[:, None])
    pos = k.clamp(min=0).to(torch.float64) * fac[:, None]
    i0 = pos.floor().clamp(max=T - 2).to(torch.int64)
    w = (pos - i0.to(torch.float64)).clamp(0.0, 1.0)
    f64 = frames_d.to(torch.float64)
    res = torch.gather(f64, 1, i0) * (1.0 - w) + torch.gather(f64, 1, i0 + 1) * w
    res = torch.where(inside, res, torch.zeros((), device=dev, dtype=torch.float64))
    rms = torch.sqrt((res * res).sum(1) / m.to(torch.float64))
    noise = torch.randn((n, W), device=dev, dtype=torch.float32, generator=g)
    win = res.to(torch.float32) + noise * (rms * 10 ** (-snr_db / 20)).to(torch.float32)[:, None]
    return win.contiguous(), off.to(torch.int32)


def lossy_channel(x: np.ndarray, *, seed: int = 5, fs: int = 48_000, delay: int = 0) -> np.ndarray:
    """Config-5 SURROGATE (NOT MP3; there is no codec in the image): 16 kHz low-pass (101-tap Hamming FIR, the default
    low-pass of 128 kbps encoders), additive noise at 1 % of the short-time (576-sample granule) RMS level -- a
    stand-in for perceptually shaped quantisation noise; the FIR's own 50-sample delay is removed and `delay` (default 0 =
    a decoder that compensates the codec delay; 1105 = an uncompensated LAME round trip) is applied, length preserved."""
    from scipy.signal import firwin, lfilter
    x = np.asarray(x, np.float32)
    h = firwin(101, 16_000, fs=fs).astype(np.float32)
    y = lfilter(h, [1.0], x.astype(np.float64), axis=-1)
    g = 576
    n = y.shape[-1]
    pad = (-n) % g
    yp = np.concatenate((y, np.zeros(y.shape[:-1] + (pad,))), axis=-1).reshape(y.shape[:-1] + (-1, g))
    lvl = np.repeat(np.sqrt(np.mean(yp * yp, axis=-1)), g, axis=-1)[..., :n]
    y = y + 0.01 * lvl * np.random.default_rng(seed).normal(size=y.shape)
    shift = delay - 50                                            # the FIR already delays by 50 samples
    out = np.zeros_like(y)
    if shift >= 0:
        if n > shift:
            out[..., shift:] = y[..., :n - shift]
    else:
        out[..., :n + shift] = y[..., -shift:]
    return out.astype(np.float32)
