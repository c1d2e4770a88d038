"""PCM16 WAV ingest for the receive path ("int16 / WAV ingest").

The reference's file entry point is `soundfile.read` in rx_app.py:25-28 (soundfile is not in this image and the CLI is out
of scope); what the receive path needs from it is the sample array and the rate.  `read_wav` returns the PCM16 samples as
int16 -- the band-pass kernels ingest int16 directly and dequantise as x / 32768, which is exactly the float value
soundfile hands the reference for a PCM16 file -- so a 16-bit recording goes from disk to the GPU at 2 bytes per sample.
Standard library only (`wave`).
"""
from __future__ import annotations

import wave

import numpy as np


def read_wav(path: str, *, channel: int | None = 0):
    """-> (samples, fs).  16-bit PCM: int16 [n] (one channel; channel=None averages the channels the way a mono mix-down
    would and returns float32).  8-bit / 24-bit / 32-bit PCM are converted to float32 in [-1, 1)."""
    with wave.open(path, "rb") as w:
        nch, width, fs, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        raw = w.readframes(n)
    if width == 2:
        x = np.frombuffer(raw, dtype="<i2").reshape(-1, nch)
        if channel is None and nch > 1:
            return (x.astype(np.float32) / np.float32(32768.0)).mean(axis=1).astype(np.float32), fs
        return np.ascontiguousarray(x[:, 0 if channel is None else channel]), fs
    if width == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    elif width == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        x = ((v ^ 0x800000) - 0x800000).astype(np.float32) / np.float32(8388608.0)
    elif width == 4:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float32) / np.float32(2147483648.0)
    else:
        raise ValueError(f"unsupported sample width {width}")
    x = x.reshape(-1, nch)
    return (x.mean(axis=1) if channel is None else x[:, channel]).astype(np.float32), fs


def write_wav_pcm16(path: str, samples: np.ndarray, fs: int) -> None:
    """Mono PCM16 file from int16 samples, or from floats in [-1, 1) (rounded to nearest, clipped)."""
    x = np.asarray(samples)
    if x.dtype != np.int16:
        x = np.clip(np.round(x.astype(np.float64) * 32768.0), -32768, 32767).astype(np.int16)
    with wave.open(path, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(int(fs))
        w.writeframes(x.astype("<i2").tobytes())
