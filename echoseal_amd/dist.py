"""Multi-GPU layout of the receive path: one process per GPU, frames sharded, one broadcast.

Frames are independent, so the data path has NO collective: rank r owns the
contiguous record range shard_range(B, r, world).  The only exchange is the key/PN schedule --
the packed PN rows (152 B) and band index (1 B) per frame counter -- which rank 0 derives from the
master key and broadcasts once (RCCL broadcast over xGMI when the backend is "nccl"; each peer is
one direct hop from the root).  Works with any torch.distributed backend, so the logic is covered
on CPU with gloo (tests/test_dist.py).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

ROW = 153          # 152 packed PN bytes + 1 band byte


def shard_range(n_total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous ceil(n/world)-sized shards; trailing ranks may be short or empty."""
    per = -(-n_total // world)
    lo = min(n_total, rank * per)
    return lo, min(n_total, lo + per)


def build_schedule(key32: bytes, ctrs) -> np.ndarray:
    """uint8 [len(ctrs), 153]: PN row | band index -- derived from the key on ONE rank."""
    from .crypto import SecureChannel
    from .utils import band_index
    ctrs = list(ctrs)
    out = np.empty((len(ctrs), ROW), dtype=np.uint8)
    out[:, :152] = SecureChannel(key32).pn_bytes_batch(ctrs, 152)
    out[:, 152] = [band_index(key32, c) for c in ctrs]
    return out


def broadcast_schedule(schedule: np.ndarray | None, n_total: int, device: torch.device, src: int = 0) -> torch.Tensor:
    """Root passes the [n_total, 153] schedule, peers pass None; everyone gets the device tensor."""
    buf = torch.empty((n_total, ROW), dtype=torch.uint8, device=device)
    if dist.is_available() and dist.is_initialized():             # (a world of one included: the same calls, trivially)
        if dist.get_rank() == src:
            buf.copy_(torch.from_numpy(np.ascontiguousarray(schedule)))
        dist.broadcast(buf, src=src)
    else:
        buf.copy_(torch.from_numpy(np.ascontiguousarray(schedule)))
    return buf


def split_schedule(buf: torch.Tensor, lo: int, hi: int) -> tuple[torch.Tensor, torch.Tensor]:
    """-> (pn_rows [n,152], band [n]) views of this rank's shard, contiguous."""
    mine = buf[lo:hi]
    return mine[:, :152].contiguous(), mine[:, 152].contiguous()


def broadcast_keys(keys48: bytes | None, device: torch.device, src: int = 0) -> bytes:
    """Alternative to broadcasting the expanded schedule: the root passes the 48 bytes of key material
    (16-byte AES PN sub-key | 32-byte hop key), peers pass None; every rank then derives the rows of its own shard
    on its GPU with `RxEngine.schedule` (es_schedule_batch).  One 48-byte broadcast instead of 153 B per counter."""
    buf = torch.zeros(48, dtype=torch.uint8, device=device)
    if dist.is_available() and dist.is_initialized():             # (a world of one included: the same calls, trivially)
        if dist.get_rank() == src:
            buf.copy_(torch.frombuffer(bytearray(keys48), dtype=torch.uint8))
        dist.broadcast(buf, src=src)
    else:
        buf.copy_(torch.frombuffer(bytearray(keys48), dtype=torch.uint8))
    return bytes(buf.cpu().numpy().tobytes())


def derive_shard_schedule(eng, keys48: bytes, lo: int, hi: int) -> tuple[torch.Tensor, torch.Tensor]:
    """(pn_rows [hi-lo,152], band [hi-lo]) of counters lo..hi-1, generated on the device from the 48 key bytes."""
    return eng.schedule(keys48[:16], keys48[16:], ctr0=lo, n=hi - lo)
