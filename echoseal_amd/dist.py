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
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
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
