"""The multi-GPU layout on CPU: world_size-2 gloo processes shard the counter range, rank 0
broadcasts the key/PN schedule, every rank ends up with exactly the rows of its own shard."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from echoseal_amd.dist import ROW, broadcast_keys, broadcast_schedule, build_schedule, shard_range, split_schedule

KEY = b"\xAA" * 32


def test_shard_ranges_cover_everything_once():
    for n, world in ((1024, 1), (1024, 8), (1000, 8), (5, 8), (0, 4), (1 << 20, 8)):
        spans = [shard_range(n, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(hi - lo for lo, hi in spans) <= -(-n // world) if n else True


def test_schedule_rows_match_single_counter_calls():
    from echoseal_amd.crypto import SecureChannel
    from echoseal_amd.utils import band_index
    sched = build_schedule(KEY, [0, 5, 70000])
    sc = SecureChannel(KEY)
    for row, c in zip(sched, (0, 5, 70000)):
        assert np.array_equal(np.unpackbits(row[:152])[:1215], sc.pn_bits(c, 1215)) and row[152] == band_index(KEY, c)


def _worker(rank, world, port, n_total, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sched = build_schedule(KEY, range(n_total)) if rank == 0 else None
        buf = broadcast_schedule(sched, n_total, torch.device("cpu"))
        lo, hi = shard_range(n_total, rank, world)
        pn, band = split_schedule(buf, lo, hi)
        # the 48-byte alternative: only key material travels; each rank would expand its shard on its GPU
        keys = broadcast_keys(bytes(range(48)) if rank == 0 else None, torch.device("cpu"))
        assert keys == bytes(range(48))
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), pn=pn.numpy(), band=band.numpy(), lo=lo, hi=hi)
        # the data path has no collective: an all_gather here is only the test reading results back
        cnt = torch.tensor([hi - lo])
        dist.all_reduce(cnt)
        assert int(cnt) == n_total
    finally:
        dist.destroy_process_group()


def test_two_rank_broadcast_and_sharding(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    n_total, world = 301, 2
    mp.spawn(_worker, args=(world, port, n_total, str(tmp_path)), nprocs=world, join=True)
    full = build_schedule(KEY, range(n_total))
    seen = 0
    for r in range(world):
        z = np.load(tmp_path / f"r{r}.npz")
        lo, hi = int(z["lo"]), int(z["hi"])
        assert np.array_equal(z["pn"], full[lo:hi, :152]) and np.array_equal(z["band"], full[lo:hi, 152])
        seen += hi - lo
    assert seen == n_total and full.shape[1] == ROW
