#!/usr/bin/env python3
"""bench.py -- watermark frames/s decoded (sync + LLR + SCL-8) on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the receive hot path over one batch of synthetic frame records that is
already resident in HBM: band-pass + 63-chip NCC + median/MAD threshold + NMS  ->  _llr (variant 0,
known start/counter)  ->  Polar(1024,448) SCL-8 (validator None).  Workload at N=1 is BASELINE
config 2 (C2): 1 024 clean 1215-sample float32 frames from the embedder, key 0xAA*32, ctr = i,
payload RNG seed 20260101.  For N>1 every rank decodes its own 1 024-frame shard of the counter
range [0, 1024 N) (weak scaling); rank 0 derives the key/PN schedule and broadcasts it once over
RCCL before the timed region; the data path has no collective.

Extra objects in the JSON line:
  roofline     es_xcorr32_kernel (the HBM-graded correlation kernel: float32 screen whose decisions are
               settled exactly in float64 by es_pick_exact_kernel): algorithmic bytes per launch
               (9 472 B/frame: SURVEY.md section 8d) / its mean duration measured with HIP events
               on the launch stream inside the timed region, against the 8 TB/s HBM peak.
  cpu_baseline the CPU oracle (C restatement of the reference, kind "port") timed on this host.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

KEY = b"\xAA" * 32
XCORR_BYTES_PER_FRAME = 4 * 1215 + 4 * (1215 - 62)        # 9 472 B
HBM_PEAK_GBS = 8000.0                                       # MI355X_MICROARCH.md: 8.0 TB/s spec


def _cpu_worker(args):
    """Decode a slice of frames `reps` times with the CPU oracle (one process = one core)."""
    frames, band, pn, L, reps = args
    from echoseal_amd.tables import pack_tables
    from oracle import oracle as O
    ba, tpl, taps, ntaps, _ = pack_tables()
    for _ in range(reps):
        for i in range(frames.shape[0]):
            b = band[i]
            O.decode_frame(frames[i], ba[b], tpl[b], taps[b, :ntaps[b]], np.unpackbits(pn[i])[:1215], L=L)
    return frames.shape[0] * reps


def cpu_baseline(frames, band, pn, L, budget_s=15.0):
    import multiprocessing as mp
    from oracle import oracle as O
    O.build()
    cores = max(1, min(os.cpu_count() or 1, 16))
    t0 = time.perf_counter()
    _cpu_worker((frames[:8], band[:8], pn[:8], L, 1))
    per_frame = (time.perf_counter() - t0) / 8
    n = frames.shape[0] - frames.shape[0] % cores
    reps = max(1, int(round(budget_s * cores / max(per_frame * n, 1e-9))))     # ~budget_s of CPU work per core
    chunks = [(frames[i:n:cores], band[i:n:cores], pn[i:n:cores], L, reps) for i in range(cores)]
    ctx = mp.get_context("fork")
    with ctx.Pool(cores) as pool:
        t0 = time.perf_counter()
        done = sum(pool.map(_cpu_worker, chunks))
        dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n} of the {frames.shape[0]} workload frames x {reps} passes = {done} decodes, oracle/c "
                      f"(C restatement of the reference: sync + _llr + SCL-{L}), {cores} processes x 1 thread, {dt:.1f} s"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=1024, help="frame records per GPU per step (C2 = 1024)")
    ap.add_argument("--list-size", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N > 1 on one GPU)")
    ap.add_argument("--scl-streams", type=int, default=2, help="list-decoder streams (= batches in flight) of the pipeline")
    ap.add_argument("--depth", type=int, default=0, help="batches in flight (0: = --scl-streams)")
    ap.add_argument("--scl-multi", type=int, default=-1, help="es_set_option scl_multi: -1 auto, 0 one frame per wave, 1 several")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.backend != "nccl":
        local = local % max(1, torch.cuda.device_count())      # rehearsal: several ranks may share a GPU
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)

    from echoseal_amd.dist import broadcast_schedule, build_schedule, shard_range, split_schedule
    from echoseal_amd.embedder import WatermarkEmbedder, synthetic_payloads
    from echoseal_amd.engine import RxEngine

    B = a.frames
    total = B * world
    lo, hi = shard_range(total, rank, world)
    # inputs: each rank synthesises its own frames; the schedule comes from rank 0 over RCCL
    tx = WatermarkEmbedder(KEY)
    ctrs = list(range(lo, hi))
    frames_h = tx.make_frames(ctrs, synthetic_payloads(tx.sec, ctrs))
    sched = build_schedule(KEY, range(total)) if rank == 0 else None
    sched_d = broadcast_schedule(sched, total, dev)
    pn_d, band_d = split_schedule(sched_d, lo, hi)
    frames_d = torch.from_numpy(frames_h).to(dev)

    eng = RxEngine(local, list_size_max=max(8, a.list_size))
    L = a.list_size
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]

    # A step = one batch through the whole hot path.  Batches are independent, so the engine's streaming
    # pipeline (echoseal_amd.engine.DecodePipeline) keeps two in flight: the front end of batch k+2 (band-pass,
    # float32 correlation screen + exact float64 peak picking on one stream, LLR on a side stream) starts when
    # batch k leaves, beside the list decoder of batch k+1, and its own list decoder then runs beside that one
    # (a 1 024-frame list decoder puts ONE wave on every SIMD and leaves half of its issue slots idle; two of
    # them fill the vector unit).
    # thr / peaks are bit-identical to the float64 path.  Every step's outputs are complete at the final sync.
    from echoseal_amd.engine import DecodePipeline
    pipe = DecodePipeline(eng, list_size=L, scl_streams=a.scl_streams, depth=a.depth or None)
    for e in pipe.scl_engs:
        e.set_option("scl_multi", a.scl_multi)

    def step(k=None):
        sync_res, _llr, res, _done = pipe.submit(frames_d, band_d, pn_d, xcorr_events=None if k is None else ev[k])
        return res, sync_res.peaks, sync_res.npeaks

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    # latency of ONE batch with nothing else in flight (reported beside the pipelined throughput)
    lat = []
    for _ in range(3):
        t1 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t1)
    single_ms = 1e3 * min(lat)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(a.steps):
        res, peaks, npeaks = step(k)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    xcorr_ms = float(np.mean([s.elapsed_time(e) for s, e in ev]))

    # Supplementary (outside the timed region, rank 0 only): the same kernel on a BASELINE config-3
    # sized launch (65 536 records), where the launch is long enough to sit on the HBM roofline.
    big = None
    if rank == 0:
        def _launch_rate(xb, bb, bytes_per_record):
            # the kernel as it runs in the pipeline: the sync stage band-pass -> correlation screen -> exact peak
            # picking, round after round without idle gaps; HIP events around the correlation launch only.  (Timed
            # in a loop of nothing but band-pass + correlation, i.e. under sustained ~4 TB/s of HBM traffic, the same
            # launch takes 1.3-1.5x longer: tools/x32_data_dep.py.)
            evs = []
            for it in range(4):                      # enqueued back to back (no idle gaps: the clocks stay up)
                yb64, yb = eng.bpf2(xb, bb)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                cb = eng.xcorr32(yb, bb)
                e1.record()
                picked = eng.pick_exact(cb, yb64, bb)            # the rest of the sync stage, as in the pipeline
                evs.append((e0, e1))
                del yb64, yb, cb, picked
            torch.cuda.synchronize()
            ms = [e0.elapsed_time(e1) for e0, e1 in evs[1:]]
            m = float(np.mean(ms))
            return bytes_per_record * xb.shape[0] / (m * 1e-3) / 1e9, m
        Bb = 65536
        reps = -(-Bb // B)
        bb = band_d.repeat(reps)[:Bb].contiguous()
        # BASELINE config 3 shape: 65 536 windows of W = 2048 float32 samples (a frame somewhere inside, noise
        # elsewhere): 4*2048 B in + 4*1986 B out = 16 136 algorithmic bytes per window
        gen = torch.Generator(device=dev); gen.manual_seed(4)
        win = torch.randn((Bb, 2048), device=dev, dtype=torch.float32, generator=gen) * 0.05
        win[:, 400:400 + 1215] += frames_d.repeat(reps, 1)[:Bb]
        big, big_ms = _launch_rate(win, bb, 4 * 2048 + 4 * (2048 - 62))
        del win
    # sanity on the results of the last step (not timed): clean frames sync at offset 0
    ok_sync = bool(torch.all((npeaks >= 1) & (npeaks < 32)).item() and torch.all(peaks[:, 0] == 0).item())
    listed = int((res.ncand > 0).sum().item())

    if rank == 0:
        achieved = XCORR_BYTES_PER_FRAME * B / (xcorr_ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC passes committed under profiles/ (FETCH_SIZE doubled as
        # the gfx950 guide prescribes, WRITE_SIZE as read); measured on the 1024-record launch.
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_xcorr32_pmc_traffic.json")) as fh:
                pmc = json.load(fh)["es_xcorr32_kernel"]["B=1024 (C2 launch)"]
            if B == 1024:
                traffic = pmc["hbm_bytes_per_launch"]
        except Exception:
            traffic = None
        out = {
            "metric": "watermark frames/sec decoded (sync+LLR+SCL-8) @ 48 kHz",
            "value": total * a.steps / dt, "unit": "frames/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"C2: {B} clean 1215-sample float32 frames per GPU, key 0xAA*32, ctr=i, "
                                   f"payload seed 20260101; sync + _llr(variant 0, start 0) + SCL-{L}, validator None",
                       "frames_per_gpu": B, "list_size": L, "frame_len": 1215, "fs": 48000,
                       "sharding": f"{world} x {B} frames, schedule broadcast from rank 0",
                       "pipelining": "two batches in flight (DecodePipeline): the list decoders of consecutive steps overlap on two streams",
                       "single_batch_latency_ms": single_ms,
                       "sync_offsets_ok": ok_sync, "frames_through_list_decoder": listed},
            "roofline": {"kernel": "es_xcorr32_kernel", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "launch_ms": xcorr_ms, "algorithmic_bytes_per_launch": XCORR_BYTES_PER_FRAME * B},
            "roofline_c3": {"kernel": "es_xcorr32_kernel", "bound": "hbm", "achieved": big, "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": big / HBM_PEAK_GBS, "launch_ms": big_ms,
                            "note": "same kernel on a BASELINE config-3 sized launch, outside the timed region: 65 536 "
                                    "windows of 2 048 float32 samples, 16 136 algorithmic bytes per window; timed "
                                    "inside the sync stage (band-pass, screen, exact picking), mean of 3 rounds after a warm-up round"},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(frames_h, band_d.cpu().numpy(), pn_d.cpu().numpy(), L)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
