#!/usr/bin/env python3
"""bench.py -- watermark frames/s decoded (sync + LLR + SCL-8) on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` without a launcher (WORLD_SIZE unset) starts N rank processes itself -- fresh children,
before anything in the parent touches the GPU -- polls all of them, and tears the job down on the first failure.

Headline (`value`): BASELINE config 3 (C3), the largest single-GPU configuration.  One step = one pass of the receive hot
path over one batch of 65 536 synthetic windows per GPU, already resident in HBM: each window is 2 048 float32 samples
holding one frame (ctr = i, key 0xAA*32) resampled by U[0.95, 1.05], at a uniform offset, in AWGN at -15 dB;
band-pass -> fused sync (float32 NCC screen in LDS + exact median/MAD threshold + NMS / top-5, one kernel) -> _llr (variant 0)
at the DETECTED peak -> Polar(1024,448) SCL-8 (validator None) -> candidate selection.  One window = one frame record, so
`value` is frames/s.  Steps rotate over four pipeline lanes (HIP streams, one context each), so that the front ends of some steps run
beside the list decoders of others.  N > 1: weak scaling -- rank r decodes its own 65 536 windows (counters
[r * 65 536, (r + 1) * 65 536)); rank 0 derives the key/PN schedule of all counters and broadcasts it once (RCCL) before the
timed region; the data path has no collective.  At --steps 20 the timed region is ~0.5 s of steady work.

Further driver-timed legs in the same JSON line (`legs`), each bracketed by barrier + synchronize like the headline:
  c2        BASELINE config 2: batches of 1 024 clean 1215-sample frames through the grouped streaming pipeline (the front ends of
            16 batches on four streams fill one LLR buffer, one list-decoder launch per group), --c2-steps batches.
  c2_lanes  the same batches through seven whole-chain lanes (one list-decoder launch per batch): rows within ~3 ms of submission.
  c2_single one 1 024-frame batch alone, submission to completion (latency, not throughput).
  c3_unfused the headline's pass with the correlation row going through HBM (es_xcorr32_batch -> es_pick_exact_batch): identical
            results; `roofline` is the stand-alone correlation kernel INSIDE this leg (HIP events on its launch stream).
  c3_pcie   the headline with every step's samples copied from pinned host memory over PCIe (copy stream, overlapped): the rate when the
            boundary hands over host buffers.  Reported beside the headline, never as `value`.
  c3_sustained  the headline's steps for ~3 s on end (--sustained-steps, default 150) instead of 20: the rate the chip holds once clocks and
            temperatures have settled, beside the 0.4 s burst the contract's K = 20 times.
  c4        BASELINE config 4, strong scaling: 2^20 frames in total, ctr 0 .. 2^20-1, sharded contiguously over the N ranks; rank 0
            derives the whole schedule (153 B per counter = 160 MB) and broadcasts it; a checksum over (frame index, payload, ok,
            sync offset) summed over ranks is the same number at every N.
  c5        BASELINE config 5 SURROGATE (no MP3 codec in the image -- echoseal_amd/workloads.lossy_channel; NOT MP3):
            list size swept over 1/4/8/16, payload bit error rate and frames/s.

Extra objects:
  roofline      es_xcorr32_kernel<17,2048> -- the kernel north_star grades against the HBM roofline -- in the timed c3_unfused leg:
                algorithmic bytes per launch (16 136 B per window) / mean launch duration against 8 TB/s;
                `traffic` from the committed PMC passes.  NOT on the headline path (the product's sync is the fused kernel): labelled so.
  roofline_fused  the fused sync kernel (8 192 + 150 B per window), timed as a stand-alone launch (beside the list decoders of other lanes its
                events span their time): LDS- and float64-bound.
  roofline_scl  the kernel that dominates the time (list decoder, ~88 % of a step), at the HEADLINE's own rate: vector instructions,
                float64 instructions and HBM-side bytes per frame from the committed PMC passes (profiles/r04_scl_pmc.json).
  cpu_baseline  the CPU oracle (C restatement of the reference, kind "port") timed on this host on a bounded sample of the headline's
                windows (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The pipelines run on up to eight HIP streams; the HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues
# (default 4), and streams that share a queue serialise.  Must be set before HIP starts (echoseal_amd.engine does the same on import).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

KEY = b"\xAA" * 32
XCORR_BYTES_PER_WINDOW = 4 * 2048 + 4 * (2048 - 62)       # 16 136 B
FUSED_BYTES_PER_WINDOW = 4 * 2048 + 150                    # fused sync: samples in, thr / peaks / npeaks / flag out
SCL_BYTES_PER_FRAME = 4096 + 520                           # list decoder: LLRs in, <= 8 x (55 + 8 + 1) + hard decision out
HBM_PEAK_GBS = 8000.0                                       # MI355X_MICROARCH.md: 8.0 TB/s spec
# FP64 vector peak: 78.6 TFLOP/s (spec) = 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz, i.e. one wave64 float64 instruction per
# 4 cycles and SIMD -> 1024 SIMDs x 0.6 G = 614.4 G wave-instructions/s
FP64_ISSUE_PEAK_GWIPS = 256 * 4 * 2.4 / 4.0
CLOCK_GHZ = 2.4
N_SIMD = 1024


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--windows", type=int, default=65536, help="C3 windows per GPU per step (BASELINE config 3 = 65 536)")
    ap.add_argument("--list-size", type=int, default=8)
    ap.add_argument("--legs", default="auto", help="comma list of c2,c2_lanes,c3_unfused,c3_pcie,c3_sustained,c4,c5 (auto: all at N = 1, c2 + c4 at N > 1; none: headline only)")
    ap.add_argument("--sustained-steps", type=int, default=150, help="steps of the c3_sustained leg (~19 ms each)")
    ap.add_argument("--big-lanes", type=int, default=8, help="pipeline lanes of the headline and of the c4 leg (launches of 65 536 records): up to 8")
    ap.add_argument("--frames", type=int, default=1024, help="frame records per batch of the c2 legs (C2 = 1024)")
    ap.add_argument("--c2-steps", type=int, default=480)
    ap.add_argument("--group", type=int, default=16, help="c2: batches per list-decoder launch of the grouped pipeline")
    ap.add_argument("--front-lanes", type=int, default=4, help="c2: front-end streams of the grouped pipeline")
    ap.add_argument("--scl-streams", type=int, default=3, help="c2: list-decoder streams of the grouped pipeline")
    ap.add_argument("--c4-frames", type=int, default=1 << 20, help="total frames of the strong-scaling leg (all ranks together)")
    ap.add_argument("--c4-chunk", type=int, default=65536)
    ap.add_argument("--c5-frames", type=int, default=16384)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-collectives", action="store_true",
                    help="N = 1 only: initialise the process group (RCCL for --backend nccl) at world size 1 and run every collective of the N > 1 "
                         "path (schedule broadcast, barrier, all_reduce) through it -- lets one GPU load librccl and exercise the calls")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N > 1 on fewer GPUs)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ self-launch
def launch_ranks(a, *, poll_s: float = 0.05) -> int:
    """Parent of an N > 1 run started without a launcher: spawn N fresh rank processes (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in their environment) and pass rank 0's JSON line through.  The parent never initialises the GPU
    (torch.cuda.device_count() does not, on this image), and the children are new processes, not re-execs.
    ALL children are polled: the first non-zero exit tears the job down (a dead rank leaves its peers waiting in a collective
    until RCCL's own timeout) and becomes the parent's exit code."""
    import torch
    ndev = torch.cuda.device_count()
    if a.backend == "nccl" and ndev < a.gpus:
        print(f"bench.py: --gpus {a.gpus} but only {ndev} GPU(s) are visible", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        # HSA_ENABLE_IPC_MODE_LEGACY=0: this pool's host driver supports only dmabuf IPC; with the legacy mode RCCL's (and torch's)
        # cross-process buffer sharing fails with `hipIpcGetMemHandle: invalid argument`.  The image exports it; a caller's own value wins.
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        live = set(range(a.gpus))
        while live and rc == 0:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0:
                    rc = abs(code) or 1
                    print(f"bench.py: rank {r} exited with code {code}; stopping the other {len(live)} rank(s)", file=sys.stderr)
                    break
            if live and rc == 0:
                time.sleep(poll_s)
    finally:
        for p in procs:                      # end exactly the processes started here
            if p.poll() is None:
                p.terminate()
        deadline = time.time() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
    return rc


# ------------------------------------------------------------------------------------------------ CPU baseline
def _cpu_worker(args):
    """Decode a slice of records `reps` times with the CPU oracle (one process = one core)."""
    import numpy as np
    frames, band, pn, L, reps, start = args
    from echoseal_amd.tables import pack_tables
    from oracle import oracle as O
    ba, tpl, taps, ntaps, _ = pack_tables()
    for _ in range(reps):
        for i in range(frames.shape[0]):
            b = band[i]
            O.decode_frame(frames[i], ba[b], tpl[b], taps[b, :ntaps[b]], np.unpackbits(pn[i])[:1215], L=L, start=start)
    return frames.shape[0] * reps


def cpu_baseline(frames, band, pn, L, *, start, what, budget_s=15.0):
    import multiprocessing as mp
    from oracle import oracle as O
    O.build()
    cores = max(1, min(os.cpu_count() or 1, 16))
    t0 = time.perf_counter()
    _cpu_worker((frames[:8], band[:8], pn[:8], L, 1, start))
    per_frame = (time.perf_counter() - t0) / 8
    n = frames.shape[0] - frames.shape[0] % cores
    reps = max(1, int(round(budget_s * cores / max(per_frame * n, 1e-9))))     # ~budget_s of CPU work per core
    chunks = [(frames[i:n:cores], band[i:n:cores], pn[i:n:cores], L, reps, start) for i in range(cores)]
    ctx = mp.get_context("fork")
    with ctx.Pool(cores) as pool:
        t0 = time.perf_counter()
        done = sum(pool.map(_cpu_worker, chunks))
        dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n} {what} x {reps} passes = {done} decodes, oracle/c (C restatement of the reference: sync + _llr + SCL-{L}), "
                      f"{cores} processes x 1 thread, {dt:.1f} s"}


def _profile_json(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as fh:
            return json.load(fh)
    except Exception:
        return None


# ------------------------------------------------------------------------------------------------ one rank
def run_rank(a) -> None:
    import gc
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if os.environ.get("ES_BENCH_FAIL_RANK") == str(rank):                      # (tests: a rank that dies before the first collective)
        raise SystemExit(f"rank {rank}: failure injected by ES_BENCH_FAIL_RANK")
    ndev = torch.cuda.device_count()
    if a.backend == "nccl" and local >= ndev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but only {ndev} GPU(s) are visible")
    if a.force_collectives and world != 1:
        raise SystemExit("--force-collectives is for N = 1 (N > 1 runs the collectives anyway)")
    use_dist = world > 1 or a.force_collectives               # the collective code path: every N > 1 run, or N = 1 on request
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:
            with socket.socket() as s_:
                s_.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
    if a.backend != "nccl":
        local = local % max(1, ndev)                           # rehearsal: several ranks may share a GPU
        if use_dist:
            dist.init_process_group(a.backend, rank=rank, world_size=world)       # (the rendezvous needs no device)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if use_dist and a.backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    collectives = {"broadcast": 0, "barrier": 0, "all_reduce": 0}

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier(); collectives["barrier"] += 1
        torch.cuda.synchronize()

    def max_over_ranks(dt: float) -> float:
        if use_dist:
            collectives["all_reduce"] += 1
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return dt

    from echoseal_amd.dist import broadcast_schedule, build_schedule, shard_range, split_schedule
    from echoseal_amd.embedder import WatermarkEmbedder, synthetic_payloads
    from echoseal_amd.engine import DecodePipeline, RxEngine, pipeline_streams
    from echoseal_amd import workloads as WL

    legs = {"auto": ["c2", "c2_lanes", "c3_unfused", "c3_pcie", "c3_sustained", "c4", "c5"] if world == 1 else ["c2", "c4"], "none": []}.get(a.legs, a.legs.split(","))
    L = a.list_size
    eng = RxEngine(local, list_size_max=max(16, L))
    tx = WatermarkEmbedder(KEY)
    # Eight hardware queues: every pipeline of this process runs on the SAME eight streams (four made at high priority for the grouped
    # pipeline's front ends, four further ones: list-decoder streams of the grouped pipeline, lanes of the 65 536-record launches)
    front_streams = pipeline_streams(dev, a.front_lanes, priority=-1)
    back_streams = pipeline_streams(dev, max(a.scl_streams, min(a.big_lanes, 8 - a.front_lanes)))
    big_streams = (back_streams + front_streams)[:a.big_lanes]                # the lanes of the 65 536-record launches
    if len(big_streams) < a.big_lanes:
        raise SystemExit(f"--big-lanes {a.big_lanes}: the process keeps {len(back_streams) + len(front_streams)} streams (GPU_MAX_HW_QUEUES = 8)")
    out_legs = {}

    # ============================================================ headline: C3, weak scaling
    Bw = a.windows
    total = Bw * world
    lo, hi = shard_range(total, rank, world)                   # this rank's counters
    # the key/PN schedule of ALL counters comes from rank 0 (derived on its GPU), one broadcast before the timed region
    tb = time.perf_counter()
    sched = torch.empty((total, 153), dtype=torch.uint8, device=dev)
    if rank == 0:
        p, b = eng.schedule(tx.sec._prng.sub_key, KEY, ctr0=0, n=total)
        sched[:, :152] = p; sched[:, 152] = b
        del p, b
    if use_dist:
        dist.broadcast(sched, src=0); collectives["broadcast"] += 1
    barrier()
    bcast_s = max_over_ranks(time.perf_counter() - tb)
    pn3, band3 = split_schedule(sched, lo, hi)
    del sched
    parts = []
    for c0 in range(0, Bw, 16384):                             # input synthesis on the device: this rank's clean frames ...
        parts.append(eng.synthetic_frames(KEY, lo + c0, min(16384, Bw - c0))[0])
    clean = torch.cat(parts); del parts
    win, off = WL.c3_windows_device(clean, seed=34 + rank)     # ... resampled, offset, in noise
    del clean

    pipe3 = DecodePipeline(eng, list_size=L, lanes=a.big_lanes, streams=big_streams)
    for e in pipe3.lane_engs:                                   # kernels by launch size: 65 536 records -> one lane per path
        e.set_option("scl_multi", -1); e.set_option("scl_lane_slab", 1)
    ev3 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]

    def c3_step(k=None):
        sy, _llr, scl, _done = pipe3.submit(win, band3, pn3, start="peak", select=True, xcorr_events=None if k is None else ev3[k])
        return sy, scl

    # Untimed preparation: every lane's context / kernels run once before the clock starts (first launches allocate and upload
    # code), then the W warm-up steps the contract asks for.
    for _ in range(a.big_lanes):
        c3_step()
    torch.cuda.synchronize()
    for _ in range(a.warmup):
        c3_step()
    torch.cuda.synchronize()
    gc.collect(); gc.disable()
    barrier()
    t0 = time.perf_counter()
    for k in range(a.steps):
        sy3, scl3 = c3_step(k)
    host_enqueue_s = time.perf_counter() - t0
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    gc.enable()
    fused_ms = float(np.mean([s.elapsed_time(e) for s, e in ev3])) if a.steps else float("nan")
    payload3, ok3, _which3 = scl3.selected
    if int((ok3 == -2).sum().item()) or int((scl3.ncand < 0).sum().item()):
        raise SystemExit("list decoder reported undecoded records (ncand < 0)")
    pk, npk = sy3.peaks, sy3.npeaks
    found = int(((pk[:, :5] - off[:, None]).abs() <= 2).any(dim=1).sum().item())
    head = {"listed": int((scl3.ncand > 0).sum().item()), "crc_ok": int((ok3 == 1).sum().item()),
            "fallback": int(((npk >> 30) & 1).sum().item()), "exact_rows": int((sy3.flags != 0).sum().item())}
    head_fps_per_gpu = Bw * a.steps / dt

    # one sequential pass (default stream): stage breakdown, and the results every other arrangement must reproduce
    def c3_seq(fused: bool, steps: int):
        st = [[torch.cuda.Event(enable_timing=True) for _ in range(7)] for _ in range(steps)]
        xe = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]

        def one(k):
            s = st[k] if k is not None else None
            if s: s[0].record()
            y, y32 = eng.bpf2(win, band3)
            if s: s[1].record(); xe[k][0].record()
            if fused:
                thr, pk_, npk_, flags = eng.sync_fused(y, y32, band3)
                if s: xe[k][1].record(); s[2].record()
            else:
                c32 = eng.xcorr32(y32, band3)
                if s: xe[k][1].record(); s[2].record()
                thr, pk_, npk_, flags = eng.pick_exact(c32, y, band3)
            if s: s[3].record()
            llr = eng.llr(y, band3, pn3, start=pk_[:, 0].clamp(min=0).contiguous())
            if s: s[4].record()
            scl = eng.scl(llr, list_size=L, skip_if_hard_ok=True)
            if s: s[5].record()
            payload, ok, _w = eng.select(scl)
            if s: s[6].record()
            return pk_, npk_, payload, ok
        one(None); torch.cuda.synchronize()
        barrier()
        t = time.perf_counter()
        for k in range(steps):
            res = one(k)
        barrier()
        dts = max_over_ranks(time.perf_counter() - t)
        names = ("bpf", "sync_fused" if fused else "xcorr32", "(in sync_fused)" if fused else "pick_exact", "llr", "scl", "select")
        stage_ms = {n: float(np.mean([s[i].elapsed_time(s[i + 1]) for s in st])) for i, n in enumerate(names)}
        return dts, float(np.mean([s.elapsed_time(e) for s, e in xe])), stage_ms, res

    eng.set_option("scl_multi", -1); eng.set_option("scl_lane_slab", 1)
    _dts, _x, stage_ms, (pk_s, npk_s, payload_s, ok_s) = c3_seq(True, 1)
    same_as_seq = bool(torch.equal(pk, pk_s) and torch.equal(npk, npk_s) and torch.equal(payload3, payload_s) and torch.equal(ok3, ok_s))

    # ============================================================ leg c3_unfused: the screen through HBM (the `roofline` kernel)
    roof = None
    if "c3_unfused" in legs:
        nu = 3
        dtu, x_ms, stage_u, (pk2, npk2, payload2, ok2) = c3_seq(False, nu)
        same = bool(torch.equal(pk_s, pk2) and torch.equal(npk_s, npk2) and torch.equal(payload_s, payload2) and torch.equal(ok_s, ok2))
        ach = XCORR_BYTES_PER_WINDOW * Bw / (x_ms * 1e-3) / 1e9
        pmc_file = next((f for f in ("r04_xcorr32_pmc_traffic.json", "r02_xcorr32_pmc_traffic.json") if _profile_json(f)), None)
        pmc = (_profile_json(pmc_file) or {}).get("c3_launch", {}) if pmc_file else {}
        roof = {"kernel": "es_xcorr32_kernel<17,2048> (es_xcorr32_batch: the stand-alone correlation kernel north_star grades against HBM)",
                "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": pmc.get("hbm_bytes_per_launch") if Bw == 65536 else None,
                "traffic_source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes committed as profiles/{pmc_file} (same kernel, same launch shape; not re-measured by this run)",
                "launch_ms": x_ms, "algorithmic_bytes_per_launch": XCORR_BYTES_PER_WINDOW * Bw,
                "on_headline_path": False,
                "where": f"HIP events around the launch inside the timed c3_unfused leg ({nu} steps); the headline's sync is the fused kernel "
                         "(roofline_fused), which keeps this row in LDS -- this kernel is the es_xcorr32_batch entry point and the HBM-graded one"}
        out_legs["c3_unfused"] = {"workload": "the headline's windows, sequential, sync = es_xcorr32_batch + es_pick_exact_batch (screen through HBM)",
                                  "value": world * Bw * nu / dtu, "unit": "frames/s", "scaling": "weak", "steps": nu, "ms_per_step": 1e3 * dtu / nu,
                                  "stage_ms": stage_u, "results_identical_to_the_headline": same}
        del pk2, npk2, payload2, ok2
    # ============================================================ leg c3_pcie: the headline with every step's samples coming over PCIe
    if "c3_pcie" in legs:
        # the boundary may hand over HOST buffers: each step's 65 536 windows (0.5 GB) are copied from pinned host memory on a copy
        # stream into one of `big_lanes` + 1 device buffers while earlier steps decode; the lane waits for its copy.  Never `value`.
        nbuf = a.big_lanes + 1
        host = torch.empty(win.shape, dtype=win.dtype, pin_memory=True)
        host.copy_(win)
        copy_st = torch.cuda.Stream(dev)
        bufs = [torch.empty_like(win) for _ in range(nbuf)]
        free_ev = [None] * nbuf                                   # the step that last read the buffer

        def pcie_step(k):
            b = k % nbuf
            with torch.cuda.stream(copy_st):
                if free_ev[b] is not None:
                    copy_st.wait_event(free_ev[b])
                bufs[b].copy_(host, non_blocking=True)
                ready = torch.cuda.Event(); ready.record()
            torch.cuda.current_stream(dev).wait_event(ready)       # (submit makes the lane wait for the caller's stream)
            sy, _llr, scl, done = pipe3.submit(bufs[b], band3, pn3, start="peak", select=True)
            free_ev[b] = done
            return sy, scl
        for k in range(nbuf):
            pcie_step(k)
        torch.cuda.synchronize()
        npc = max(8, a.steps // 2)
        barrier()
        tp = time.perf_counter()
        for k in range(npc):
            syp, sclp = pcie_step(k)
        barrier()
        dtp = max_over_ranks(time.perf_counter() - tp)
        same_p = bool(torch.equal(syp.peaks, pk) and torch.equal(sclp.selected[0], payload3))
        out_legs["c3_pcie"] = {"workload": "the headline with each step's 0.5 GB of samples copied from pinned host memory (copy stream, "
                                           f"{nbuf} device buffers) instead of resident in HBM",
                               "value": world * Bw * npc / dtp, "unit": "frames/s", "scaling": "weak", "steps": npc, "ms_per_step": 1e3 * dtp / npc,
                               "host_to_device_GBps_needed": win.numel() * 4 * npc / dtp / 1e9, "results_identical_to_the_headline": same_p}
        del host, bufs, syp, sclp
    # ============================================================ leg c3_sustained: the headline's steps for seconds on end
    if "c3_sustained" in legs:
        ns = max(1, a.sustained_steps)
        torch.cuda.synchronize()
        gc.collect(); gc.disable()
        barrier()
        ts = time.perf_counter()
        for _ in range(ns):
            sys_, scls_ = c3_step()
        barrier()
        dts = max_over_ranks(time.perf_counter() - ts)
        gc.enable()
        same_s = bool(torch.equal(sys_.peaks, pk) and torch.equal(scls_.selected[0], payload3))
        out_legs["c3_sustained"] = {"workload": f"the headline's step (same windows, same {a.big_lanes} lanes) {ns} times in a row",
                                    "value": world * Bw * ns / dts, "unit": "frames/s", "scaling": "weak", "steps": ns, "seconds": dts,
                                    "ms_per_step": 1e3 * dts / ns, "ratio_to_the_headline": (world * Bw * ns / dts) / (total * a.steps / dt),
                                    "results_identical_to_the_headline": same_s}
        del sys_, scls_
    del pk_s, npk_s, payload_s, ok_s, sy3, scl3, payload3, ok3, pk, npk
    win_h = band_h = pn_h = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        ns = min(Bw, 1024)
        win_h, band_h, pn_h = win[:ns].cpu().numpy(), band3[:ns].cpu().numpy(), pn3[:ns].cpu().numpy()
    del win, off, pn3, band3, pipe3
    torch.cuda.empty_cache()

    # ============================================================ legs c2 / c2_lanes / c2_single: 1 024-frame batches, streamed
    if "c2" in legs or "c2_lanes" in legs:
        B = a.frames
        tot2 = B * world
        lo2, hi2 = shard_range(tot2, rank, world)
        ctrs = list(range(lo2, hi2))
        frames_h = tx.make_frames(ctrs, synthetic_payloads(tx.sec, ctrs))       # each rank synthesises its own frames (host embedder)
        sched2 = build_schedule(KEY, range(tot2)) if rank == 0 else None
        sched2_d = broadcast_schedule(sched2, tot2, dev)
        collectives["broadcast"] += int(use_dist)
        pn_d, band_d = split_schedule(sched2_d, lo2, hi2)
        frames_d = torch.from_numpy(frames_h).to(dev)
    if "c2" in legs:
        pipe = DecodePipeline(eng, list_size=L, scl_streams=a.scl_streams, lanes=a.front_lanes, group=a.group,
                              streams=(front_streams, back_streams[:a.scl_streams]))

        def step():
            sync_res, _llr, res, _done = pipe.submit(frames_d, band_d, pn_d, inputs_ready=True)   # (resident since long before the clock starts)
            return res, sync_res.peaks, sync_res.npeaks
        for _ in range((len(pipe.backs) + 1) * a.group):        # every stream / context / kernel instantiation once, group buffers exist
            step()
        pipe.synchronize(); torch.cuda.synchronize()
        lat = []                                                # one batch alone: a group of one is decoded by the kernel the library picks for 1 024 frames
        step(); pipe.synchronize(); torch.cuda.synchronize()
        for _ in range(3):
            t1 = time.perf_counter()
            step()
            pipe.synchronize()
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t1)
        gc.collect(); gc.disable()
        barrier()
        t2 = time.perf_counter()
        for _ in range(a.c2_steps):
            res, peaks, npeaks = step()
        pipe.synchronize()                                      # (decodes the last, possibly incomplete, group inside the timed region)
        barrier()
        dt2 = max_over_ranks(time.perf_counter() - t2)
        gc.enable()
        ok_sync = bool(torch.all((npeaks >= 1) & (npeaks < 32)).item() and torch.all(peaks[:, 0] == 0).item())
        res = res.result()
        out_legs["c2"] = {"workload": f"C2: batches of {B} clean 1215-sample float32 frames per GPU (ctr = i, payload seed 20260101), sync + _llr(variant 0, start 0) + SCL-{L}; "
                                      f"grouped pipeline: the front ends of {a.group} batches on {pipe.lanes} streams fill one LLR buffer, one list-decoder launch per group on one of "
                                      f"{len(pipe.backs)} streams, the last (incomplete) group decoded inside the timed region",
                          "value": tot2 * a.c2_steps / dt2, "unit": "frames/s", "scaling": "weak", "steps": a.c2_steps, "ms_per_step": 1e3 * dt2 / a.c2_steps,
                          "timed_region_ms": 1e3 * dt2, "sync_offsets_ok": ok_sync, "frames_through_list_decoder": int((res.ncand > 0).sum().item())}
        out_legs["c2_single"] = {"workload": "one C2 batch alone, submit -> complete (host clock, minimum of 3)", "value": 1e3 * min(lat), "unit": "ms",
                                 "higher_is_better": False}
        del res, peaks, npeaks, pipe
    if "c2_lanes" in legs:
        st7 = (front_streams + back_streams)[:7]
        st7 += pipeline_streams(dev, 7 - len(st7))
        pipe7 = DecodePipeline(eng, list_size=L, lanes=7, streams=st7)
        for e in pipe7.scl_engs:
            e.set_option("scl_multi", 1); e.set_option("scl_lanes", 4)
        n7 = min(a.c2_steps, 200)
        for _ in range(14):                                     # every lane's context at least twice before the clock starts
            pipe7.submit(frames_d, band_d, pn_d)
        pipe7.synchronize(); barrier()
        t7 = time.perf_counter()
        for _ in range(n7):
            _sy7, _l7, res7, _d7 = pipe7.submit(frames_d, band_d, pn_d)
        pipe7.synchronize(); barrier()
        dt7 = max_over_ranks(time.perf_counter() - t7)
        out_legs["c2_lanes"] = {"workload": "C2 batches through seven whole-chain lanes (one list-decoder launch -- several frames per wave, four lanes per path -- per "
                                            "1 024-frame batch): rows within ~3 ms of submission",
                                "value": tot2 * n7 / dt7, "unit": "frames/s", "scaling": "weak", "steps": n7, "ms_per_step": 1e3 * dt7 / n7,
                                "frames_through_list_decoder": int((res7.ncand > 0).sum().item())}
        eng.set_option("scl_multi", -1); eng.set_option("scl_lanes", 0)
        del pipe7, res7
    if "c2" in legs or "c2_lanes" in legs:
        del frames_d, pn_d, band_d
        torch.cuda.empty_cache()

    # ============================================================ leg c4: 2^20 frames, strong scaling
    if "c4" in legs:
        T4 = a.c4_frames
        lo4, hi4 = shard_range(T4, rank, world)
        n4 = hi4 - lo4
        barrier()
        tb4 = time.perf_counter()
        sched4 = torch.empty((T4, 153), dtype=torch.uint8, device=dev)
        if rank == 0:
            p, b = eng.schedule(tx.sec._prng.sub_key, KEY, ctr0=0, n=T4)
            sched4[:, :152] = p; sched4[:, 152] = b
            del p, b
        if use_dist:
            dist.broadcast(sched4, src=0); collectives["broadcast"] += 1
        barrier()
        bcast4_s = max_over_ranks(time.perf_counter() - tb4)
        pn4, band4 = split_schedule(sched4, lo4, hi4)
        del sched4
        frames4 = torch.empty((n4, 1215), dtype=torch.float32, device=dev)
        for c0 in range(0, n4, 65536):                     # this rank's shard, made on its GPU (input synthesis)
            m = min(65536, n4 - c0)
            frames4[c0:c0 + m] = eng.synthetic_frames(KEY, lo4 + c0, m)[0]
        payload4 = torch.empty((n4, 55), dtype=torch.uint8, device=dev)
        ok4 = torch.empty(n4, dtype=torch.int8, device=dev)
        peak4 = torch.empty(n4, dtype=torch.int32, device=dev)
        chunk = max(1, min(a.c4_chunk, n4))
        pipe4 = DecodePipeline(eng, list_size=L, lanes=a.big_lanes, streams=big_streams)
        for e in pipe4.lane_engs:
            e.set_option("scl_multi", -1); e.set_option("scl_lane_slab", 1)

        def c4_pass(limit=None):
            for c0 in range(0, n4 if limit is None else min(n4, limit), chunk):
                c1 = min(n4, c0 + chunk)
                sy, _llr, scl, _done = pipe4.submit(frames4[c0:c1], band4[c0:c1], pn4[c0:c1], select=True)
                with torch.cuda.stream(pipe4.lane_streams[(pipe4._k - 1) % pipe4.lanes]):     # <= 64 B per frame kept, on the lane's stream
                    payload4[c0:c1] = scl.selected[0]; ok4[c0:c1] = scl.selected[1]; peak4[c0:c1] = sy.peaks[:, 0]
                del sy, _llr, scl
            pipe4.synchronize()

        c4_pass(limit=chunk * a.big_lanes)                 # warm-up: one chunk per lane
        barrier()
        t4 = time.perf_counter()
        c4_pass()
        barrier()
        dt4 = max_over_ranks(time.perf_counter() - t4)
        idx = torch.arange(lo4, hi4, dtype=torch.int64, device=dev)
        w = torch.arange(1, 56, dtype=torch.int64, device=dev)
        cks = (((payload4.to(torch.int64) * w).sum(1) + 1000 * ok4.to(torch.int64) + 7 * peak4.to(torch.int64)) * (idx % 65521 + 1)).sum().reshape(1)
        good = (peak4 == 0).sum().reshape(1)
        bad = (ok4 == -2).sum().reshape(1)
        if use_dist:
            dist.all_reduce(cks); dist.all_reduce(good); dist.all_reduce(bad); collectives["all_reduce"] += 3
        if int(bad.item()):
            raise SystemExit("c4: list decoder reported undecoded records")
        out_legs["c4"] = {"workload": f"C4: {T4} clean 1215-sample frames in total (ctr 0..{T4 - 1}, generated on the device), sharded contiguously over "
                                      f"{world} rank(s); schedule (153 B/ctr) derived on rank 0 and broadcast; sync + _llr(start 0) + SCL-{L} + selection "
                                      f"in launches of {chunk} frames on {a.big_lanes} pipeline lanes",
                          "value": T4 / dt4, "unit": "frames/s", "scaling": "strong", "seconds": dt4, "frames_total": T4,
                          "frames_per_rank": n4, "schedule_broadcast_s": bcast4_s, "schedule_bytes": T4 * 153,
                          "value_incl_broadcast": T4 / (dt4 + bcast4_s), "checksum": int(cks.item()),
                          "frames_with_sync_offset_0": int(good.item())}
        del frames4, payload4, ok4, peak4, pn4, band4, pipe4
        torch.cuda.empty_cache()

    # ============================================================ leg c5 (surrogate; one GPU): list-size sweep
    if "c5" in legs and world == 1:
        out_legs["c5"] = c5_leg(eng, a, torch, np, WL)

    if rank == 0:
        fused_alone_ms = stage_ms["sync_fused"]                      # the launch alone (one sequential step, HIP events around it)
        achf = FUSED_BYTES_PER_WINDOW * Bw / (fused_alone_ms * 1e-3) / 1e9
        out = {
            "metric": "watermark frames/sec decoded (sync+LLR+SCL-8) @ 48 kHz",
            "value": total * a.steps / dt, "unit": "frames/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"C3: {Bw} windows of 2048 float32 samples per GPU and step, one frame each (key 0xAA*32, ctr = i, resampled by U[0.95,1.05] with linear "
                                   f"interpolation, uniform offset, AWGN at -15 dB SNR; generated on the device); band-pass -> float32 NCC screen + exact median/MAD "
                                   f"threshold + NMS / top-5 (one fused kernel) -> _llr(variant 0) at the detected peak -> SCL-{L} (validator None) -> selection",
                       "windows_per_gpu_per_step": Bw, "window_len": 2048, "list_size": L, "frame_len": 1215, "fs": 48000,
                       "world_size": world, "backend": ("nccl (RCCL)" if a.backend == "nccl" else a.backend) if use_dist else "none (single rank)",
                       "collectives_issued": dict(collectives, process_group=bool(use_dist)),
                       "sharding": f"{world} x {Bw} windows per step (rank r: counters [r*{Bw}, (r+1)*{Bw})); schedule of all {total} counters derived on rank 0, "
                                   f"one broadcast before the timed region ({total * 153} B, {bcast_s:.4f} s incl. derivation)",
                       "pipelining": f"steps rotate over {a.big_lanes} pipeline lanes (HIP streams, one context each): the front ends of some steps run beside the "
                                     f"list decoders of others; one list-decoder launch of {Bw} frames per step (one lane per path, 64/L frames per wave)",
                       "untimed_preparation": "one step per lane (first launches allocate scratch and upload code), then the --warmup steps; the garbage collector is off inside the timed region",
                       "host_enqueue_ms_of_the_timed_steps": 1e3 * host_enqueue_s, "timed_region_ms": 1e3 * dt,
                       "stage_ms_one_step_alone": stage_ms, "results_identical_to_a_sequential_pass": same_as_seq,
                       "frames_through_list_decoder": head["listed"], "frames_with_crc_ok": head["crc_ok"],
                       "windows_with_a_top5_peak_within_2_samples_of_the_true_offset": found, "fallback_records": head["fallback"],
                       "records_settled_by_the_exact_float64_row": head["exact_rows"]},
            "legs": out_legs,
        }
        roof_fused = {"kernel": "es_xcorr32_kernel<17,2048,FUSED> (es_sync_fused_batch: screen row kept in LDS, threshold and peaks settled in the same kernel)",
                      "bound": "hbm", "achieved": achf, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achf / HBM_PEAK_GBS,
                      "traffic": ((_profile_json("r04_xcorr32_pmc_traffic.json") or _profile_json("r02_xcorr32_pmc_traffic.json") or {}).get("c3_launch_fused", {}).get("hbm_bytes_per_launch") if Bw == 65536 else None),
                      "launch_ms": fused_alone_ms, "launch_ms_inside_the_timed_steps": fused_ms,
                      "algorithmic_bytes_per_launch": FUSED_BYTES_PER_WINDOW * Bw, "on_headline_path": True,
                      "note": "8 192 B of samples in + <= 150 B out per window (SURVEY 8d fused figure); bound by LDS passes and float64 re-evaluations, not by HBM -- "
                              "it exists to take 2 x 7 944 B per window of screen traffic and two launches away",
                      "where": "`launch_ms` (and `achieved` / `frac`): HIP events around the launch in one sequential step, nothing beside it; "
                               f"`launch_ms_inside_the_timed_steps`: the same events inside the timed headline steps ({a.steps} steps), where the launch is queued beside "
                               "other lanes' list decoders and the events span their time too -- reported, not used"}
        if roof is not None:
            out["roofline"] = roof
            out["roofline_fused"] = roof_fused
        else:
            out["roofline"] = roof_fused                  # no c3_unfused leg in this run: the sync kernel of the headline steps
        scl_roof = scl_roofline(head_fps_per_gpu, out_legs.get("c4", {}).get("value", 0.0) / world if "c4" in out_legs else None, L, head["listed"] / Bw)
        if scl_roof:
            out["roofline_scl"] = scl_roof
        if win_h is not None:
            out["cpu_baseline"] = cpu_baseline(win_h, band_h, pn_h, L, start="peak", what=f"of the {Bw} headline windows (2048 samples each)")
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


def scl_roofline(head_fps, c4_fps, L, listed_share=1.0):
    """roofline_scl at the headline's own per-GPU rate, from the committed counter passes of es_scl_wide_kernel<64,8>.  Only the frames that
    go through the list loop cost what the counters say (the others are settled by the hard-decision shortcut, in a kernel of its own)."""
    all_fps, head_fps = head_fps, head_fps * listed_share
    pmc_name = next((f for f in ("r04_scl_pmc.json", "r03_scl_pmc.json") if _profile_json(f)), None)
    pmc = _profile_json(pmc_name) if pmc_name else None
    if not pmc or L != 8:
        return None
    pf = pmc["per_frame"]
    vi, f64 = pf["valu_instructions"], pf["fp64_instructions"]
    other = vi - f64 - pf.get("trans_f64_instructions", 0)
    rate = vi * head_fps / 1e9
    cyc = pmc.get("issue_cycles", {"fp64": 4.0, "trans_f64": 16.0, "other": 4.0})
    busy_cycles_per_frame = f64 * cyc["fp64"] + pf.get("trans_f64_instructions", 0) * cyc["trans_f64"] + other * cyc["other"]
    out = {"kernel": "es_scl_wide_kernel<64,8> (one lane per path; the dominant kernel by time: ~88 % of a step's GPU work)", "bound": "fp64 vector issue",
           "achieved": rate, "peak": FP64_ISSUE_PEAK_GWIPS, "unit": "G wave-instructions/s", "frac": rate / FP64_ISSUE_PEAK_GWIPS,
           "frames_per_s_used": head_fps, "frames_per_s_from": f"the timed headline steps (per GPU): {all_fps:.0f} frames/s x {listed_share:.4f}, the share of them that go through "
                                                               "the list loop (rank 0's last step; the rest pass the hard-decision shortcut)",
           "frac_at_the_c4_rate": (vi * c4_fps / 1e9 / FP64_ISSUE_PEAK_GWIPS) if c4_fps else None,
           "valu_wave_instructions_per_frame": vi, "fp64_wave_instructions_per_frame": f64,
           "fp64_pipe_frac": (f64 * 4.0 + pf.get("trans_f64_instructions", 0) * 16.0) * head_fps / (N_SIMD * CLOCK_GHZ * 1e9),
           "issue_slot_frac_mixed_ceiling": busy_cycles_per_frame * head_fps / (N_SIMD * CLOCK_GHZ * 1e9),
           "frac_guide_ceiling": (f64 * 4.0 + pf.get("trans_f64_instructions", 0) * 16.0 + other * 2.0) * head_fps / (N_SIMD * CLOCK_GHZ * 1e9),
           "issue_cycles_charged": cyc,
           "traffic": pf.get("hbm_side_bytes"), "algorithmic_bytes_per_frame": SCL_BYTES_PER_FRAME,
           "traffic_note": "bytes per frame through the L2's memory side (FETCH_SIZE x 2 for the 128-byte read requests + WRITE_SIZE; Infinity-Cache hits are counted): "
                           "the tree levels kept in the scratch slab, not the 4.6 KB a frame brings and leaves",
           "how": f"per-frame counts from the rocprofv3 --pmc passes committed as profiles/{pmc_name} (same kernel, B = 65 536) x the headline's frames/s per GPU; "
                  "`frac` charges every vector instruction 4 cycles (peak 78.6 TFLOP/s FP64 vector = one wave64 instruction per 4 cycles on each of 1 024 SIMDs at 2.4 GHz); "
                  "`issue_slot_frac_mixed_ceiling` charges each class what tools/ub/ub_issue.hip measures at three waves per SIMD; `frac_guide_ceiling` charges float64 4, "
                  "v_rcp_f64 16 and every other vector instruction the 2 cycles of MI355X_MICROARCH.md (the lowest of the three readings); `fp64_pipe_frac` counts only float64 arithmetic"}
    return out


def c5_leg(eng, a, torch, np, WL):
    """Config-5 surrogate: frames through workloads.lossy_channel (NOT MP3), list size swept."""
    dev = eng.device
    U, Bn = 1024, a.c5_frames
    frames, band, pn, payloads = WL.c2_frames(range(U))
    lossy = WL.lossy_channel(frames)
    reps = -(-Bn // U)
    f = torch.from_numpy(lossy).to(dev).repeat(reps, 1)[:Bn].contiguous()
    b = torch.from_numpy(band).to(dev).repeat(reps)[:Bn].contiguous()
    p = torch.from_numpy(pn).to(dev).repeat(reps, 1)[:Bn].contiguous()
    want = torch.from_numpy(np.unpackbits(np.frombuffer(b"".join(payloads), np.uint8).reshape(U, 55), axis=1)).to(dev).repeat(reps, 1)[:Bn]
    sweep = {}
    for Ls in (1, 4, 8, 16):
        eng.decode_batch(f, b, p, list_size=Ls); torch.cuda.synchronize()
        t0 = time.perf_counter()
        sy, llr, scl = eng.decode_batch(f, b, p, list_size=Ls)
        payload, ok, which = eng.select(scl)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        got = payload[:, :, None].bitwise_right_shift(torch.arange(7, -1, -1, device=dev, dtype=torch.uint8)).bitwise_and(1).reshape(Bn, 440)
        sweep[f"L{Ls}"] = {"frames_per_s": Bn / dt, "payload_ber": float((got != want).float().mean().item()),
                           "crc_ok_frames": int((ok == 1).sum().item())}
    return {"workload": f"C5 SURROGATE (MP3 128 kbps itself: skipped, no codec in the image): {Bn} frames = {U} clean C2 frames through "
                        "workloads.lossy_channel (16 kHz low-pass + level-shaped noise; NOT MP3), sync + _llr + SCL-L + selection",
            "mp3": "skipped: no codec", "sweep": sweep,
            "note": "the reference's chain does not recover payloads even on clean frames (SURVEY section 0.2: BER ~0.5 by construction); "
                    "the BER is reported because the config asks for it"}


def main() -> None:
    a = parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(a))
    run_rank(a)


if __name__ == "__main__":
    main()
