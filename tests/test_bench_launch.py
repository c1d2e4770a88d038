"""bench.py's own launcher: `python bench.py --gpus N` without WORLD_SIZE starts N rank processes itself and fails
loudly when the devices are not there; on a GPU box a two-rank rehearsal (gloo, both ranks on the one GPU) must print
n_gpus = 2 and the strong-scaling leg's checksum must not depend on the number of ranks."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["MASTER_ADDR"] = "127.0.0.1"
    return env


def test_refuses_more_ranks_than_gpus():
    import torch
    n = torch.cuda.device_count()
    p = subprocess.run([sys.executable, BENCH, "--gpus", str(n + 2), "--steps", "1", "--warmup", "0"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and "GPU(s) are visible" in p.stderr and not p.stdout.strip()


def test_rank_refuses_world_size_mismatch():
    env = dict(_env(), WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=3" in (p.stderr + p.stdout)


def test_failing_rank_tears_the_job_down_within_seconds():
    """A rank that dies while its peers wait in the rendezvous / a collective must not wedge the job: the launcher polls ALL children,
    stops the others on the first non-zero exit and returns non-zero (CPU, gloo: rank 0 blocks in init_process_group waiting for
    rank 1, which exits through the injected failure)."""
    import time
    env = dict(_env(), ES_BENCH_FAIL_RANK="1")
    t0 = time.time()
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0", "--legs", "none"], env=env,
                       capture_output=True, text=True, timeout=300)
    dt = time.time() - t0
    assert p.returncode != 0 and "rank 1 exited" in p.stderr and not p.stdout.strip(), (p.returncode, p.stderr[-800:])
    assert dt < 90, dt                           # (two interpreter starts with `import torch`; the default rendezvous timeout is 30 min)


@pytest.mark.gpu
def test_two_rank_rehearsal_and_checksum_independent_of_world_size():
    """The default N > 1 line (headline C3 + legs c2 and c4) at two ranks (gloo rehearsal, both ranks on the one GPU), sized down."""
    common = ["--steps", "2", "--warmup", "1", "--windows", "8192", "--legs", "c2,c4", "--c2-steps", "40", "--c4-frames", "16384", "--c4-chunk", "4096",
              "--no-cpu-baseline"]
    one = subprocess.run([sys.executable, BENCH, "--gpus", "1"] + common, env=_env(), capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo"] + common, env=_env(), capture_output=True,
                         text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-2000:]
    j1 = json.loads(one.stdout.strip().splitlines()[-1]); j2 = json.loads(two.stdout.strip().splitlines()[-1])
    assert j1["n_gpus"] == 1 and j2["n_gpus"] == 2 and j2["config"]["world_size"] == 2
    assert j1["config"]["workload"].startswith("C3:") and j2["scaling"] == "weak"
    assert j1["config"]["results_identical_to_a_sequential_pass"] and j2["config"]["results_identical_to_a_sequential_pass"]
    assert j2["legs"]["c2"]["scaling"] == "weak" and j2["legs"]["c2"]["sync_offsets_ok"]
    assert j2["legs"]["c4"]["frames_per_rank"] == 8192 and j2["legs"]["c4"]["scaling"] == "strong"
    assert j1["legs"]["c4"]["checksum"] == j2["legs"]["c4"]["checksum"]
    assert j1["legs"]["c4"]["frames_with_sync_offset_0"] == j2["legs"]["c4"]["frames_with_sync_offset_0"] == 16384


def _forced(backend, extra=()):
    return subprocess.run([sys.executable, BENCH, "--gpus", "1", "--backend", backend, "--force-collectives", "--steps", "2", "--warmup", "1",
                           "--windows", "4096", "--legs", "c2,c4", "--c2-steps", "20", "--c4-frames", "8192", "--c4-chunk", "4096", "--no-cpu-baseline",
                           *extra], env=_env(), capture_output=True, text=True, timeout=900)


@pytest.mark.gpu
def test_rccl_runs_once_at_world_size_one():
    """RCCL (backend "nccl") must have run here before the driver's 8-GPU box runs it: a fresh process (nothing in this one has to touch
    the GPU for it) goes through bench.py's whole collective code path at world size 1 -- init_process_group("nccl", device_id=...), the
    schedule broadcasts of the headline and of legs c2 / c4, the barriers, the float64 all_reduce(MAX) of the timing and the int64
    all_reduces of the c4 checksum -- and must agree with the run that issues no collective at all."""
    forced = _forced("nccl")
    assert forced.returncode == 0, forced.stderr[-3000:]
    j = json.loads(forced.stdout.strip().splitlines()[-1])
    c = j["config"]["collectives_issued"]
    assert j["n_gpus"] == 1 and j["config"]["backend"] == "nccl (RCCL)" and c["process_group"]
    assert c["broadcast"] >= 3 and c["barrier"] >= 10 and c["all_reduce"] >= 6, c
    assert j["config"]["results_identical_to_a_sequential_pass"] and j["legs"]["c2"]["sync_offsets_ok"]
    plain = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "2", "--warmup", "1", "--windows", "4096", "--legs", "c4",
                            "--c4-frames", "8192", "--c4-chunk", "4096", "--no-cpu-baseline"], env=_env(), capture_output=True, text=True, timeout=900)
    assert plain.returncode == 0, plain.stderr[-2000:]
    j0 = json.loads(plain.stdout.strip().splitlines()[-1])
    assert not j0["config"]["collectives_issued"]["process_group"] and j0["config"]["backend"] == "none (single rank)"
    assert j0["legs"]["c4"]["checksum"] == j["legs"]["c4"]["checksum"]
    assert j0["legs"]["c4"]["frames_with_sync_offset_0"] == j["legs"]["c4"]["frames_with_sync_offset_0"] == 8192
