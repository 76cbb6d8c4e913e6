"""bench.py must be launchable for N > 1 exactly the way the driver launches it for N = 1 — `python bench.py --gpus N` —
and under torch.distributed.run (the N > 1 line used to exit with "launch with torch.distributed.run").
MTR_BENCH_DRY_RUN=1 runs the launcher logic only: self-launch, rendezvous on 127.0.0.1, shard arithmetic; no GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, **env):
    e = dict(os.environ, MTR_BENCH_DRY_RUN="1", **env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                         # exactly one JSON line, from rank 0
    return json.loads(lines[0])


def test_plain_python_launch_spawns_the_ranks():
    d = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--streams", "101"])
    assert d["dry_run"] and d["n_gpus"] == 2
    rows = sorted(d["ranks"])
    assert [r[0] for r in rows] == [0, 1] and [r[1] for r in rows] == [0, 1]          # rank, local rank = GPU ordinal
    # weak scaling: every rank owns `--streams` streams, contiguous and disjoint
    assert [r[3] for r in rows] == [101, 101] and [r[2] for r in rows] == [0, 101]
    # the fields with which the real line proves its collective (config.rccl_nranks = ncclCommCount, per rank ncclCommCuDevice): gathered
    # over the same control plane, one row per rank on its own device; no communicator exists in a dry run
    assert d["rccl_nranks"] is None
    assert sorted((x["rank"], x["hip_device"], x["rccl_device"]) for x in d["rank_devices"]) == [(0, 0, None), (1, 1, None)]


def test_torchrun_launch_and_single_rank():
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
              "--master-port", "29613", "bench.py", "--gpus", "3", "--streams", "7"])
    assert d["n_gpus"] == 3 and sorted(r[2] for r in d["ranks"]) == [0, 7, 14]
    d = _run([sys.executable, "bench.py", "--gpus", "1", "--streams", "5"])
    assert d["n_gpus"] == 1 and d["ranks"] == [[0, 0, 0, 5]]


def test_rank_count_mismatch_is_an_error():
    e = dict(os.environ, MTR_BENCH_DRY_RUN="1", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "4"], cwd=ROOT, env=e, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "--gpus 4" in r.stderr


def test_a_rank_that_never_arrives_yields_one_line_with_an_error():
    """the control plane has a deadline (MTR_BENCH_CTRL_TIMEOUT_S; gloo's default is 30 minutes) — a rank
    that hangs before the rendezvous costs the job that long, and rank 0 still prints exactly one JSON line, with `error`."""
    e = dict(os.environ, MTR_BENCH_DRY_RUN="1", MTR_BENCH_CTRL_TIMEOUT_S="5", MTR_BENCH_FAULT="hang_before_rendezvous:1:40")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    import time
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--streams", "9"], cwd=ROOT, env=e, capture_output=True, text=True, timeout=240)
    took = time.monotonic() - t0
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode != 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    d = json.loads(lines[0])
    assert d["value"] is None and d["n_gpus"] == 2 and d["error"]
    assert took < 120, took
