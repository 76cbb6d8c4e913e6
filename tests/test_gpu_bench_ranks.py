"""The N > 1 DATA path of bench.py on the one-GPU box: two ranks share GPU 0
(MTR_BENCH_SHARED_GPU=1), rendezvous over gloo on 127.0.0.1, try the engine's own RCCL communicator (MTR_BENCH_TRY_RCCL=1:
RCCL refuses two ranks on one device, on BOTH ranks), agree on the fallback, meter their shards of the job's streams and
reduce the aggregates every step.  What must hold: one JSON line, the agreed fallback named in it, and a programme record —
summed histograms -> mtr_hist_loudness, max of peaks — that equals, bit for bit, ONE engine over all the job's streams
(seeds 777 ... 777 + 127) fed the same calls.  With a second GPU the same command without the two variables is the real
thing (tests/test_gpu_reduce.py::test_two_ranks_against_one_engine covers mtr_engine_reduce over RCCL there)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S, SECONDS, STEPS, WARMUP = 64, 3, 2, 1


def _one_engine(n_streams, tune_segments):
    """The job as ONE engine: every stream of both shards, the same (warm-up + timed) calls over the same buffer."""
    import torch
    import meters.lv2_amd as M
    from meters.lv2_amd import dist as mdist
    fs, T = 48000.0, SECONDS * 48000
    buf = torch.empty((n_streams, T, 2), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    M.synth_fill_device(buf.data_ptr(), n_streams, T, T, 777, fs, 1, st)
    hist = torch.zeros(2 * 751, dtype=torch.int32, device="cuda")
    mx = torch.zeros(4, dtype=torch.float32, device="cuda")
    with M.Engine(n_streams, fs, M.METER_EBU | M.METER_TRUEPEAK, tune_segments=tune_segments) as e:
        e.integr_start()
        for _ in range(WARMUP + STEPS):
            e.process_device(buf.data_ptr(), T, T, st)
        e.aggregate_device(hist.data_ptr(), mx.data_ptr(), st)
        torch.cuda.synchronize()
        per_stream = [r.hist_M_count for r in e.results()]
        seg_calls = e.seg_stats()[0]
    return mdist.programme_summary(hist, mx), per_stream, seg_calls


@pytest.mark.parametrize("segments", [0, 2], ids=["default-routing", "lane=segment-kernel"])
def test_two_ranks_share_the_gpu_and_reduce_like_one_engine(segments):
    env = dict(os.environ, MTR_BENCH_SHARED_GPU="1", MTR_BENCH_TRY_RCCL="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--streams", str(S), "--seconds", str(SECONDS),
           "--steps", str(STEPS), "--warmup", str(WARMUP), "--no-extra", "--no-cpu-baseline"]
    if segments:
        cmd += ["--segments", str(segments)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == STEPS and line["scaling"] == "weak"
    # the agreed fallback, and why: both ranks must have seen mtr_comm_init fail and must say how they reduced instead
    coll = line["config"]["collective"]
    assert "gloo" in coll and "mtr_comm_init failed on a rank" in coll, coll
    assert len(line["per_rank_ms"]) == 2 and all(v > 0 for v in line["per_rank_ms"])
    assert len(line["per_rank_kernel_ms"]) == 2 and all(v > 0 for v in line["per_rank_kernel_ms"])
    assert 0 < line["speedup_vs_ideal"] <= 1.0 + 1e-9
    assert line["ms_per_step_min_max"][0] <= line["ms_per_step_median"] <= line["ms_per_step_min_max"][1]
    want, per_stream, seg_calls = _one_engine(2 * S, segments)
    assert seg_calls == ((WARMUP + STEPS) if segments else 0)          # (which kernel the comparison went through)
    got = line["programme"]
    # every stream adds a point to its momentary histogram every second 50 ms fragment (ebu_r128_proc.cc:229-233; a point
    # below -70 LUFS is not counted, :71): the job's count is the sum over both shards' streams
    full = (WARMUP + STEPS) * SECONDS * 10
    assert len(per_stream) == 2 * S and max(per_stream) == full and min(per_stream) > full // 2
    assert got["hist_M_count"] == sum(per_stream)
    for k, v in want.items():
        g = got[k]
        assert (list(g) == list(v)) if isinstance(v, tuple) else (g == v), (k, g, v)


def test_a_rank_that_sleeps_through_the_communicator_cannot_hang_the_job():
    """the job's first contact with RCCL is bounded.  Rank 1 holds the communicator's id and sleeps 25 s
    before it joins mtr_comm_init; rank 0 is INSIDE RCCL meanwhile (the bootstrap waits for every rank), and its
    mtr_comm_init_timeout (ncclCommInitRankConfig non-blocking, ncclCommGetAsyncError polled) gives up after
    MTR_BENCH_COMM_TIMEOUT_S = 6 s and aborts the communicator; rank 1, arriving late, fails or times out as well; the ranks
    vote over gloo, fall back together, and rank 0 prints exactly ONE line that names the fallback and the deadline that
    passed — in tens of seconds, not after the 30 minutes a blocking ncclCommInitRank plus gloo's default timeout would take."""
    import time
    env = dict(os.environ, MTR_BENCH_SHARED_GPU="1", MTR_BENCH_TRY_RCCL="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               MTR_BENCH_COMM_TIMEOUT_S="6", MTR_BENCH_FAULT="sleep_in_init:1:25", MTR_BENCH_CTRL_TIMEOUT_S="120")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--streams", "16", "--seconds", "1",
           "--steps", "1", "--warmup", "1", "--no-extra", "--no-cpu-baseline"]
    t0 = time.monotonic()
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    took = time.monotonic() - t0
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    coll = line["config"]["collective"]
    assert line["n_gpus"] == 2 and "gloo" in coll and "mtr_comm_init failed on a rank" in coll, coll
    assert "no answer from RCCL within 6000 ms" in coll, coll          # rank 0 (whose reason the line carries) ran into its deadline
    assert line["config"]["comm_init_ms"] is None and line["config"]["rccl_version"] >= 21800
    assert 20e3 < line["config"]["comm_negotiation_ms"] < 120e3, line["config"]
    assert line["value"] > 0 and took < 300, took
