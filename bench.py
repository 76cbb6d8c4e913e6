#!/usr/bin/env python3
"""bench.py — throughput of the meters.lv2 hot path on MI355X.

A "step" is one pass of the fused EBU R128 (K-weighting + gated loudness) + 4x true-peak path
over one batch of synthetic 48 kHz stereo audio that is already resident in HBM: per GPU,
8192 streams x 10 s = 3.93 G stereo frames = 31.5 GB (the per-GPU shard of BASELINE.json
configs[4]; the metric is quoted on batched EBU R128 + true-peak).  Streams are independent, so
ranks shard them with no data-path collective (weak scaling: per-GPU work is fixed); the only
RCCL traffic is the final all-reduce of the two 751-bin loudness histograms (sum) and the
peak / max-loudness values (max), done once per step by mtr_engine_reduce() — RCCL inside the C ABI
(torch.distributed only ships the 128-byte communicator id and provides the barrier).

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: re-executes itself under
                                                            torch.distributed.run on 127.0.0.1, one rank per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

The step's TAIL — k_gate, k_aggregate and the all-reduce of step i — runs on the engine's own side stream beside k_seg of step i + 1
(mtr_engine_set_deferred_tail, auto for this workload; `--tail 1` keeps everything on one stream; bit for bit the same results either way:
tests/test_gpu_tail.py); the timed region ends behind a device-wide wait, so the last step's tail is inside it.  `config.tail` says which ran.

Prints ONE JSON line on rank 0.  `value` counts channel-samples/s (2 per stereo frame), whole job.
`roofline` prices the dominant kernel (k_seg, mtr_seg.hip: K-weighting as the reference's recurrence with lane = time
segment + the 4x interpolator on the matrix pipe at f32 grade; the layout every call of this shape takes by default) at
8 algorithmic bytes per stereo frame (one read of the input) against the 8 TB/s HBM peak, with the
kernel's duration measured by HIP events on the launching stream inside the timed steps.  `extra.configs` carries every BASELINE config at its stated size, `extra.lv2_run_latency`
the per-block cost of the LV2 plugins.  `cpu_baseline` times the reference's
own DSP objects (oracle/_ref, kind "reference") — or the repo's restatement (kind "port") where
that build is absent — on a bounded sample of the same buffers on the host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_FRAME = 8            # one stereo f32 frame, read once


def host_cpu():
    """(model string, physical cores this process may run on, logical CPUs) of the host: /proc/cpuinfo + the affinity mask."""
    model, cores = "unknown", set()
    try:
        allowed = os.sched_getaffinity(0)
    except (AttributeError, OSError):
        allowed = set(range(os.cpu_count() or 1))
    try:
        cur = {}
        for line in open("/proc/cpuinfo"):
            if ":" in line:
                k, v = [t.strip() for t in line.split(":", 1)]
                cur[k] = v
                if k == "model name" and model == "unknown":
                    model = v
            elif cur:
                if int(cur.get("processor", -1)) in allowed:
                    cores.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
                cur = {}
        if cur and int(cur.get("processor", -1)) in allowed:
            cores.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
    except (OSError, ValueError):
        pass
    return model, (len(cores) or len(allowed)), len(allowed)


def cpu_baseline(host_audio, fs, budget_s=9.0):
    """EBU R128 + true-peak of the reference (or the port) on host cores over a bounded sample (SURVEY.md 8d ii).
    host_audio: float32 [n, T, 2] copied from the benchmark buffers.  `value` is a SINGLE thread — the reference's operating
    mode (one instance on one RT thread) — over EBU R128 process() + process_max() x2; `per_meter` times each meter alone on
    that thread (EBU / true peak x2 / 30-band bank, frames/s); `all_cores` runs one instance per PHYSICAL core of the host
    (streams are independent; the foreign calls release the GIL).  About 15 s of wall in all.  A reported baseline, never the target."""
    import numpy as np
    from _oracle import Oracle, Reference, have_reference
    impl, kind = (Reference(), "reference") if have_reference() else (Oracle(), "port")
    n, T = host_audio.shape[0], host_audio.shape[1]
    model, phys, logical = host_cpu()
    done, t0 = 0, time.perf_counter()
    for s in range(n):
        impl.ebu(host_audio[s], fs, 1024)
        impl.tp(host_audio[s], fs, 1024)
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    out = {"value": 2.0 * done * T / dt, "unit": "samples/s", "cores": 1, "kind": kind,
           "sample": f"{done} streams x {T / fs:.0f} s of the benchmark's own buffers, EBU R128 process() + "
                     f"process_max() x2, block 1024, {dt:.1f} s wall",
           "cpu_model": model, "host_physical_cores": phys, "host_cpus": logical}
    # each meter alone on the same thread (frames/s = stereo frames/s; true peak = both channels)
    try:
        split = {}
        for name, fn, budget in (("ebu_r128", lambda a: impl.ebu(a, fs, 1024), 1.2), ("truepeak_x2", lambda a: impl.tp(a, fs, 1024), 1.2),
                                 ("bank_30band", lambda a: impl.spectr(a, fs, 1024), 1.5)):
            k, t1 = 0, time.perf_counter()
            while True:
                fn(host_audio[k % n]); k += 1
                if time.perf_counter() - t1 > budget:
                    break
            split[name] = {"frames_per_s": k * T / (time.perf_counter() - t1), "streams": k}
        out["per_meter"] = split
    except Exception as exc:                                  # never let a context figure break the benchmark line
        out["per_meter"] = {"error": repr(exc)}
    try:
        from concurrent.futures import ThreadPoolExecutor
        w = max(1, phys)                                      # one instance per physical core (SURVEY.md 8d ii)

        def one(s):
            impl.ebu(host_audio[s], fs, 1024)
            impl.tp(host_audio[s], fs, 1024)

        per_stream_s = dt / max(done, 1)
        rounds = max(1, min(8, int(3.0 / max(per_stream_s, 1e-3))))      # ~3 s of wall: `rounds` streams per thread
        jobs = [s % n for s in range(rounds * w)]
        t1 = time.perf_counter()
        with ThreadPoolExecutor(w) as ex:
            list(ex.map(one, jobs))
        dm = time.perf_counter() - t1
        out["all_cores"] = {"value": 2.0 * len(jobs) * T / dm, "unit": "samples/s", "threads": w,
                            "sample": f"{len(jobs)} stream passes over {w} threads (one per physical core), {dm:.1f} s wall"}
    except Exception as exc:
        out["all_cores"] = {"error": repr(exc)}
    return out


def tpb_binding(S, T, k_ms, channels):
    """What bounds k_tpb (TruePeakdsp::process for a batch, mtr_tpb.hip), as a number one can divide by: a workgroup = one CU walks
    64 (stream, channel) columns in chunks of 16 frames, and a chunk costs every SIMD its share of what the twelve waves issue.
    Per chunk and CU (profiles/r06_tpb.md: instruction counts from SQ_INSTS_*, busy cycles from SQ_ACTIVE_INST_VALU x 4,
    SQ_VALU_MFMA_BUSY_CYCLES and SQ_LDS_IDX_ACTIVE, all / 7.68 M chunks): 725 VALU instructions = 2973 issue cycles, 72 MFMAs
    (16x16x32 f16: 4 blocks x 18) = 1152 cycles of the matrix pipe, 168 LDS instructions = 680 cycles of the LDS.  VALU issue and
    the matrix pipe do not overlap across the three waves of a SIMD (measured: their busy cycles ADD to 82 - 85 % of the kernel's), so the
    floor is their sum spread evenly over the four SIMDs — 1031 cycles per chunk — at the clock the profile ran at (2.05 GHz:
    GRBM_GUI_ACTIVE).  The chain waves alone (16 frames x 4 dependent steps of ~8 cycles + the maps' LDS round trip) need ~1040."""
    cols = S * channels
    chunks_per_cu = (cols + 63) // 64 * ((T + 15) // 16) / 256.0
    floor = 2973 / 4.0 + 1152 / 4.0
    clock_mhz = 2050.0
    return {"bound": "SIMD issue per 16-frame chunk: VALU (725 instructions per chunk and CU) + the matrix pipe (72 MFMAs), which do not overlap across the "
                     "waves of a SIMD; one barrier per chunk; the LDS is busy 680 cycles per chunk beside it",
            "valu_issue_cycles_per_chunk_and_simd": 2973 / 4.0, "mfma_pipe_cycles_per_chunk_and_simd": 1152 / 4.0,
            "lds_busy_cycles_per_chunk_and_cu": 680.0, "issue_floor_cycles_per_chunk_and_simd": floor,
            "chain_wave_alone_cycles_per_chunk": 1040.0, "profiled_cycles_per_chunk": 1220.0,
            "chunks_per_cu": chunks_per_cu, "kernel_ms_at_floor_and_profiled_clock_2p05_ghz": chunks_per_cu * floor / (clock_mhz * 1e3),
            "kernel_ms_at_floor_and_2p4_ghz": chunks_per_cu * floor / 2.4e6,
            "frac_of_floor_at_profiled_clock": chunks_per_cu * floor / (clock_mhz * 1e3) / k_ms,
            "mfma_achieved_tflops": 2.0 * 16 * 16 * 32 * 72 * chunks_per_cu * 256 / (k_ms * 1e-3) / 1e12, "mfma_peak_tflops": 2500.0}


def bank_binding(S, T, k_ms):
    """k_bank's own floor (mtr_bank.hip; profiles/r06_bank.md): 28.75 VALU instructions per (frame, wave) in the compiled loop — 25 fp64
    (19 fma, 5 add, 1 mul), a v_cvt_f32_f64, 2 fp32 fma, half a v_max3, a quarter of the loop's own — on lanes = (stream, band); a pure
    register-to-register stream of that mix issues at 2.00 ns per wave-instruction and SIMD with four waves resident (tools/ubench64.hip on
    the same chip: the fp64 unit under the power cap, ~1.95 GHz x 4 cycles) and the launch's 3840 waves are 3.75 per SIMD: the fullest SIMDs carry four."""
    waves = (S * 30 + 63) // 64
    per_simd = -(-waves // 1024)                                  # waves on the fullest SIMD
    inst = 28.75 * T * per_simd
    ns = 2.00
    return {"bound": "fp64 VALU issue (25 fp64 + 3.75 other instructions per frame and band; lane = (stream, band)) on the fullest SIMDs",
            "valu_instructions_per_frame_and_wave": 28.75, "waves": waves, "waves_on_the_fullest_simd": per_simd, "mean_waves_per_simd": waves / 1024.0,
            "ns_per_wave_instruction_and_simd_at_4_waves": ns, "kernel_ms_at_floor": inst * ns * 1e-6,
            "frac_of_floor": inst * ns * 1e-6 / k_ms}


def end_to_end_host(M, torch, buf, fs, meters, device, n=1024, reps=3):
    """Host-resident audio: n streams of the benchmark's own buffers in pageable memory -> mtr_engine_process_host."""
    T = buf.shape[1]
    x = buf[:n].cpu().numpy()
    dst = torch.empty_like(buf[:n])
    src = torch.from_numpy(x)
    dst.copy_(src); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        dst.copy_(src)
    torch.cuda.synchronize()
    h2d = x.nbytes * reps / (time.perf_counter() - t0)
    del dst
    with M.Engine(n, fs, meters, device=device) as e:
        e.integr_start()
        e.process(x); e.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            e.process(x)
        e.sync()
        dt = (time.perf_counter() - t0) / reps
        e.timing_enable(True)
        e.process(x)
        q = e.timing_query()
    return {"GB_per_s": x.nbytes / dt / 1e9, "frames_per_s": n * T / dt, "samples_per_s": 2.0 * n * T / dt,
            "plain_h2d_GB_per_s": h2d / 1e9, "frac_of_pcie": (x.nbytes / dt) / h2d,
            "chunks": q["calls"], "kernel_ms_total": q["ms_fused"] + q["ms_gate"], "wall_ms": dt * 1e3,
            "sample": f"{n} streams x {T / fs:.0f} s = {x.nbytes / 1e9:.2f} GB of pageable host memory, {reps} passes; "
                      "256 MiB chunks of streams, the copy of chunk k + 1 under the kernels of chunk k"}


def kernel_sha():
    """Hash of the sources of the dominant kernel — and of the header that fixes how much of a stream its segments re-read
    for their warm-up: the committed PMC traffic figure is valid for exactly this code."""
    import hashlib
    h = hashlib.sha256()
    for f in ("mtr_seg.hip", "mtr_mfma16_fir.h", "mtr_internal.h", "Makefile"):
        h.update(open(os.path.join(ROOT, "meters.lv2_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def committed_traffic(meters, S, T, layout):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, MI355X_MICROARCH.md):
    the counters cannot be read from inside this process, so the figure is reported only for the very workload AND the
    very kernel sources it was measured on (profiles/r06_traffic.json carries their hash); otherwise null."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r06_traffic.json")))
        w = tj["workload"]
        if (meters, S, T, layout) == (w["meters"], w["streams_per_gpu"], w["frames_per_stream"], w["layout"]) \
                and tj["kernel_sha16"] == kernel_sha():
            return tj["traffic_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def self_launch(n):
    """`python bench.py --gpus N` with no launcher around it: become `torch.distributed.run --nproc-per-node N bench.py ...`
    (the command the driver itself uses), rendezvous on 127.0.0.1 and a free port.  Rank 0 prints the one JSON line."""
    import random
    import socket
    # a free port BELOW the kernel's ephemeral range (32768 ..): a port handed out by bind (0) can be given to somebody's outgoing
    # connection between this probe and the launcher's listen () — seen once on a GPU box: EADDRINUSE, no line
    port = None
    for _ in range(64):
        cand = random.randrange(20000, 32000)
        with socket.socket() as so:
            try:
                so.bind(("127.0.0.1", cand))
            except OSError:
                continue
        port = cand
        break
    if port is None:
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    # what the multi-rank tests set: this host's driver only supports dmabuf IPC, and without it RCCL's (and torch's)
    # cross-process device-memory handles fail with "hipIpcGetMemHandle: invalid argument" — the job would then land on
    # the gloo fallback, silently but for config.collective
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def ctrl_timeout():
    """Deadline of every collective of the control plane (gloo): a rank that is lost surfaces as an exception on the others
    after this long — and rank 0 prints a line with an `error` field — well inside the driver's 1800 s, instead of after gloo's
    default 30 minutes.  MTR_BENCH_CTRL_TIMEOUT_S overrides (default 300)."""
    import datetime
    return datetime.timedelta(seconds=float(os.environ.get("MTR_BENCH_CTRL_TIMEOUT_S", "300")))


_REAL_STDOUT = None


def own_stdout():
    """Stdout carries ONE JSON line: libraries that write to the C-level stdout (RCCL prints its version banner there when a
    communicator is created) are sent to stderr from here on; `emit` writes the line to the stdout the process was started with."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    line = (json.dumps(obj) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(line.decode()); sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, line)


def dry_run(args, rank, world):
    """MTR_BENCH_DRY_RUN=1: the launcher logic without a GPU (tests/test_bench_launch.py) — ranks rendezvous over gloo, shard
    the job's streams, agree on the shards and rank 0 prints them.  Never a measurement."""
    import torch
    import torch.distributed as dist
    from meters.lv2_amd import dist as mdist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", timeout=ctrl_timeout())
    first, count = mdist.shard(args.streams * world, world, rank)
    mine = torch.tensor([rank, int(os.environ.get("LOCAL_RANK", "0")), first, count], dtype=torch.int64)
    rows = [torch.zeros(4, dtype=torch.int64) for _ in range(world)]
    if world > 1:
        dist.all_gather(rows, mine)
        dist.barrier()
    else:
        rows = [mine]
    if rank == 0:
        emit({"dry_run": True, "n_gpus": world, "streams_per_gpu": args.streams, "ranks": [[int(v) for v in r] for r in rows],
              # the shape of the real line's proof of the collective (config.rccl_nranks / rank_devices): no communicator in a dry run
              "rccl_nranks": None, "rank_devices": [{"rank": int(r[0]), "hip_device": int(r[1]), "rccl_device": None} for r in rows]})
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=8192, help="streams per GPU")
    ap.add_argument("--seconds", type=float, default=10.0, help="audio seconds per stream per step")
    ap.add_argument("--fs", type=float, default=48000.0, help="sample rate (the headline is 48 kHz; 44100 exercises fragments that are not whole 16-frame steps)")
    ap.add_argument("--meters", default="ebu+tp", choices=["ebu+tp", "ebu", "tp", "ebu+tp+spectr30", "spectr30",
                                                           "bitstats", "sigdist", "tpb", "dr14", "kmeter"])
    ap.add_argument("--run", type=int, default=0, help="frames per lane run (0 = engine default)")
    ap.add_argument("--segments", type=int, default=0)
    ap.add_argument("--layout", type=int, default=0, help="0 auto (7 with true peak, 4 without), 3 exact-f32 VALU interpolator, 4 K-weighting only, "
                                                          "6 matrix pipe at f32 grade with a wave per (stream, segment), 7 = 6 + lane = segment for the calls that fit")
    ap.add_argument("--signal", type=int, default=1, help="synthetic signal: 1 programme-like (default), 0 stationary noise, 2 noise under a monotonically rising level")
    ap.add_argument("--no-extra", action="store_true", help="skip extra.configs / extra.lv2_run_latency (N = 1 only anyway)")
    ap.add_argument("--fir", type=int, default=0, help="0 auto (mirror-symmetric form), 1 dense 3x48 taps")
    ap.add_argument("--prune", type=int, default=0, help="1 = exact true-peak pruning (identical result, data-dependent speed)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tail", type=int, default=0, help="0 auto (deferred for batches), 1 serial (everything on the caller's stream), 2 always deferred")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)                                 # does not return

    own_stdout()
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started %d ranks" % (args.gpus, world))
    done = []

    def error_line(msg):
        if rank == 0 and not done:
            done.append(1)
            emit({"metric": "audio samples/s (48 kHz stereo) EBU R128 + true-peak", "value": None, "unit": "samples/s",
                  "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "error": msg})

    if world > 1 and rank == 0:
        # torch.distributed.run answers a rank that died with SIGTERM for the others: rank 0 still owes the driver its ONE line
        import signal

        def on_term(signum, frame):
            error_line("terminated by the launcher (signal %d): a rank was lost" % signum)
            os._exit(1)
        signal.signal(signal.SIGTERM, on_term)
    try:
        fault = os.environ.get("MTR_BENCH_FAULT", "")            # tests only
        if fault.startswith("hang_before_rendezvous:") and int(fault.split(":")[1]) == rank:
            time.sleep(float(fault.split(":")[2]))
            os._exit(3)
        if os.environ.get("MTR_BENCH_DRY_RUN") == "1":
            return dry_run(args, rank, world)
        return run(args, rank, local, world)
    except BaseException as exc:                                  # noqa: BLE001 — the driver gets ONE line whatever happened
        if isinstance(exc, SystemExit) and not exc.code:
            raise
        import traceback
        traceback.print_exc(file=sys.stderr)
        error_line("%s: %s" % (type(exc).__name__, exc))
        os._exit(1)                                               # (no destructor may wait for a rank that is gone)


def run(args, rank, local, world):
    import numpy as np
    import torch
    import torch.distributed as dist
    import meters.lv2_amd as M
    from meters.lv2_amd import dist as mdist

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the engine has no CPU path")
    # MTR_BENCH_SHARED_GPU=1: every rank on GPU 0 — a rehearsal of the N > 1 control and reduction flow on a one-GPU box
    # (RCCL refuses two ranks on one device, so the agreed fallback reduces); never a measurement.
    shared = os.environ.get("MTR_BENCH_SHARED_GPU") == "1"
    if shared:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        # torch.distributed is the CONTROL plane only (the 128-byte communicator id, barriers, the max over ranks of the
        # clock): gloo on 127.0.0.1.  The job's one RCCL communicator is the engine's own (mtr_comm_init, below) — no second
        # one is created beside it, so nothing can race its ncclCommInitRank on the same devices.
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", timeout=ctrl_timeout())

    fs = args.fs
    S, T = args.streams, int(round(args.seconds * fs))
    meters = {"ebu+tp": M.METER_EBU | M.METER_TRUEPEAK, "ebu": M.METER_EBU, "tp": M.METER_TRUEPEAK,
              "ebu+tp+spectr30": M.METER_EBU | M.METER_TRUEPEAK | M.METER_SPECTR30,
              "spectr30": M.METER_SPECTR30, "bitstats": M.METER_BITSTATS, "sigdist": M.METER_SIGDIST,
              "tpb": M.METER_TPBALLIST, "dr14": M.METER_DR14, "kmeter": M.METER_KMETER}[args.meters]
    mono = bool(meters & (M.METER_BITSTATS | M.METER_SIGDIST))   # integer paths: the same buffer read as [S][2T] mono

    free, _ = torch.cuda.mem_get_info()
    need = S * T * 8
    if need > 0.92 * free:
        sys.exit(f"batch of {need / 1e9:.1f} GB does not fit the {free / 1e9:.1f} GB free on this GPU")
    buf = torch.empty((S, T, 2), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    # synthetic programme-like signal, a different LCG seed per stream across the whole job
    first, _ = mdist.shard(S * world, world, rank)
    M.synth_fill_device(buf.data_ptr(), S, T, T, 777 + first, fs, args.signal, stream)
    agg_hist = torch.zeros(2 * 751, dtype=torch.int32, device=dev)
    agg_max = torch.zeros(4, dtype=torch.float32, device=dev)

    eng = M.Engine(S, fs, meters, n_channels=1 if mono else 2, device=local, tune_run=args.run,
                   tune_segments=args.segments, tune_layout=args.layout, tune_fir=args.fir, tune_prune=args.prune)
    eng.integr_start()
    eng.set_deferred_tail(args.tail)
    # The job's communicator: RCCL behind the C ABI.  Every rank is past its allocations and its first kernels, and the ranks
    # AGREE on how they reduce (meters.lv2_amd.dist.agree_on_collective: a vote over the gloo control plane; fallbacks: a
    # torch.distributed NCCL (= RCCL) group, then gloo on the same device buffers — what the shared-GPU rehearsal ends up with).
    torch.cuda.synchronize()

    # The first contact with RCCL cannot hang the job: the communicator is built with a deadline (mtr_comm_init_timeout:
    # ncclCommInitRankConfig non-blocking + ncclCommGetAsyncError polled), the first collective on it is probed with a
    # deadline (mtr_comm_probe), and the ranks vote after each.  MTR_BENCH_COMM_TIMEOUT_S (default 120) is that deadline.
    comm_timeout_s = float(os.environ.get("MTR_BENCH_COMM_TIMEOUT_S", "120"))
    fault = os.environ.get("MTR_BENCH_FAULT", "")                # tests only: "sleep_in_init:<rank>:<seconds>"

    def make():
        if shared and os.environ.get("MTR_BENCH_TRY_RCCL") != "1":
            raise RuntimeError("not tried (MTR_BENCH_SHARED_GPU)")
        def late():                                               # (this rank holds the id and keeps the others waiting inside RCCL)
            if fault.startswith("sleep_in_init:") and int(fault.split(":")[1]) == rank:
                time.sleep(float(fault.split(":")[2]))
        return mdist.make_comm(rank, world, local, timeout_ms=int(1e3 * comm_timeout_s) if world > 1 else 0, before_init=late)

    t_neg = time.perf_counter()
    comm, group, collective = mdist.agree_on_collective(
        rank, world, make, device=dev, allow_nccl=not shared, probe_timeout_s=comm_timeout_s,
        log=lambda d: print("bench.py: rank %d: %s" % (rank, d), file=sys.stderr))
    negotiation_ms = 1e3 * (time.perf_counter() - t_neg)

    def step():
        if mono:
            eng.process_device(buf.data_ptr(), 2 * T, 2 * T, stream)
            return
        eng.process_device(buf.data_ptr(), T, T, stream)
        if meters & (M.METER_EBU | M.METER_TRUEPEAK):
            if comm is not None:
                eng.reduce(comm, agg_hist.data_ptr(), agg_max.data_ptr(), stream)   # the only collective of the job (RCCL)
            else:
                eng.aggregate_device(agg_hist.data_ptr(), agg_max.data_ptr(), stream)
                mdist.all_reduce_aggregate(agg_hist, agg_max, group)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    eng.timing_enable(True)
    # per-step GPU time on the launching stream (the engine runs on torch's current stream here, so torch's events see it):
    # the median beside the mean — the headline kernel is power-limited and its clock moves within a run
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        marks[i].record()
        step()
    marks[args.steps].record()
    torch.cuda.synchronize()
    t_own = time.perf_counter() - t0                              # this rank's own loop, before it waits for the others
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    per_call = eng.timing_calls()
    tq = eng.timing_query()
    # this rank's own GPU time per step: the fused kernel + the gate — or the fused kernel alone where the gate ran deferred, inside the next
    # step's fused kernel (its column is then elapsed time on the side stream, not time the step needed)
    gate_cols = per_call[:, 1].sum() if eng.deferred_calls() == 0 else 0.0
    my_kernel_ms = float(per_call[:, 0].sum() + gate_cols) / max(len(per_call), 1)
    ranks_ms = [[1e3 * t_own / args.steps, my_kernel_ms]]
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        rows = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(rows, torch.tensor(ranks_ms[0], dtype=torch.float64))
        ranks_ms = [[float(v) for v in r] for r in rows]

    # RCCL's own view of the job's communicator: ranks it holds, the device it bound this rank to (and HIP's ordinal beside it)
    rccl_nranks = comm.nranks() if comm is not None else None
    mine_dev = [rank, local, comm.device() if comm is not None else -1]
    rank_devices = [mine_dev]
    if world > 1:
        rows = [torch.zeros(3, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(rows, torch.tensor(mine_dev, dtype=torch.int64))
        rank_devices = [[int(v) for v in r] for r in rows]
    rank_devices = [{"rank": r[0], "hip_device": r[1], "rccl_device": (r[2] if r[2] >= 0 else None)} for r in rank_devices]
    deferred_calls = eng.deferred_calls()
    if rank == 0:
        frames_job = float(world) * S * T * args.steps
        ms_step = 1e3 * dt / args.steps
        out = {
            "metric": "audio samples/s (48 kHz stereo) EBU R128 + true-peak",
            "value": 2.0 * frames_job / dt,
            "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step,
            "ms_per_step_median": step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2]),
            "ms_per_step_min_max": [step_ms[0], step_ms[-1]],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.meters != "spectr30" else "f64",
            "data": "synthetic",
            "config": {"workload": f"{args.meters}: {S} streams/GPU x {args.seconds:g} s, 48 kHz stereo f32 "
                                   f"(per-GPU shard of BASELINE configs[4]); integration on; "
                                   f"per-step RCCL all-reduce of 2x751 histograms + peaks",
                       "streams_per_gpu": S, "frames_per_stream": T, "sample_rate": fs,
                       "frames_per_s": frames_job / dt, "parallelism": f"streams sharded x{world}",
                       "collective": collective, "rccl_version": M.engine.rccl_version(),
                       # what RCCL itself says about the communicator the reduction ran on (ncclCommCount / ncclCommCuDevice per
                       # rank): null when the ranks agreed on a fallback — the string above is the host's word, this is RCCL's
                       "rccl_nranks": rccl_nranks, "rank_devices": rank_devices,
                       "tail": ("deferred: k_gate + k_aggregate + the all-reduce of step i on the engine's side stream, beside k_seg of step i + 1"
                                if deferred_calls else "serial: everything on the caller's stream"),
                       "deferred_calls": deferred_calls,
                       "comm_init_ms": getattr(comm, "init_ms", None), "comm_probe_ms": getattr(comm, "probe_ms", None),
                       "comm_negotiation_ms": negotiation_ms, "comm_timeout_s": comm_timeout_s if world > 1 else None},
        }
        if world > 1:
            # What the ranks did on their own: every rank's loop before it waited for the others, and the GPU time of
            # its own kernels (fused + gate, HIP events).  `speedup_vs_ideal` = the job's rate over the rate N ranks would have
            # if each ran at its own kernels' speed with no collective, no launch gaps and no skew between ranks.
            out["per_rank_ms"] = [r[0] for r in ranks_ms]
            out["per_rank_kernel_ms"] = [r[1] for r in ranks_ms]
            ideal = sum(2.0 * S * T / (r[1] * 1e-3) for r in ranks_ms if r[1] > 0)
            out["speedup_vs_ideal"] = (out["value"] / ideal) if ideal > 0 else None
        layout = eng.layout()
        if tq["calls"] and (meters & (M.METER_EBU | M.METER_TRUEPEAK)):
            k_ms = tq["ms_fused"] / tq["calls"]
            achieved = S * T * BYTES_PER_FRAME / (k_ms * 1e-3) / 1e9
            kname = "k_seg" if (layout == 7 and eng.seg_stats()[0]) else {7: "k_kwtp16", 6: "k_kwtp16", 4: "k_kw"}.get(layout, "k_fused2")
            out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": committed_traffic(args.meters, S, T, layout),
                               "kernel": kname, "kernel_ms": k_ms, "gate_ms": tq["ms_gate"] / tq["calls"],
                               "kernel_ms_median": float(np.median(per_call[:, 0])), "kernel_ms_min_max": [float(per_call[:, 0].min()), float(per_call[:, 0].max())],
                               "frac_at_kernel_ms_median": S * T * BYTES_PER_FRAME / (float(np.median(per_call[:, 0])) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               # the whole step (kernel + gate + aggregate + the RCCL reduce): what `value` is made of
                               "whole_step_frac": S * T * BYTES_PER_FRAME / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "algorithmic_bytes_per_launch": S * T * BYTES_PER_FRAME}
            if layout in (6, 7):
                seg = layout == 7 and eng.seg_stats()[0] > 0
                # What binds: issue cycles of the SIMDs under the socket's power cap, not HBM.  Per 16 frames of 64 lanes (one step
                # of a k_seg wave, 1024 stereo frames) the matrix pipe is busy 144 MFMAs x 16 cycles = 2304 cycles (three f16
                # partial products for f32 grade).  The issue model of one wave (tools/issue_model.hip, profiles/r03_issue_model.txt):
                # an MFMA with the two full-rate (or one half-rate) VALU instructions that ride behind it issues every 19.2 cycles,
                # the K-weighting recurrence that does not fit there (128 of its 176 packed instructions per step) runs as one
                # packed block at 5.1 cycles per instruction + one 17-cycle wait for the matrix pipe, an LDS instruction costs 6.
                # profiles/r04_kseg_step_cycles.txt has the measured step (4.07 k cycles; 3.11 k without the recurrence) and
                # profiles/r04a_kseg_ebu_tp.md the clock the power cap allows under this kernel (1.585 GHz of 2.4; r03: 1.56 - 1.60) —
                # both are properties of the committed profile, not of this run; this run contributes kernel_ms.
                steps = S * T / 1024.0
                n_simd = 1024.0
                ebu_on = bool(meters & M.METER_EBU)
                mfma_cycles = steps * 144 * 16 / n_simd
                model_step = 144 * 19.2 + 40 * 6.0 + ((128 * 5.1 + 17) if ebu_on else 0)
                floor_cycles = steps * model_step / n_simd
                out["roofline"]["binding_roofline"] = {
                    "bound": "SIMD issue cycles under the power cap: f16 MFMA (3 partial products) with two VALU riders each + the packed recurrence block + LDS",
                    "mfma_pipe_cycles_per_step": 144 * 16, "issue_model_cycles_per_step": model_step,
                    "profiled_cycles_per_step": (4069.8 if ebu_on else 3113.1) if seg else None,
                    "mfma_pipe_cycles_per_simd": mfma_cycles, "issue_floor_cycles_per_simd": floor_cycles,
                    "kernel_ms_at_floor_and_2p4_ghz": floor_cycles / 2.4e6, "kernel_ms_at_floor_and_profiled_clock_1p58_ghz": floor_cycles / 1.58e6,
                    "frac_of_floor_at_profiled_clock": (floor_cycles / 1.58e6) / k_ms,
                    "mfma_achieved_tflops": 2.0 * 16 * 16 * 32 * 144 * steps / (k_ms * 1e-3) / 1e12, "mfma_peak_tflops": 2500.0}
                out["roofline"]["note"] = (("lane = time segment (k_seg): " if seg else "wave per (stream, segment) (k_kwtp16): ") +
                                           "K-weighting = the reference's recurrence in packed f32; interpolator on the matrix pipe at f32 grade "
                                           "(samples and taps as two f16 halves, three partial products, f32 accumulation): same 2e-6 parity "
                                           "bound as the exact-f32 VALU path")
                out["dtype"] = "f32 (K-filter: packed f32 VALU; interpolator: f16x2-split samples x f16x2-split taps on MFMA, f32 accumulate)"
            elif layout == 3 and meters & M.METER_TRUEPEAK:
                valu_ops = 143.0 * S * T / (k_ms * 1e-3) * 2 * 2        # flop/s: 2 lanes x FMA
                out["roofline"]["binding_roofline"] = {"bound": "fp32 VALU (v_pk_fma_f32)", "achieved_tflops": valu_ops / 1e12,
                                                       "peak_tflops": 157.3, "frac": valu_ops / 157.3e12}
        elif tq["calls"] and tq["ms_bank"] > 0:
            k_ms = tq["ms_bank"] / tq["calls"]
            achieved = S * T * BYTES_PER_FRAME / (k_ms * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                               "kernel": {"spectr30": "k_bank", "bitstats": "k_bitstats", "sigdist": "k_sigdist",
                                          "tpb": "k_tpb", "dr14": "k_dr14_sums", "kmeter": "k_kmeter_pieces"}.get(args.meters, "k_bank"),
                               "kernel_ms": k_ms}
            if args.meters == "tpb":
                out["roofline"]["binding_roofline"] = tpb_binding(S, T, k_ms, 2)
            elif args.meters == "spectr30":
                out["roofline"]["binding_roofline"] = bank_binding(S, T, k_ms)
        if mono:
            emit(out)
            eng.close()
            return
        res = eng.results(0, 1)[0]
        out["check"] = {"stream0_integrated_lufs": res.integrated, "stream0_dbtp":
                        float(20 * np.log10(max(res.truepeak[0], res.truepeak[1], 1e-30))),
                        "job_max_truepeak": float(agg_max[:2].max().item())}
        if world == 1 and not args.no_cpu_baseline and (meters & (M.METER_EBU | M.METER_TRUEPEAK)):
            n = min(S, 256)
            out["cpu_baseline"] = cpu_baseline(buf[:n].cpu().numpy(), fs)
        headline = world == 1 and not args.no_cpu_baseline and not args.no_extra and args.meters == "ebu+tp" and not args.prune \
            and args.layout == 0 and (S, T) == (8192, 480000) and fs == 48000.0
        if headline:
            extra = {}
            peaks = eng.truepeak()
            o9 = eng.out9()

            def timed(eS, eT, emeters, steps=3, efs=None, **kw):
                """ms per launch of the fused / gate / bank kernels for eS streams x eT frames of the same buffer."""
                with M.Engine(eS, efs or fs, emeters, device=local, **kw) as x:
                    if emeters & M.METER_EBU:
                        x.integr_start()
                    x.process_device(buf.data_ptr(), eT, eT, stream)
                    torch.cuda.synchronize()
                    x.timing_enable(True)
                    t0 = time.perf_counter()
                    for _ in range(steps):
                        x.process_device(buf.data_ptr(), eT, eT, stream)
                    torch.cuda.synchronize()
                    wall = 1e3 * (time.perf_counter() - t0) / steps
                    q = x.timing_query()
                    c = max(q["calls"], 1)
                    keep = {"tp": x.truepeak() if emeters & M.METER_TRUEPEAK else None,
                            "o9": x.out9() if emeters & M.METER_EBU else None, "prune": x.prune_stats(), "refine": x.refine_stats(),
                            "layout": x.layout(),
                            "kernel": "k_seg" if x.seg_stats()[0] else {7: "k_kwtp16", 6: "k_kwtp16", 4: "k_kw", 3: "k_fused2"}.get(x.layout(), "?")}
                return q["ms_fused"] / c, q["ms_gate"] / c, q["ms_bank"] / c, wall, keep

            def frac(eS, eT, ms):
                return eS * eT * BYTES_PER_FRAME / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS

            # Next to — never instead of — `value`: the wave-per-segment kernel this layout falls back to (k_kwtp16, round 2's
            # default), its exact peak pruning (bit-identical peaks, data-dependent speed), and the exact-f32 VALU interpolator
            # of round 1 (layout 3), whose peaks the matrix-pipe paths must reproduce.
            f, g, _, _, k6 = timed(S, T, meters, tune_layout=6)
            rel = np.abs(k6["tp"].astype(np.float64) - peaks) / np.maximum(peaks.astype(np.float64), 1e-30)
            extra["wave_per_segment_k_kwtp16"] = {"kernel": "k_kwtp16", "kernel_ms": f, "frac": frac(S, T, f),
                                                  "max_rel_dev_of_default_peaks": float(rel.max()),
                                                  "max_abs_dev_integrated_lufs": float(np.abs(k6["o9"][:, 4].astype(np.float64) - o9[:, 4]).max()),
                                                  "note": "tune_layout=6: round 2's default; serves the calls k_seg does not fit"}
            # exact pruning with evidence — on the bench programme, on stationary noise without an envelope and
            # on a level that rises monotonically (every block beats everything before it: nothing can be pruned)
            pr = {}
            tmp = torch.empty_like(buf)
            for name, kind in (("programme (the bench signal)", 1), ("stationary noise, no envelope", 0), ("monotonically rising level (worst case)", 2)):
                src = buf
                if kind != 1:
                    M.synth_fill_device(tmp.data_ptr(), S, T, T, 777 + first, fs, kind, stream)
                    src = tmp
                row = {}
                for lvl in (0, 1, 2):
                    with M.Engine(S, fs, meters, device=local, tune_layout=6, tune_prune=lvl) as x:
                        x.integr_start()
                        x.process_device(src.data_ptr(), T, T, stream)
                        torch.cuda.synchronize()
                        first_skip = x.prune_stats()[1]
                        first_ref = x.refine_stats()
                        x.timing_enable(True)
                        for _ in range(3):
                            # every timed call starts from a reset engine: a second call on the same buffer would begin with
                            # the first one's (loud) last frames as history, and on the rising signal that alone prunes 83 %
                            x.reset()
                            x.integr_start()
                            x.process_device(src.data_ptr(), T, T, stream)
                        torch.cuda.synchronize()
                        q = x.timing_query()
                        ms = q["ms_fused"] / max(q["calls"], 1)
                        tpk = x.truepeak()
                        if lvl == 0:
                            dense_tp, dense_ms = tpk, ms
                            row["dense_kernel_ms"] = ms
                        else:
                            c, kk = x.prune_stats()
                            c, kk = c - c // 4, kk - first_skip     # (the three timed calls)
                            e = {"kernel_ms": ms, "frac": frac(S, T, ms), "vs_dense": ms / dense_ms, "tiles_skipped_frac": kk / max(c, 1),
                                 "peaks_identical_to_dense": bool(np.array_equal(tpk, dense_tp))}
                            if lvl == 2:
                                a_, b_ = x.refine_stats()
                                e["blocks_completed_frac"] = (b_ - first_ref[1]) / max(a_ - first_ref[0], 1)
                            row["tune_prune=%d" % lvl] = e
                pr[name] = row
            del tmp
            pr["note"] = ("k_kwtp16 (layout 6) only; exact branch and bound on L1 * max|x| per tile (1) and per 256-frame block after the first "
                          "f16 product (2): identical peaks, data-dependent speed — never part of `value`.  The worst case costs a few per "
                          "cent over dense; with k_seg as the default the pruning stays an option of layout 6 (it is a per-tile decision, "
                          "k_seg has no tiles) and is not the engine default.")
            extra["exact_pruning"] = pr
            f, g, _, _, k = timed(S, T, meters, tune_layout=3)
            rel = np.abs(k["tp"].astype(np.float64) - peaks) / np.maximum(peaks.astype(np.float64), 1e-30)
            extra["f32_valu_interpolator"] = {"kernel": "k_fused2", "kernel_ms": f, "frac": frac(S, T, f),
                                              "max_rel_dev_of_default_peaks": float(rel.max()),
                                              "max_abs_dev_integrated_lufs": float(np.abs(k["o9"][:, 4].astype(np.float64) - o9[:, 4]).max()),
                                              "note": "layout 3: bit-for-bit an fmaf chain on the VALU (round 1's default)"}
            # Every BASELINE config at its stated size (configs[0] is the CPU plumbing case: tests/test_lv2_plugin.py).
            cfgs = {}
            c1 = 3600 * 48000
            f, g, _, w, _ = timed(1, c1, M.METER_EBU, steps=5)
            cfgs["1: EBU R128, 1 stream x 3600 s"] = {"kernel": "k_kw", "kernel_ms": f, "gate_ms": g, "wall_ms": w, "frac": frac(1, c1, f),
                                                      "whole_step_frac": frac(1, c1, w), "bound": "hbm"}
            f, g, _, w, k = timed(1, c1, meters, steps=5)
            cfgs["1 + true peak: 1 stream x 3600 s"] = {"kernel": k["kernel"], "kernel_ms": f, "gate_ms": g, "wall_ms": w, "frac": frac(1, c1, f),
                                                        "whole_step_frac": frac(1, c1, w), "bound": "SIMD issue (MFMA + VALU)"}
            f, _, _, w, k = timed(1024, 60 * 48000, M.METER_TRUEPEAK)
            cfgs["2: 4x true peak, 1024 streams x 60 s"] = {"kernel": k["kernel"], "kernel_ms": f, "wall_ms": w, "frac": frac(1024, 60 * 48000, f),
                                                          "bound": "SIMD issue (MFMA) under the power cap"}
            _, _, bk, w, _ = timed(4096, T, M.METER_SPECTR30, steps=2)
            cfgs["3: 30-band bank, 4096 streams x 10 s"] = {"kernel": "k_bank", "kernel_ms": bk, "wall_ms": w, "frac": frac(4096, T, bk), "binding_roofline": bank_binding(4096, T, bk),
                                                           "bound": "fp64 VALU (ceiling 4.2-5.0 % of HBM peak, SURVEY.md 8d)"}
            f, g, bk, w, k = timed(S, T, meters | M.METER_SPECTR30, steps=2)
            cfgs["4: EBU + true peak + bank, 8192 streams x 10 s (one of 8 shards)"] = {
                "kernel_ms": {k["kernel"]: f, "k_gate": g, "k_bank": bk}, "wall_ms": w, "frac": frac(S, T, w),
                "bound": "fp64 VALU (k_bank) + SIMD issue (%s); the bank's second read of the audio is %.1f %% of the step" % (k["kernel"], 100 * (S * T * 8 / 5.0e12 * 1e3) / w)}
            # the other meter sets at the headline shape, under the same clock as everything else in this line
            f, g, _, w, k = timed(S, T, M.METER_EBU, steps=5)
            cfgs["EBU R128 only, 8192 streams x 10 s"] = {"kernel": k["kernel"], "kernel_ms": f, "gate_ms": g, "wall_ms": w, "frac": frac(S, T, f), "bound": "hbm"}
            f, _, _, w, k = timed(S, T, M.METER_TRUEPEAK, steps=5)
            cfgs["true peak only, 8192 streams x 10 s"] = {"kernel": k["kernel"], "kernel_ms": f, "wall_ms": w, "frac": frac(S, T, f),
                                                           "bound": "SIMD issue (MFMA) under the power cap"}
            T44 = 441000                                          # 10 s at 44.1 kHz: 200 fragments of 2205 frames, which are not whole 16-frame steps
            f, g, _, w, k = timed(S, T44, meters, steps=5, efs=44100.0)
            cfgs["EBU R128 + true peak at 44.1 kHz, 8192 streams x 10 s"] = {"kernel": k["kernel"], "kernel_ms": f, "gate_ms": g, "wall_ms": w,
                                                                           "frac": frac(S, T44, f), "bound": "SIMD issue under the power cap"}
            _, _, bk, w, _ = timed(S, T, M.METER_TPBALLIST, steps=2)
            cfgs["true-peak ballistics (TruePeakdsp::process), 8192 streams x 10 s"] = {
                "kernel": "k_tpb", "kernel_ms": bk, "wall_ms": w, "frac": frac(S, T, bk), "binding_roofline": tpb_binding(S, T, bk, 2),
                "bound": "what a SIMD can issue per 16-frame chunk: the chains (two waves: lane = (column, filter), 6.5 instructions per frame), the "
                         "products (two units per block on two waves, operands read an iteration ahead) with their maps, the split (two waves) — "
                         "VALU + matrix pipe busy 84 % of the time, three waves per SIMD, one barrier per chunk (DESIGN.md 3.5)"}
            extra["configs"] = cfgs
            # SURVEY.md 8d "Timing": the end-to-end figure of a host whose audio is NOT resident — pageable host memory through
            # mtr_engine_process_host (chunks of streams: chunk k + 1 crosses the link under the kernels of chunk k) — next to a
            # plain copy of the same bytes from the same memory.  Never `value`.
            try:
                extra["end_to_end_host"] = end_to_end_host(M, torch, buf, fs, meters, local)
            except Exception as exc:                              # never let a context figure break the benchmark line
                extra["end_to_end_host"] = {"error": repr(exc)}
            try:
                from _lv2host import Host
                from _lv2lat import run_latency
                host, lat = Host(), {}
                for name in ("EBUr128", "dBTPstereo", "spectr30stereo"):
                    lat[name] = {str(n): {k: round(v, 1) for k, v in run_latency(host, name, n, blocks=100, warm=10).items()}
                                 for n in (64, 256, 1024, 8192)}
                extra["lv2_run_latency"] = {"unit": "us per run() at 48 kHz, by block size (budget_us = the block's real time)", **lat}
            except Exception as exc:                              # never let a context figure break the benchmark line
                extra["lv2_run_latency"] = {"error": repr(exc)}
            out["extra"] = extra
        out["programme"] = mdist.programme_summary(agg_hist, agg_max)
        if args.prune:
            c, k = eng.prune_stats()
            out["prune"] = {"tile_passes": c, "skipped": k, "skipped_frac": k / max(c, 1),
                            "note": "exact branch-and-bound on L1*max|x|: identical peaks, data-dependent speed; "
                                    "NOT the default and not the dense headline number"}
        emit(out)
    eng.close()
    if comm is not None:
        comm.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
