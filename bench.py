#!/usr/bin/env python3
"""bench.py — throughput of the meters.lv2 hot path on MI355X.

A "step" is one pass of the fused EBU R128 (K-weighting + gated loudness) + 4x true-peak path
over one batch of synthetic 48 kHz stereo audio that is already resident in HBM: per GPU,
8192 streams x 10 s = 3.93 G stereo frames = 31.5 GB (the per-GPU shard of BASELINE.json
configs[4]; the metric is quoted on batched EBU R128 + true-peak).  Streams are independent, so
ranks shard them with no data-path collective (weak scaling: per-GPU work is fixed); the only
RCCL traffic is the final all-reduce of the two 751-bin loudness histograms (sum) and the
peak / max-loudness values (max), done once per step.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  `value` counts channel-samples/s (2 per stereo frame), whole job.
`roofline` prices the dominant kernel (k_fused) at 8 algorithmic bytes per stereo frame (one read
of the input) against the 8 TB/s HBM peak, with the kernel's duration measured by
HIP events on the launching stream inside the timed steps.  `cpu_baseline` times the reference's
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


def cpu_baseline(host_audio, fs, budget_s=12.0):
    """EBU R128 + true-peak of the reference (or the port) on host cores over a bounded sample.
    host_audio: float32 [n, T, 2] copied from the benchmark buffers. Single thread: the
    reference's operating mode (one instance on one RT thread)."""
    import numpy as np
    from _oracle import Oracle, Reference, have_reference
    impl, kind = (Reference(), "reference") if have_reference() else (Oracle(), "port")
    n, T = host_audio.shape[0], host_audio.shape[1]
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
           "host_cpus": os.cpu_count()}
    # The whole host for context (SURVEY.md 8d ii): one instance per thread, streams are independent.  The
    # foreign calls release the GIL.  Still a reported baseline, not a target.
    try:
        from concurrent.futures import ThreadPoolExecutor
        w = max(1, min(os.cpu_count() or 1, 64, n))

        def one(s):
            impl.ebu(host_audio[s], fs, 1024)
            impl.tp(host_audio[s], fs, 1024)

        jobs = [s % n for s in range(max(n, 16 * w))]         # a few seconds of wall time
        t1 = time.perf_counter()
        with ThreadPoolExecutor(w) as ex:
            list(ex.map(one, jobs))
        dm = time.perf_counter() - t1
        out["all_cores"] = {"value": 2.0 * len(jobs) * T / dm, "threads": w,
                            "sample": f"{len(jobs)} stream passes over {w} threads, {dm:.1f} s wall"}
    except Exception as exc:                                  # never let the context figure break the benchmark line
        out["all_cores"] = {"error": repr(exc)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=8192, help="streams per GPU")
    ap.add_argument("--seconds", type=float, default=10.0, help="audio seconds per stream per step")
    ap.add_argument("--meters", default="ebu+tp", choices=["ebu+tp", "ebu", "tp", "ebu+tp+spectr30", "spectr30",
                                                           "bitstats", "sigdist", "tpb", "dr14", "kmeter"])
    ap.add_argument("--run", type=int, default=0, help="frames per lane run (0 = engine default)")
    ap.add_argument("--segments", type=int, default=0)
    ap.add_argument("--layout", type=int, default=0, help="0 auto, 1 wave per segment, 2/3 wave-specialised, 4 K-weighting only, 5 matrix-pipe interpolator")
    ap.add_argument("--fir", type=int, default=0, help="0 auto (mirror-symmetric form), 1 dense 3x48 taps")
    ap.add_argument("--prune", type=int, default=0, help="1 = exact true-peak pruning (identical result, data-dependent speed)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import meters.lv2_amd as M
    from meters.lv2_amd import dist as mdist

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the engine has no CPU path")
    # MTR_BENCH_SHARED_GPU=1: every rank on GPU 0 with the gloo backend — a rehearsal of the N > 1 control
    # flow on a one-GPU box (RCCL refuses two ranks on one device); never a measurement.
    shared = os.environ.get("MTR_BENCH_SHARED_GPU") == "1"
    if shared:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)     # "nccl" is RCCL on ROCm

    fs = 48000.0
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
    M.synth_fill_device(buf.data_ptr(), S, T, T, 777 + first, fs, 1, stream)
    agg_hist = torch.zeros(2 * 751, dtype=torch.int32, device=dev)
    agg_max = torch.zeros(4, dtype=torch.float32, device=dev)

    eng = M.Engine(S, fs, meters, n_channels=1 if mono else 2, device=local, tune_run=args.run,
                   tune_segments=args.segments, tune_layout=args.layout, tune_fir=args.fir, tune_prune=args.prune)
    eng.integr_start()

    def step():
        if mono:
            eng.process_device(buf.data_ptr(), 2 * T, 2 * T, stream)
            return
        eng.process_device(buf.data_ptr(), T, T, stream)
        if meters & (M.METER_EBU | M.METER_TRUEPEAK):
            eng.aggregate_device(agg_hist.data_ptr(), agg_max.data_ptr(), stream)
            mdist.all_reduce_aggregate(agg_hist, agg_max)   # the only collective of the job (RCCL)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    eng.timing_enable(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    tq = eng.timing_query()

    if rank == 0:
        frames_job = float(world) * S * T * args.steps
        ms_step = 1e3 * dt / args.steps
        out = {
            "metric": "audio samples/s (48 kHz stereo) EBU R128 + true-peak",
            "value": 2.0 * frames_job / dt,
            "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.meters != "spectr30" else "f64",
            "data": "synthetic",
            "config": {"workload": f"{args.meters}: {S} streams/GPU x {args.seconds:g} s, 48 kHz stereo f32 "
                                   f"(per-GPU shard of BASELINE configs[4]); integration on; "
                                   f"per-step RCCL all-reduce of 2x751 histograms + peaks",
                       "streams_per_gpu": S, "frames_per_stream": T, "sample_rate": fs,
                       "frames_per_s": frames_job / dt, "parallelism": f"streams sharded x{world}"},
        }
        if tq["calls"] and (meters & (M.METER_EBU | M.METER_TRUEPEAK)):
            k_ms = tq["ms_fused"] / tq["calls"]
            achieved = S * T * BYTES_PER_FRAME / (k_ms * 1e-3) / 1e9
            # HBM bytes per launch from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE), valid for the
            # profiled workload only; the counters cannot be read from inside this process
            traffic = None
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
                w = tj["workload"]
                if (args.meters, S, T) == (w["meters"], w["streams_per_gpu"], w["frames_per_stream"]):
                    traffic = tj["traffic_bytes_per_launch"]
            except (OSError, KeyError, ValueError):
                pass
            # the roofline that actually binds: packed fp32 VALU. Useful work = 120 (mirror-symmetric
            # interpolator) + 23 (K-weighting, two passes) packed operations per stereo frame
            valu_ops = 143.0 * S * T / (k_ms * 1e-3) * 2 * 2        # flop/s: 2 lanes x FMA
            out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                               "kernel": "k_fused2", "kernel_ms": k_ms, "gate_ms": tq["ms_gate"] / tq["calls"],
                               "algorithmic_bytes_per_launch": S * T * BYTES_PER_FRAME}
            if args.layout == 5:
                out["roofline"]["kernel"] = "k_kwtp"
                out["roofline"]["note"] = ("OPTIONAL layout 5: interpolator on the matrix pipe with f16-split samples "
                                           "(peaks within 0.0056 dB of the f32 interpolator, not bit-identical)")
                out["dtype"] = "f32 K-filter; f16x2-split samples, f16 taps, f32 accumulation in the interpolator"
            elif meters & M.METER_TRUEPEAK:
                out["roofline"]["binding_roofline"] = {"bound": "fp32 VALU (v_pk_fma_f32)", "achieved_tflops": valu_ops / 1e12,
                                                       "peak_tflops": 157.3, "frac": valu_ops / 157.3e12}
                out["roofline"]["note"] = ("fp32-VALU bound, not HBM bound: the 4x interpolator alone needs 120 packed "
                                           "VALU ops per 8-byte frame (SURVEY.md 8d: ceiling ~27% of HBM peak)")
            elif args.layout in (0, 4):
                out["roofline"]["kernel"] = "k_kw"              # K-weighting only: the HBM-bound kernel (mtr_kw.hip)
        elif tq["calls"] and tq["ms_bank"] > 0:
            k_ms = tq["ms_bank"] / tq["calls"]
            achieved = S * T * BYTES_PER_FRAME / (k_ms * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                               "kernel": {"spectr30": "k_bank", "bitstats": "k_bitstats", "sigdist": "k_sigdist",
                                          "tpb": "k_tpb", "dr14": "k_dr14_sums", "kmeter": "k_kmeter_pieces"}.get(args.meters, "k_bank"),
                               "kernel_ms": k_ms}
        if mono:
            print(json.dumps(out), flush=True)
            eng.close()
            return
        res = eng.results(0, 1)[0]
        out["check"] = {"stream0_integrated_lufs": res.integrated, "stream0_dbtp":
                        float(20 * np.log10(max(res.truepeak[0], res.truepeak[1], 1e-30))),
                        "job_max_truepeak": float(agg_max[:2].max().item())}
        if world == 1 and not args.no_cpu_baseline and (meters & (M.METER_EBU | M.METER_TRUEPEAK)):
            n = min(S, 256)
            out["cpu_baseline"] = cpu_baseline(buf[:n].cpu().numpy(), fs)
        if world == 1 and not args.no_cpu_baseline and args.meters == "ebu+tp" and not args.prune:
            # Reported NEXT TO the dense number, never instead of it: the same workload with exact peak
            # pruning (bit-identical results, tests/test_gpu_layouts.py; speed depends on the programme).
            with M.Engine(S, fs, meters, device=local, tune_run=args.run, tune_segments=args.segments,
                          tune_layout=args.layout, tune_fir=args.fir, tune_prune=1) as pe:
                pe.integr_start()
                pe.process_device(buf.data_ptr(), T, T, stream)
                torch.cuda.synchronize()
                pe.timing_enable(True)
                for _ in range(max(args.steps // 2, 1)):
                    pe.process_device(buf.data_ptr(), T, T, stream)
                torch.cuda.synchronize()
                pq = pe.timing_query()
                pc, pk = pe.prune_stats()
                p_ms = pq["ms_fused"] / max(pq["calls"], 1)
                same = bool(np.array_equal(pe.truepeak(), eng.truepeak()))
            out["exact_pruning"] = {"kernel_ms": p_ms, "frac": S * T * BYTES_PER_FRAME / (p_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    "tiles_skipped_frac": pk / max(pc, 1), "peaks_identical_to_dense": same,
                                    "note": "optional (tune_prune=1); not part of `value`"}
        if world == 1 and not args.no_cpu_baseline and args.meters == "ebu+tp" and not args.prune and args.layout != 5:
            # Also next to — not instead of — `value`: the same workload through layout 5 (mtr_fused3.hip), whose
            # true peaks carry the rounding of f16 taps: at most 0.0056 dB from the f32 interpolator.
            with M.Engine(S, fs, meters, device=local, tune_segments=args.segments, tune_layout=5) as me:
                me.integr_start()
                me.process_device(buf.data_ptr(), T, T, stream)
                torch.cuda.synchronize()
                me.timing_enable(True)
                for _ in range(max(args.steps // 2, 1)):
                    me.process_device(buf.data_ptr(), T, T, stream)
                torch.cuda.synchronize()
                mq = me.timing_query()
                m_ms = mq["ms_fused"] / max(mq["calls"], 1)
                a5, a3 = np.maximum(me.truepeak().astype(np.float64), 1e-30), np.maximum(eng.truepeak().astype(np.float64), 1e-30)
                ddb = float(np.abs(20 * np.log10(a5 / a3)).max())
                dlu = float(np.abs(me.out9()[:, 4].astype(np.float64) - eng.out9()[:, 4].astype(np.float64)).max())
                m_peaks = me.truepeak()
            with M.Engine(S, fs, meters, device=local, tune_segments=args.segments, tune_layout=5, tune_prune=1) as mp:
                mp.integr_start()
                mp.process_device(buf.data_ptr(), T, T, stream)
                torch.cuda.synchronize()
                mp.timing_enable(True)
                for _ in range(max(args.steps // 2, 1)):
                    mp.process_device(buf.data_ptr(), T, T, stream)
                torch.cuda.synchronize()
                mpq = mp.timing_query()
                mp_ms = mpq["ms_fused"] / max(mpq["calls"], 1)
                mc, mk = mp.prune_stats()
                mp_same = bool(np.array_equal(mp.truepeak(), m_peaks))
            out["matrix_pipe_interpolator"] = {
                "kernel": "k_kwtp", "kernel_ms": m_ms, "frac": S * T * BYTES_PER_FRAME / (m_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "max_abs_db_vs_f32_peaks": ddb, "bound_db": 0.0056, "integrated_lufs_max_abs_diff": dlu,
                "with_exact_pruning": {"kernel_ms": mp_ms, "frac": S * T * BYTES_PER_FRAME / (mp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                       "tiles_skipped_frac": mk / max(mc, 1), "peaks_identical_to_unpruned": mp_same},
                "note": "optional (tune_layout=5): f16-split samples x f16 taps on v_mfma_f32_32x32x16_f16, f32 accumulation; "
                        "within the +-0.01 dB parity tolerance but not bit-identical, so not part of `value`"}
        out["programme"] = mdist.programme_summary(agg_hist, agg_max)
        if args.prune:
            c, k = eng.prune_stats()
            out["prune"] = {"tile_passes": c, "skipped": k, "skipped_frac": k / max(c, 1),
                            "note": "exact branch-and-bound on L1*max|x|: identical peaks, data-dependent speed; "
                                    "NOT the default and not the dense headline number"}
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
