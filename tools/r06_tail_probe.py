#!/usr/bin/env python3
"""tools/r06_tail_probe.py [mode ...] — what does a second HIP stream cost k_seg?  8192 x 10 s, EBU + TP, 10 steps per mode:
  serial      everything on the caller's stream (tail mode 1), reduce each step
  deferred    gate + reduce on the side stream (tail mode 2)
  gate_only   deferred gate, no reduce at all
  noreduce    serial, no reduce at all
Prints k_seg ms (HIP events), gate ms, wall ms per step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import meters.lv2_amd as M

S, T, fs = 8192, 480000, 48000.0
modes = sys.argv[1:] or ["serial", "deferred", "gate_only", "noreduce"]
buf = torch.empty((S, T, 2), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
M.synth_fill_device(buf.data_ptr(), S, T, T, 777, fs, 1, st)
h = torch.zeros(2 * 751, dtype=torch.int32, device="cuda")
m = torch.zeros(4, dtype=torch.float32, device="cuda")
steps = int(os.environ.get("STEPS", "10"))
with M.Comm(0, 1, M.comm_unique_id(), 0) as comm:
    for mode in modes:
        with M.Engine(S, fs, M.METER_EBU | M.METER_TRUEPEAK) as e:
            e.integr_start()
            e.set_deferred_tail(2 if mode in ("deferred", "gate_only") else 1)
            red = mode in ("serial", "deferred")
            for _ in range(3):
                e.process_device(buf.data_ptr(), T, T, st)
                if red:
                    e.reduce(comm, h.data_ptr(), m.data_ptr(), st)
            torch.cuda.synchronize()
            e.timing_enable(True)
            t0 = time.perf_counter()
            for _ in range(steps):
                e.process_device(buf.data_ptr(), T, T, st)
                if red:
                    e.reduce(comm, h.data_ptr(), m.data_ptr(), st)
            torch.cuda.synchronize()
            wall = 1e3 * (time.perf_counter() - t0) / steps
            q = e.timing_query()
            print("%-10s k_seg %.3f ms  gate %.3f ms  wall/step %.3f ms" % (mode, q["ms_fused"] / q["calls"], q["ms_gate"] / q["calls"], wall), flush=True)
