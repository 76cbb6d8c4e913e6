"""tools/lat_ab.py — wall time of one host-path call (1 stream x 1024 frames: what an LV2 run() costs) for the engine named by MTR_LIB;
MTR_PY = a directory with another revision's python package (old libraries lack newer symbols)."""
import os, sys, time
sys.path.insert(0, os.environ.get("MTR_PY", os.getcwd()))
import numpy as np
import meters.lv2_amd as M
x = (np.random.default_rng(1).standard_normal((1, 1024, 2)) * 0.1).astype(np.float32)
for meters, nm in ((M.METER_EBU, "ebu"), (M.METER_EBU | M.METER_TRUEPEAK, "ebu+tp"), (M.METER_TPBALLIST, "tpb")):
    with M.Engine(1, 48000.0, meters) as e:
        if meters & M.METER_EBU: e.integr_start()
        for _ in range(50): e.process(x)
        ts = []
        for _ in range(400):
            t0 = time.perf_counter(); e.process(x); ts.append(time.perf_counter() - t0)
        ts = np.array(ts) * 1e6
        print(os.environ.get("MTR_LIB", "lib").split("/")[-2], nm, "median %.1f us  p99 %.1f" % (np.median(ts), np.percentile(ts, 99)))
