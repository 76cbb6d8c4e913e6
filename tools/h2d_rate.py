"""tools/h2d_rate.py — the PCIe-inclusive rate of mtr_engine_process_host (pageable host memory -> staging
buffer -> kernels): never the benchmark's `value`, reported in DESIGN.md §5 for completeness."""
import time

import numpy as np

import meters.lv2_amd as M

S, T = 1024, 480000
x = np.random.default_rng(1).uniform(-0.5, 0.5, (S, T, 2)).astype(np.float32)
with M.Engine(S, 48000.0, M.METER_EBU | M.METER_TRUEPEAK) as e:
    e.integr_start()
    e.process(x)
    e.sync()
    t0 = time.perf_counter()
    for _ in range(3):
        e.process(x)
    e.sync()
    dt = (time.perf_counter() - t0) / 3
print(f"process_host: {S} streams x {T} frames = {x.nbytes / 1e9:.2f} GB in {dt * 1e3:.1f} ms "
      f"-> {x.nbytes / dt / 1e9:.1f} GB/s, {S * T / dt / 1e9:.2f} G frames/s")
