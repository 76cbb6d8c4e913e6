#!/usr/bin/env python3
"""tools/make_traffic.py <gpurun_out/prof_TAG/summary.txt> <profiles/rNN_traffic.json> — the HBM bytes per launch of the dominant kernel
(k_seg, EBU R128 + true peak, the headline shape) from the PMC passes of tools/prof_seg.sh: FETCH_SIZE (KiB; on gfx950 it reports half
the bytes of a wide coalesced stream: x 2 — MI355X_MICROARCH.md, HBM section) + WRITE_SIZE (KiB), together with the hash of the kernel
sources the passes were taken on (bench.py reports `roofline.traffic` only while the sources still hash to it)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src, dst = sys.argv[1], sys.argv[2]
vals = {}
for line in open(src):
    m = re.match(r"\('(.*?)', '(\w+)'\) avg/dispatch = ([0-9.e+]+)", line)
    if m and "k_seg" in m.group(1):
        vals[m.group(2)] = float(m.group(3))
fetch, write = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
S, T = 8192, 480000
out = {
    "_comment": "HBM traffic per launch of the dominant kernel from rocprofv3 PMC passes (tools/prof_seg.sh -> tools/make_traffic.py). FETCH_SIZE is in KiB "
                "and on gfx950 reports half the bytes of a wide stream (MI355X_MICROARCH.md, HBM section): bytes = FETCH_SIZE * 1024 * 2 + WRITE_SIZE * 1024. "
                "bench.py reports the figure only for this workload AND for kernel sources with this hash (sha256 of csrc/mtr_seg.hip + mtr_mfma16_fir.h + "
                "mtr_internal.h + Makefile, first 16 hex digits).  What lies over the algorithmic bytes is the K-filter warm-up: 7 of 8 segments read the "
                "0.075 s in front of their 1.25 s twice, in whole 16-frame steps.",
    "workload": {"meters": "ebu+tp", "streams_per_gpu": S, "frames_per_stream": T, "layout": 7},
    "kernel": "k_seg<true, true>",
    "kernel_sha16": bench.kernel_sha(),
    "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
    "traffic_bytes_per_launch": int(fetch * 1024 * 2 + write * 1024),
    "algorithmic_bytes_per_launch": S * T * 8,
    "source": os.path.relpath(src, ROOT),
}
json.dump(out, open(dst, "w"), indent=1)
print("wrote", dst, "traffic / algorithmic = %.4f" % (out["traffic_bytes_per_launch"] / out["algorithmic_bytes_per_launch"]))
