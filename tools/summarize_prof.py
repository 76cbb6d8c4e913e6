#!/usr/bin/env python3
"""tools/summarize_prof.py <gpurun_out/prof_TAG> <profiles/NAME.md> — condense the rocprofv3 output of
tools/prof.sh (kernel-trace stats + separate --pmc passes) into one small tracked markdown file."""
import collections
import csv
import glob
import sys

src, dst = sys.argv[1], sys.argv[2]
out = ["# rocprofv3 summary: %s" % src, "",
       "Command: `python bench.py --no-cpu-baseline --steps 2 --warmup 1 %s` under `rocprofv3 --kernel-trace --stats`" % " ".join (sys.argv[3:]),
       "and, in separate runs, `rocprofv3 --pmc ...` (tools/prof.sh). 3 dispatches per kernel.", "",
       "## kernel stats (rocprofv3 --kernel-trace --stats)", "",
       "| kernel | calls | avg ms | min ms | max ms | % |", "|---|---|---|---|---|---|"]
for f in sorted(glob.glob(src + "/trace/**/*kernel_stats.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        out.append("| `%s` | %s | %.3f | %.3f | %.3f | %s |" % (
            row["Name"][:70], row["Calls"], float(row["AverageNs"]) / 1e6, float(row["MinNs"]) / 1e6,
            float(row["MaxNs"]) / 1e6, row["Percentage"]))
out += ["", "## PMC counters, average per dispatch (summed over XCDs/SEs as rocprofv3 reports them)", "",
        "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are in quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE (x 8 XCDs) in cycles. FETCH_SIZE / WRITE_SIZE are in KiB;",
        "on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced stream (MI355X_MICROARCH.md §HBM): x2.", ""]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob(src + "/pmc*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = (row.get("Kernel_Name", "?")[:48], row.get("Counter_Name"))
        acc[k][0] += float(row.get("Counter_Value", 0))
        acc[k][1] += 1
kernels = sorted({k[0] for k in acc})
for kn in kernels:
    if not any(t in kn for t in ("k_seg", "fused", "gate", "bank", "aggregate", "k_kw", "kwtp", "bitstats", "sigdist", "k_tpb")):
        continue
    out += ["### `%s`" % kn, "", "| counter | avg / dispatch |", "|---|---|"]
    for (k, c), (v, n) in sorted(acc.items()):
        if k == kn:
            out.append("| %s | %.5g |" % (c, v / n))
    out.append("")
open(dst, "w").write("\n".join(out) + "\n")
print("wrote", dst)
