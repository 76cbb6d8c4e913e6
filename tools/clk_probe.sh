#!/bin/bash
# tools/clk_probe.sh — shader clock and duration of the dominant kernel for each library build and meter set:
# GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / duration.
export TMPDIR=/tmp
top=$PWD
for L in "$@"; do
for m in ebu+tp tp; do
	out=$top/gpurun_out/clk_${L}_$m
	rm -rf $out; mkdir -p $out
	(cd /tmp && MTR_ALLOW_TIMING_ONLY_BUILD=1 MTR_LIB=$top/meters.lv2_amd/$L/libmtr_engine.so rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES SQ_INSTS_VALU --kernel-trace -d $out --output-format csv -- python $top/bench.py --no-cpu-baseline --no-extra --steps 4 --warmup 1 --meters $m > $out/log 2>&1)
	python - <<PY
import csv, glob, collections, statistics
dur = collections.defaultdict(list)
for f in glob.glob("$out/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"][:40]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in dur.items():
    if "k_seg" in k or "k_kwtp" in k:
        g = statistics.median(cnt[k]["GRBM_GUI_ACTIVE"]) / 8 if cnt[k]["GRBM_GUI_ACTIVE"] else 0
        d = statistics.median(v)
        print("$L $m %-28s n=%d median %.3f ms  cycles/XCD %.4g  clock %.3f GHz  wait_any/wave_total %.3g  valu %.4g" % (
            k[:28], len(v), d, g, g / d / 1e6, statistics.median(cnt[k]["SQ_WAIT_INST_ANY"] or [0]), statistics.median(cnt[k]["SQ_INSTS_VALU"] or [0])))
PY
done
done
