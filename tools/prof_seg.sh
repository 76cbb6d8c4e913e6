#!/bin/bash
# tools/prof_seg.sh <tag> [bench args] — rocprofv3 kernel-trace stats (>= 5 dispatches per kernel) + PMC passes for the
# K-weighting + true-peak step, counters in their own runs.  Summary: gpurun_out/prof_<tag>/summary.txt
tag=$1; shift
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
B="python $PWD/bench.py --no-cpu-baseline --no-extra --steps 6 --warmup 1 $*"
# the kernel trace on the driver's own flags (20 timed steps behind 5 of warm-up: 25 dispatches, so that the one cold first dispatch — + 2 ms: fresh pages —
# does not carry a seventh of the average the judge divides by); a later --steps in "$@" wins
BT="python $PWD/bench.py --no-cpu-baseline --no-extra --steps 20 --warmup 5 $*"
cd /tmp
rocprofv3 --kernel-trace --stats -d $out/trace --output-format csv -- $BT > $out/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d $out/pmc1 --output-format csv -- $B > $out/pmc1.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM -d $out/pmc2 --output-format csv -- $B > $out/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $out/pmc3 --output-format csv -- $B > $out/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d $out/pmc4 --output-format csv -- $B > $out/pmc4.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $out/pmc5 --output-format csv -- $B > $out/pmc5.log 2>&1
rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum -d $out/pmc6 --output-format csv -- $B > $out/pmc6.log 2>&1
python - > $out/summary.txt <<PY
import csv, glob, collections, statistics
for f in sorted(glob.glob("$out/trace/**/*kernel_trace.csv", recursive=True)):
    d = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        d[row["Kernel_Name"][:60]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    print("== kernel durations (ms): name, n, median, min, max        (average)")
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        print("%-60s %3d %9.4f %9.4f %9.4f   (avg %.4f)" % (k, len(v), statistics.median(v), min(v), max(v), sum(v) / len(v)))
for p in ("pmc1","pmc2","pmc3","pmc4","pmc5","pmc6"):
    for f in sorted(glob.glob("$out/%s/**/*counter_collection.csv" % p, recursive=True)):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            k = (row.get("Kernel_Name","?")[:40], row.get("Counter_Name"))
            acc[k][0] += float(row.get("Counter_Value", 0)); acc[k][1] += 1
        print("==", p)
        for k, (v, n) in sorted(acc.items()):
            if any(t in k[0] for t in ("k_seg", "k_kwtp", "k_kw", "k_gate", "k_bank", "k_fused", "k_tpb")):
                print(k, "avg/dispatch = %.5g" % (v / n), "n =", n)
PY
cat $out/summary.txt
