#!/bin/bash
# tools/prof_others.sh — rocprofv3 kernel traces (no counters) of every meter set but the headline's, at the headline shape
# (8192 streams x 10 s; the bank also at BASELINE config 3's 4096): median kernel durations -> gpurun_out/r05/r05_other_kernels.txt
O=$PWD/gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
cd /tmp
: > $O/r05_other_kernels.txt
for spec in "ebu:" "tp:" "spectr30:" "spectr30:--streams 4096" "tpb:" "dr14:" "kmeter:" "bitstats:" "sigdist:" "ebu+tp+spectr30:"; do
	m=${spec%%:*}; extra=${spec#*:}
	d=$R/gpurun_out/prof_r05o_$(echo "$m$extra" | tr -c 'a-z0-9\n' '_')
	rm -rf $d; mkdir -p $d
	timeout 300 rocprofv3 --kernel-trace -d $d --output-format csv -- python $R/bench.py --no-cpu-baseline --no-extra --steps 4 --warmup 1 --meters $m $extra > $d/log.txt 2>&1
	python - "$d" "$m $extra" >> $O/r05_other_kernels.txt <<'PY'
import csv, glob, collections, statistics, sys
d, what = sys.argv[1], sys.argv[2]
print("== --meters %s: kernel, n, median ms, min ms (first dispatch cold)" % what.strip())
for f in sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)):
    t = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        t[row["Kernel_Name"][:70]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    for k, v in sorted(t.items(), key=lambda kv: -sum(kv[1])):
        if max(v) >= 0.02 and not k.startswith(("k_synth", "__amd", "void at::")):
            print("  %-70s %3d %9.4f %9.4f" % (k, len(v), statistics.median(v), min(v)))
PY
	find $d -name "*.csv" -size +1M -delete
done
cat $O/r05_other_kernels.txt
