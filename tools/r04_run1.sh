#!/bin/bash
# round 4, GPU call 1: the new tests + the whole -m gpu suite, the driver-shaped bench line, the alignment probe,
# step cycles at 48 / 44.1 kHz, FETCH_SIZE / L2 counters at both rates
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q --maxfail=20 > $O/gputests1.txt 2>&1; echo "pytest rc $?" >> $O/gputests1.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench1.json 2> $O/bench1.err
timeout 600 python tools/align_probe.py > $O/align.txt 2>&1
for fs in 48000 44100; do
  MTR_LIB=$PWD/meters.lv2_amd/lib_prof/libmtr_engine.so timeout 300 python tools/seg_prof.py ebu+tp $fs >> $O/seg_prof.txt 2>&1
done
top=$PWD
for fs in 48000 44100; do
  (cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE -d $top/$O/pmc_fetch_$fs --output-format csv -- python $top/bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 --fs $fs > $top/$O/pmc_fetch_$fs.log 2>&1)
  (cd /tmp && timeout 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_PENDING_STALL_CYCLES_sum -d $top/$O/pmc_l2_$fs --output-format csv -- python $top/bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 --fs $fs > $top/$O/pmc_l2_$fs.log 2>&1)
done
python - > $O/pmc_summary.txt <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/pmc_*/")):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = (row.get("Kernel_Name","?")[:30], row.get("Counter_Name"))
            acc[k][0] += float(row.get("Counter_Value", 0)); acc[k][1] += 1
    print("==", d)
    for k, (v, n) in sorted(acc.items()):
        if "k_seg" in k[0] or "k_kwtp" in k[0]:
            print(k, "avg/dispatch = %.6g" % (v / n), "n =", n)
PY
find $O -name "*.csv" -size +1M -delete
tail -5 $O/gputests1.txt; cat $O/align.txt; cat $O/seg_prof.txt; cat $O/pmc_summary.txt
