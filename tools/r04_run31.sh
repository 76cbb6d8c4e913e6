#!/bin/bash
O=$PWD/gpurun_out/r04/pmc31; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
cd /tmp
for pad in 0 1; do
rocprofv3 --kernel-trace --stats -d $O/t$pad --output-format csv -- python $R/tools/stride_one.py $pad > $O/t$pad.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum -d $O/a$pad --output-format csv -- python $R/tools/stride_one.py $pad > $O/a$pad.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/b$pad --output-format csv -- python $R/tools/stride_one.py $pad > $O/b$pad.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INST_CYCLES_VMEM -d $O/c$pad --output-format csv -- python $R/tools/stride_one.py $pad > $O/c$pad.log 2>&1
done
python - <<PY
import csv, glob, collections
for pad in (0, 1):
    for p in "tabc":
        for f in sorted(glob.glob("$O/%s%d/**/*counter_collection.csv" % (p, pad), recursive=True)):
            acc = collections.defaultdict(lambda: [0.0, 0])
            for row in csv.DictReader(open(f)):
                if "k_seg" in row["Kernel_Name"]:
                    acc[row["Counter_Name"]][0] += float(row["Counter_Value"]); acc[row["Counter_Name"]][1] += 1
            for k, (v, n) in sorted(acc.items()): print("pad", pad, k, "%.5g" % (v / n))
        for f in sorted(glob.glob("$O/%s%d/**/*kernel_trace.csv" % (p, pad), recursive=True)):
            d = [ (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(f)) if "k_seg" in r["Kernel_Name"]]
            print("pad", pad, "k_seg ms", sorted(d))
PY
find $O -name "*.csv" -size +1M -delete
