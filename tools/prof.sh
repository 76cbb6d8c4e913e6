#!/bin/bash
# tools/prof.sh <tag> [bench args...] — rocprofv3 kernel-trace stats + PMC passes for bench.py on the GPU box.
# Counters are collected in their own runs (no trace domains combined with --pmc).
tag=$1; shift
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
B="python $PWD/bench.py --no-cpu-baseline --steps 2 --warmup 1 $*"
cd /tmp
rocprofv3 --kernel-trace --stats -d $out/trace --output-format csv -- $B > $out/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $out/pmc1 --output-format csv -- $B > $out/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $out/pmc2 --output-format csv -- $B > $out/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $out/pmc3 --output-format csv -- $B > $out/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d $out/pmc4 --output-format csv -- $B > $out/pmc4.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC -d $out/pmc5 --output-format csv -- $B > $out/pmc5.log 2>&1
find $out -name "*.csv" | head -30
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$out/trace/**/*kernel_stats.csv", recursive=True)):
    print("==", f)
    for i, row in enumerate(csv.reader(open(f))):
        if i < 8: print(row)
for p in ("pmc1","pmc2","pmc3","pmc4","pmc5"):
    for f in sorted(glob.glob("$out/%s/**/*counter_collection.csv" % p, recursive=True)):
        acc = collections.defaultdict(lambda: [0.0, 0])
        rd = csv.DictReader(open(f))
        for row in rd:
            k = (row.get("Kernel_Name","?")[:40], row.get("Counter_Name"))
            acc[k][0] += float(row.get("Counter_Value", 0)); acc[k][1] += 1
        print("==", p)
        for k, (v, n) in sorted(acc.items()):
            if any(t in k[0] for t in ("fused", "bank", "gate", "k_kw", "bitstats", "sigdist", "k_tpb", "kwtp")):
                print(k, "avg/dispatch = %.4g" % (v / n), "n =", n)
PY
