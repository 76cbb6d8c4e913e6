#!/bin/bash
# tools/r06_trace.sh <tag> <cmd...> — rocprofv3 kernel trace of a command; prints every dispatch: queue, start (ms from the first), duration
tag=$1; shift
out=$PWD/gpurun_out/trace_$tag
mkdir -p $out
export TMPDIR=/tmp
cmd="$*"
( cd /tmp && rocprofv3 --kernel-trace --stats -d $out --output-format csv -- $cmd > $out/log.txt 2>&1 )
python - <<PY
import csv, glob
for f in sorted(glob.glob("$out/**/*kernel_trace.csv", recursive=True)):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t0 = int(rows[0]["Start_Timestamp"])
    keep = rows[-60:]
    for r in keep:
        print("q%-3s %-44s start %10.3f ms  dur %8.3f ms  grid %s wg %s" % (r.get("Queue_Id", "?"), r["Kernel_Name"][:44], (int(r["Start_Timestamp"]) - t0) / 1e6,
              (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))))
PY
