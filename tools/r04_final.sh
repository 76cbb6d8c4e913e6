#!/bin/bash
# the round's last GPU call: the whole -m gpu suite, smoke(), every fuzzer on fresh seeds, the bench line (with the committed traffic figure)
O=gpurun_out/r04f; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 > $O/gputests.txt 2>&1; echo "pytest rc $?" >> $O/gputests.txt
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" $O/gputests.txt | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu | tail -2 | tee $O/smoke.txt
( timeout 500 python tools/fuzz_more.py 12000 600; timeout 500 python tools/fuzz_unaligned.py 2000 600; timeout 400 python tools/fuzz_tpb.py 4000 800; timeout 300 python tools/fuzz_intstat.py 5000 600 ) 2>&1 | grep -v amdgpu | tee $O/fuzz.txt | tail -8
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; cut -c1-400 $O/bench_line.json
python - <<PY
import json; d = json.load(open("$O/bench_line.json")); r = d["roofline"]
print({k: r[k] for k in ("frac", "traffic", "kernel_ms", "kernel_ms_median", "whole_step_frac")}, d["cpu_baseline"]["value"], list(d.get("extra", {}).keys()))
for k, v in d["extra"]["configs"].items(): print("  %-90s %8.3f ms  %.3f" % (k[:90], v["kernel_ms"], v["frac"]))
print(d["extra"].get("end_to_end_host"))
PY
