#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q --maxfail=30 > $O/gputests12.txt 2>&1; echo "pytest rc $?" >> $O/gputests12.txt; tail -4 $O/gputests12.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench12.json 2> $O/bench12.err
python - <<PY
import json
d=json.loads(open("$O/bench12.json").read().strip().splitlines()[-1])
r=d['roofline']
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_median')})
print({k:r[k] for k in ('frac','kernel_ms','kernel_ms_median','whole_step_frac','gate_ms')})
for k,v in d['extra']['configs'].items(): print(k, {a:b for a,b in v.items() if a in ('kernel_ms','frac','wall_ms')})
print(d['extra'].get('end_to_end_host'))
PY
for fs in 48000 44100; do
  MTR_LIB=$PWD/meters.lv2_amd/lib_prof/libmtr_engine.so timeout 300 python tools/seg_prof.py ebu+tp $fs 2>&1 | grep -v amdgpu
done > $O/seg_prof12.txt; cat $O/seg_prof12.txt
timeout 300 python tools/align_probe.py 2>&1 | grep -v amdgpu > $O/align12.txt; cat $O/align12.txt
