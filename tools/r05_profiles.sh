#!/bin/bash
# tools/r05_profiles.sh — everything profiles/r05* is made from, in ONE GPU-box call (outputs under gpurun_out/r05/; what is judged is
# copied to profiles/ by hand afterwards).  Needs tools/build_tpb_prof.sh run in the build container first (the binaries travel).
O=gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids"
# 1. the whole -m gpu suite, smoke (), every fuzzer on fresh seeds
timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 > $O/gputests.txt 2>&1; echo "pytest rc $?" >> $O/gputests.txt; grep -v "$F" $O/gputests.txt | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v "$F" | tail -2 | tee $O/smoke.txt
( timeout 400 python tools/fuzz_more.py 15000 400; timeout 400 python tools/fuzz_unaligned.py 3000 400; timeout 500 python tools/fuzz_tpb.py 11000 800; timeout 300 python tools/fuzz_intstat.py 7000 400 ) 2>&1 | grep -v "$F" | tee $O/fuzz.txt | tail -8
# 2. rocprofv3: kernel trace + PMC passes (counters in their own runs) for the headline kernel at 48 and 44.1 kHz and for k_tpb
bash tools/prof_seg.sh r05a_seg_ebu_tp > /dev/null 2>&1; cp gpurun_out/prof_r05a_seg_ebu_tp/summary.txt $O/r05a_seg_ebu_tp.txt
bash tools/prof_seg.sh r05a44_seg_ebu_tp --fs 44100 > /dev/null 2>&1; cp gpurun_out/prof_r05a44_seg_ebu_tp/summary.txt $O/r05a44_seg_ebu_tp.txt
bash tools/prof_seg.sh r05_tpb --meters tpb --steps 4 > /dev/null 2>&1; cp gpurun_out/prof_r05_tpb/summary.txt $O/r05_tpb.txt
python tools/make_traffic.py $O/r05a_seg_ebu_tp.txt $O/r05_traffic.json
# 3. k_tpb: cycles per role and wave — the shipped form, round 4's, and the eliminations (timing-only builds)
for v in "" _r4 _nochain _noprod _nosplit _nodma; do [ -x tools/tpb_prof$v ] || continue; echo "=== tpb_prof$v"; timeout 120 ./tools/tpb_prof$v 8192 96000 2>&1 | grep -v amdgpu.ids; done > $O/r05_tpb_roles.txt 2>&1
[ -f meters.lv2_amd/lib_r4/libmtr_engine.so ] && bash tools/tpb_ab.sh lib lib_r4 > $O/r05_tpb_ab.txt 2>&1
# 4. the N = 2 rehearsal on one GPU, and the sleeping rank
MTR_BENCH_SHARED_GPU=1 MTR_BENCH_TRY_RCCL=1 timeout 600 python bench.py --gpus 2 --streams 1024 --seconds 10 --steps 5 --warmup 2 --no-extra --no-cpu-baseline > $O/r05_two_ranks_one_gpu.json 2> $O/r05_two_ranks_one_gpu.err
MTR_BENCH_SHARED_GPU=1 MTR_BENCH_TRY_RCCL=1 MTR_BENCH_COMM_TIMEOUT_S=6 MTR_BENCH_FAULT=sleep_in_init:1:25 timeout 600 python bench.py --gpus 2 --streams 64 --seconds 3 --steps 2 --warmup 1 --no-extra --no-cpu-baseline > $O/r05_sleeping_rank.json 2> $O/r05_sleeping_rank.err
# 5. the line (with the traffic figure just taken: bench.py reads profiles/r05_traffic.json)
cp $O/r05_traffic.json profiles/r05_traffic.json
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r05_bench_line.json 2> $O/r05_bench.err; cut -c1-300 $O/r05_bench_line.json
find gpurun_out/prof_r05* -name "*.csv" -size +2M -delete
python - <<PY
import json; d = json.load(open("$O/r05_bench_line.json")); r = d["roofline"]
print({k: r[k] for k in ("frac", "traffic", "kernel_ms", "kernel_ms_median", "whole_step_frac")}, d["cpu_baseline"]["value"])
for k, v in d["extra"]["configs"].items(): print("  %-90s %s ms  %.3f" % (k[:90], v["kernel_ms"], v["frac"]))
print(d["extra"].get("lv2_run_latency", {}).get("dBTPstereo"))
PY
