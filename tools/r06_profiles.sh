#!/bin/bash
# tools/r06_profiles.sh — everything profiles/r06* is made from, in ONE GPU-box call (outputs under gpurun_out/r06/; what is judged is
# copied to profiles/ afterwards).
O=gpurun_out/r06; mkdir -p $O; export TMPDIR=/tmp
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids"
# 1. the whole -m gpu suite (with the prints of the full-size tests: the histogram flip rates), smoke (), every fuzzer on fresh seeds
timeout 1800 python -m pytest tests -m gpu -q -s --maxfail=30 > $O/gputests.txt 2>&1; echo "pytest rc $?" >> $O/gputests.txt; grep -v "$F" $O/gputests.txt | grep "passed\|failed\|histogram flips\|1 stream x 3600\|pytest rc"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v "$F" | tail -2 | tee $O/smoke.txt
( timeout 400 python tools/fuzz_more.py 25000 400; timeout 400 python tools/fuzz_unaligned.py 5000 400; timeout 500 python tools/fuzz_tpb.py 31000 800; timeout 300 python tools/fuzz_intstat.py 9000 400 ) 2>&1 | grep -v "$F" | tee $O/fuzz.txt | tail -8
# 2. rocprofv3: kernel trace + PMC passes (counters in their own runs) for the headline kernel at 48 and 44.1 kHz and for k_tpb
bash tools/prof_seg.sh r06a_seg_ebu_tp > /dev/null 2>&1; cp gpurun_out/prof_r06a_seg_ebu_tp/summary.txt $O/r06a_seg_ebu_tp.txt
bash tools/prof_seg.sh r06a44_seg_ebu_tp --fs 44100 > /dev/null 2>&1; cp gpurun_out/prof_r06a44_seg_ebu_tp/summary.txt $O/r06a44_seg_ebu_tp.txt
bash tools/prof_seg.sh r06_tpb --meters tpb --steps 4 > /dev/null 2>&1; cp gpurun_out/prof_r06_tpb/summary.txt $O/r06_tpb.txt
python tools/make_traffic.py $O/r06a_seg_ebu_tp.txt $O/r06_traffic.json
# 3. the deferred tail: same-box A/B against the serial order, and the kernel trace of a deferred run (who overlaps whom)
bash tools/r06_tail_ab.sh 2>&1 | grep -v "$F" > $O/r06_tail_ab.txt; cat $O/r06_tail_ab.txt
STEPS=3 bash tools/r06_trace.sh r06_deferred python $PWD/tools/r06_tail_probe.py deferred 2>&1 | grep -v "$F" | tail -24 > $O/r06_trace_deferred.txt
# 4. the N = 2 rehearsal on one GPU, and the sleeping rank
MTR_BENCH_SHARED_GPU=1 MTR_BENCH_TRY_RCCL=1 timeout 600 python bench.py --gpus 2 --streams 1024 --seconds 10 --steps 5 --warmup 2 --no-extra --no-cpu-baseline > $O/r06_two_ranks_one_gpu.json 2> $O/r06_two_ranks_one_gpu.err
MTR_BENCH_SHARED_GPU=1 MTR_BENCH_TRY_RCCL=1 MTR_BENCH_COMM_TIMEOUT_S=6 MTR_BENCH_FAULT=sleep_in_init:1:25 timeout 600 python bench.py --gpus 2 --streams 64 --seconds 3 --steps 2 --warmup 1 --no-extra --no-cpu-baseline > $O/r06_sleeping_rank.json 2> $O/r06_sleeping_rank.err
# 5. the line (with the traffic figure just taken: bench.py reads profiles/r06_traffic.json)
cp $O/r06_traffic.json profiles/r06_traffic.json
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r06_bench_line.json 2> $O/r06_bench.err; cut -c1-300 $O/r06_bench_line.json
find gpurun_out/prof_r06* gpurun_out/trace_r06* -name "*.csv" -size +2M -delete
python - <<PY
import json; d = json.load(open("$O/r06_bench_line.json")); r = d["roofline"]
print({k: r[k] for k in ("frac", "traffic", "kernel_ms", "kernel_ms_median", "whole_step_frac", "gate_ms")}, d["ms_per_step"], d["config"]["tail"][:20], d["config"]["rccl_nranks"], d["config"]["rank_devices"])
print({k: v for k, v in d["cpu_baseline"].items() if k != "sample"})
for k, v in d["extra"]["configs"].items(): print("  %-90s %s ms  %.3f" % (k[:90], v["kernel_ms"], v["frac"]))
print(d["extra"].get("lv2_run_latency", {}).get("dBTPstereo"))
PY
