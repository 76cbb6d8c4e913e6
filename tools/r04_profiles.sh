#!/bin/bash
# tools/r04_profiles.sh — everything profiles/r04* is made from, in one GPU-box call (outputs under gpurun_out/r04p/).
o=gpurun_out/r04p; mkdir -p $o
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q --maxfail=30 > $o/gputests_final.txt 2>&1; echo "pytest rc $?" >> $o/gputests_final.txt
bash tools/prof_seg.sh r04a_seg_ebu_tp > /dev/null 2>&1; cp gpurun_out/prof_r04a_seg_ebu_tp/summary.txt $o/r04a_seg_ebu_tp.txt
bash tools/prof_seg.sh r04a44_seg_ebu_tp --fs 44100 > /dev/null 2>&1; cp gpurun_out/prof_r04a44_seg_ebu_tp/summary.txt $o/r04a44_seg_ebu_tp.txt
bash tools/prof_seg.sh r04_tpb --meters tpb --steps 4 > /dev/null 2>&1; cp gpurun_out/prof_r04_tpb/summary.txt $o/r04_tpb.txt
for v in "" _nofetch _nochain _noprod _nosplit _pw4 _pw4_unfused _pw2_unfused _pw2_unfused_fourmaps; do echo "=== tpb_prof$v"; timeout 120 ./tools/tpb_prof$v 8192 96000 2>&1 | grep -v amdgpu.ids; done > $o/r04_tpb_roles.txt 2>&1
for fs in 48000 44100; do MTR_LIB=$PWD/meters.lv2_amd/lib_prof/libmtr_engine.so timeout 300 python tools/seg_prof.py ebu+tp $fs 2>&1 | grep -v amdgpu; done > $o/r04_kseg_step_cycles.txt
MTR_LIB=$PWD/meters.lv2_amd/lib_prof/libmtr_engine.so timeout 300 python tools/seg_prof.py tp 48000 2>&1 | grep -v amdgpu >> $o/r04_kseg_step_cycles.txt
timeout 300 python tools/bank_mono_probe.py 2>&1 | grep -v amdgpu > $o/r04_bank_mono.txt
MTR_BENCH_SHARED_GPU=1 MTR_BENCH_TRY_RCCL=1 timeout 600 python bench.py --gpus 2 --streams 1024 --seconds 10 --steps 5 --warmup 2 --no-extra --no-cpu-baseline > $o/r04_two_ranks_one_gpu.json 2> $o/r04_two_ranks_one_gpu.err
timeout 900 python bench.py --steps 20 --warmup 5 > $o/r04_bench_line.json 2> $o/r04_bench.err
find gpurun_out/prof_r04* -name "*.csv" -size +2M -delete
ls -la $o; tail -c 600 $o/r04_two_ranks_one_gpu.json
