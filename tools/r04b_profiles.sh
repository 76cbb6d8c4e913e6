#!/bin/bash
# tools/r04b_profiles.sh — the k_seg profiles re-taken after the 44.1 kHz change (the call's last fragment in k_seg): trace + PMC passes at
# 48 and 44.1 kHz, the step's cycle breakdown, the bench line.  Outputs under gpurun_out/r04p/.
o=gpurun_out/r04p; mkdir -p $o
export TMPDIR=/tmp
bash tools/prof_seg.sh r04b_seg_ebu_tp > /dev/null 2>&1; cp gpurun_out/prof_r04b_seg_ebu_tp/summary.txt $o/r04b_seg_ebu_tp.txt
bash tools/prof_seg.sh r04b44_seg_ebu_tp --fs 44100 > /dev/null 2>&1; cp gpurun_out/prof_r04b44_seg_ebu_tp/summary.txt $o/r04b44_seg_ebu_tp.txt
for fs in 48000 44100; do MTR_LIB=$PWD/meters.lv2_amd/lib_prof/libmtr_engine.so timeout 300 python tools/seg_prof.py ebu+tp $fs 2>&1 | grep -v amdgpu; done > $o/r04b_kseg_step_cycles.txt
MTR_LIB=$PWD/meters.lv2_amd/lib_prof/libmtr_engine.so timeout 300 python tools/seg_prof.py tp 48000 2>&1 | grep -v amdgpu >> $o/r04b_kseg_step_cycles.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $o/r04b_bench_line.json 2> $o/r04b_bench.err
find gpurun_out/prof_r04b* -name "*.csv" -size +2M -delete
head -8 $o/r04b_seg_ebu_tp.txt; head -8 $o/r04b44_seg_ebu_tp.txt; cat $o/r04b_kseg_step_cycles.txt; cut -c1-600 $o/r04b_bench_line.json
