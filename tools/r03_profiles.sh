#!/bin/bash
# tools/r03_profiles.sh — everything profiles/r03* is made from, in one GPU-box call (outputs under gpurun_out/r03/).
o=gpurun_out/r03; mkdir -p $o
bash tools/prof_seg.sh r03a_seg_ebu_tp > /dev/null 2>&1; cp gpurun_out/prof_r03a_seg_ebu_tp/summary.txt $o/r03a_seg_ebu_tp.txt
bash tools/prof_seg.sh r03b_seg_tp --meters tp > /dev/null 2>&1; cp gpurun_out/prof_r03b_seg_tp/summary.txt $o/r03b_seg_tp.txt
bash tools/prof_seg.sh r03c_bank --meters spectr30 --steps 3 > /dev/null 2>&1; cp gpurun_out/prof_r03c_bank/summary.txt $o/r03c_bank.txt
bash tools/prof_seg.sh r03d_kw --meters ebu > /dev/null 2>&1; cp gpurun_out/prof_r03d_kw/summary.txt $o/r03d_kw_ebu_only.txt
bash tools/prof_seg.sh r03e_kwtp16 --layout 6 > /dev/null 2>&1; cp gpurun_out/prof_r03e_kwtp16/summary.txt $o/r03e_kwtp16_layout6.txt
bash tools/clk_probe.sh lib lib_NOADV lib_NOPROD 2>&1 | grep -v amdgpu.ids > $o/r03_kseg_clock.txt
for m in ebu+tp tp; do MTR_LIB=$PWD/meters.lv2_amd/lib_prof/libmtr_engine.so python tools/seg_prof.py $m 2>&1 | grep -v amdgpu; done > $o/r03_kseg_step_cycles.txt
bash tools/seg_ab.sh lib lib_NOADV lib_NOPROD > $o/r03_kseg_elimination.txt 2>&1
python bench.py > $o/r03_bench_line.json 2> $o/r03_bench.err
ls -la $o
