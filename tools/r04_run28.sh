#!/bin/bash
# k_seg's step head: A/B against HEAD (lib_ab) on the bench shape, step cycles, quick parity
O=gpurun_out/r04; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_seg.py -m gpu -q -x 2>&1 | tail -2
for i in 1 2; do bash tools/seg_ab.sh lib lib_ab; done 2>&1 | grep "layout 7\|==" | tee $O/ab28.txt
for fs in 48000 44100; do MTR_LIB=$PWD/meters.lv2_amd/lib_prof/libmtr_engine.so timeout 300 python tools/seg_prof.py ebu+tp $fs 2>&1 | grep -v amdgpu; done | tee $O/seg_prof28.txt
