#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
for L in lib lib_ab; do echo "== $L"; MTR_LIB=$PWD/meters.lv2_amd/$L/libmtr_engine.so timeout 600 python tools/span_probe.py 2>&1 | grep -v amdgpu; done | tee $O/span32.txt
