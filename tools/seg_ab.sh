#!/bin/bash
# tools/seg_ab.sh — elimination runs of k_seg (layout 7) on the GPU box: the shipped build against builds without the
# stream (every step re-reads one line) and without the products, EBU R128 + true peak and true peak alone.
# Build the variants first (CPU container):
#   for v in NOADV NOPROD; do make -s -C meters.lv2_amd/csrc OUT=../lib_$v EXTRA_mtr_seg="$(make -s -C meters.lv2_amd/csrc print-seg-flags) -DMTR_SEG_DBG_$v" ../lib_$v/libmtr_engine.so; done
for L in lib lib_NOADV lib_NOPROD; do
	[ -f meters.lv2_amd/$L/libmtr_engine.so ] || continue
	echo "== $L"
	MTR_LIB=$PWD/meters.lv2_amd/$L/libmtr_engine.so python tools/seg_try.py big 2>&1 | grep -v amdgpu.ids | grep "layout 7"
done
