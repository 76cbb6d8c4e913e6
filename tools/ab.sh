#!/bin/bash
# tools/ab.sh [bench args...] — same-box A/B of two engine builds (box-to-box spread is +-3 %, an A/B on one box resolves 0.5 %):
# meters.lv2_amd/lib (A) against meters.lv2_amd/lib_ab (B, e.g. `make -C meters.lv2_amd/csrc OUT=../lib_ab EXTRA_mtr_fused4="... -DX" ../lib_ab/libmtr_engine.so`),
# alternating, three rounds; prints the dominant kernel, k_gate (HIP events) and the whole step in milliseconds.
for i in 1 2 3; do
	for L in lib lib_ab; do
		echo -n "$L $* : "
		MTR_LIB=$PWD/meters.lv2_amd/$L/libmtr_engine.so python bench.py --no-extra --no-cpu-baseline "$@" 2>/dev/null |
			python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['roofline']['kernel_ms'], d['roofline'].get('gate_ms'), d['ms_per_step'])"
	done
done
