#!/bin/bash
# tools/tpb_ab.sh [lib dirs] — same-box comparison of k_tpb builds (TruePeakdsp::process for a batch): 8192 and 16384 streams x 10 s
[ $# -eq 0 ] && set -- lib lib_ab
for i in 1 2; do for L in "$@"; do for S in 8192 16384; do
	echo -n "$L streams=$S : "
	MTR_LIB=$PWD/meters.lv2_amd/$L/libmtr_engine.so python bench.py --no-extra --no-cpu-baseline --meters tpb --streams $S --steps 3 --warmup 1 2>/dev/null |
		python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('k_tpb %.2f ms  frac %.4f' % (d['roofline']['kernel_ms'], d['roofline']['frac']))"
done; done; done
