#!/bin/bash
# tools/r06_tail_ab.sh — same-box A/B of the serial and the deferred tail (bench.py --tail 1 / --tail 0), alternating:
# the dominant kernel, k_gate (HIP events), the whole step as the wall clock sees it, whole_step_frac.
for i in 1 2 3; do
	for t in 1 0; do
		echo -n "tail=$t : "
		python bench.py --no-extra --no-cpu-baseline --steps 20 --warmup 5 --tail $t "$@" 2>/dev/null |
			python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('k_ms %.4f gate_ms %.4f step_ms %.4f frac %.4f whole %.4f median_step %.4f deferred %s' % (r['kernel_ms'], r['gate_ms'], d['ms_per_step'], r['frac'], r['whole_step_frac'], d['ms_per_step_median'], d['config']['deferred_calls']))"
	done
done
