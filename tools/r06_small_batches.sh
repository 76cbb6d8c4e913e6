#!/bin/bash
# tools/r06_small_batches.sh — does k_seg beat k_kwtp16 on mid-size batches the planner gives to k_kwtp16?  
for S in 512 1024 2048 4096; do for g in 0 16 25 50; do
	echo -n "streams=$S segments=$g : "
	python bench.py --no-extra --no-cpu-baseline --steps 10 --warmup 3 --streams $S --segments $g 2>/dev/null |
		python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('%s k_ms %.4f step_ms %.4f frac %.4f' % (r['kernel'], r['kernel_ms'], d['ms_per_step'], r['frac']))"
done; done
