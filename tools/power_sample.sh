#!/bin/bash
# tools/power_sample.sh [bench args...] — socket power and shader clock while bench.py repeats its step (rocm-smi, 0.25 s grid)
rocm-smi --showmaxpower 2>/dev/null | grep -i "max\|cap" | head -3
python bench.py --no-extra --no-cpu-baseline --steps 600 --warmup 5 "$@" > /tmp/ps_bench.json 2>/dev/null &
pid=$!
sleep 6      # import torch + synthetic fill
for i in $(seq 1 16); do
	rocm-smi --showpower --showclocks 2>/dev/null | grep -iE "Socket Graphics Package Power|sclk" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ' '; echo
	sleep 0.25
done
wait $pid
python -c "import json; d=json.loads(open('/tmp/ps_bench.json').readline()); print('kernel_ms', d['roofline']['kernel_ms'], 'step_ms', d['ms_per_step'])"
