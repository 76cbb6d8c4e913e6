#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q --maxfail=30 > $O/gputests19.txt 2>&1; echo "pytest rc $?" >> $O/gputests19.txt; tail -4 $O/gputests19.txt
timeout 900 python tools/fuzz_tpb.py 0 600 2>&1 | tail -2
timeout 300 bash tools/tpb_ab.sh lib 2>&1 | grep k_tpb
timeout 300 python bench.py --no-extra --no-cpu-baseline --meters tpb --streams 8259 --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('8259 streams', d['roofline']['kernel_ms'])"
