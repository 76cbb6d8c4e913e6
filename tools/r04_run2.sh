#!/bin/bash
# round 4, GPU call 2: k_tpb v3 (split once, f16 ring), unaligned k_seg without the second step form, chunked host path
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q --maxfail=30 > $O/gputests2.txt 2>&1; echo "pytest rc $?" >> $O/gputests2.txt
tail -30 $O/gputests2.txt
timeout 300 ./tools/tpb_prof 8192 96000 > $O/tpb_prof2.txt 2>&1; cat $O/tpb_prof2.txt
timeout 600 bash tools/tpb_ab.sh lib lib_ab > $O/tpb_ab2.txt 2>&1; cat $O/tpb_ab2.txt
timeout 900 python tools/fuzz_tpb.py 0 400 > $O/fuzz_tpb2.txt 2>&1; tail -5 $O/fuzz_tpb2.txt
timeout 600 bash tools/ab.sh --fs 44100 --steps 10 > $O/seg44_ab2.txt 2>&1; cat $O/seg44_ab2.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench2.json 2> $O/bench2.err; tail -c 1500 $O/bench2.json
