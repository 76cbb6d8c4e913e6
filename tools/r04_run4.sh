#!/bin/bash
# round 4, GPU call 4: k_tpb with the fetch's latency off the critical path, per-role loops, pair maps
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
for v in "" _fourmaps _s1 _s2 _s3 _s4 _s5 _nochain _noprod _nomaps _nofetch _chainonly _prodonly _mapsonly; do
  echo "=== tpb_prof$v"; timeout 120 ./tools/tpb_prof$v 8192 96000 2>&1 | grep -v amdgpu.ids
done > $O/tpb_variants4.txt 2>&1
cat $O/tpb_variants4.txt
timeout 900 python -m pytest tests/test_gpu_hostpath.py tests/test_gpu_parity.py tests/test_lv2_plugin.py tests/test_lv2_dr14.py -m gpu -q -k "host or ballistics or dBTP or dr14 or TPnRMS" > $O/gputests4.txt 2>&1; tail -5 $O/gputests4.txt
timeout 900 python tools/fuzz_tpb.py 0 600 > $O/fuzz_tpb4.txt 2>&1; tail -3 $O/fuzz_tpb4.txt
timeout 300 bash tools/tpb_ab.sh lib > $O/tpb_ab4.txt 2>&1; grep k_tpb $O/tpb_ab4.txt
