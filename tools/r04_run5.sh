#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
for v in "" _nofetch _b3 _b4 _b5 _b6 _b7 _p2; do
  echo "=== tpb_prof$v"; timeout 120 ./tools/tpb_prof$v 8192 96000 2>&1 | grep -v amdgpu.ids
done > $O/tpb_variants5.txt 2>&1
cat $O/tpb_variants5.txt
timeout 900 python -m pytest tests/test_gpu_hostpath.py tests/test_gpu_parity.py tests/test_lv2_plugin.py tests/test_lv2_dr14.py -m gpu -q -k "host or ballistics or dBTP or dr14 or TPnRMS" > $O/gputests5.txt 2>&1; tail -5 $O/gputests5.txt
timeout 900 python tools/fuzz_tpb.py 0 300 > $O/fuzz_tpb5.txt 2>&1; tail -3 $O/fuzz_tpb5.txt
timeout 300 bash tools/tpb_ab.sh lib > $O/tpb_ab5.txt 2>&1; grep k_tpb $O/tpb_ab5.txt
