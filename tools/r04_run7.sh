#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
for v in "" _nofetch _unfused_b9 _nochain _noprod _nosplit; do
  echo "=== tpb_prof$v"; timeout 120 ./tools/tpb_prof$v 8192 96000 2>&1 | grep -v amdgpu.ids
done > $O/tpb_variants7.txt 2>&1
cat $O/tpb_variants7.txt
timeout 1800 python -m pytest tests -m gpu -q --maxfail=30 > $O/gputests7.txt 2>&1; echo "pytest rc $?" >> $O/gputests7.txt; tail -12 $O/gputests7.txt
timeout 900 python tools/fuzz_tpb.py 0 600 > $O/fuzz_tpb7.txt 2>&1; tail -3 $O/fuzz_tpb7.txt
timeout 300 bash tools/tpb_ab.sh lib > $O/tpb_ab7.txt 2>&1; grep k_tpb $O/tpb_ab7.txt
