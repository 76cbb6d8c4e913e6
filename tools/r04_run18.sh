#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
for v in "" _prio3 _prio1 _prio3b; do echo "=== tpb_prof$v"; timeout 120 ./tools/tpb_prof$v 8192 96000 2>&1 | grep -v "amdgpu.ids\|shader cycles\| wave "; done > $O/tpb_prio18.txt 2>&1
cat $O/tpb_prio18.txt
