#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
for v in "" _s1247 _s1257 _s1267 _s1524 _s1246; do echo "=== tpb_prof$v"; timeout 120 ./tools/tpb_prof$v 8192 96000 2>&1 | grep -v "amdgpu.ids\|shader cycles\| wave "; done > $O/tpb_sets14.txt 2>&1
cat $O/tpb_sets14.txt
