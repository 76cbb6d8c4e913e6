#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
for v in nosplit noput nodma dmaonly putonly splitonly; do echo "=== $v"; timeout 60 ./tools/tpb_prof_$v 8192 96000 2>&1 | grep -v amdgpu.ids | sed -n '1p;4p;7p;8p;11p'; done > $O/tpb_w3_elim.txt 2>&1
cat $O/tpb_w3_elim.txt
