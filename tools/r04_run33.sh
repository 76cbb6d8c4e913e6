#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout 900 python tools/span_probe_tpb.py 2>&1 | grep -v amdgpu | tee $O/span33.txt
