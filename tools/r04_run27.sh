#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout 800 python tools/fuzz_unaligned.py ${1:-0} ${2:-400} 2>&1 | grep -v amdgpu | tail -12 | tee $O/fuzz27.txt
