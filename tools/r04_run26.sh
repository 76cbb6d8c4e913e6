#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 > $O/t26.txt 2>&1; echo "rc $?" >> $O/t26.txt
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" $O/t26.txt | tail -8
timeout 500 python tools/fuzz_more.py 9000 500 2>&1 | grep -v amdgpu | tail -5 | tee $O/fuzz26.txt
