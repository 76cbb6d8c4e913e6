#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "fetch_paths or ballistics" > $O/t22.txt 2>&1; echo "rc $?" >> $O/t22.txt
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" $O/t22.txt | tail -25
