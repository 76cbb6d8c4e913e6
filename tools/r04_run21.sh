#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bench_ranks.py tests/test_gpu_reduce.py -m gpu -q > $O/ranks21.txt 2>&1; echo "rc $?" >> $O/ranks21.txt
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" $O/ranks21.txt | tail -8
