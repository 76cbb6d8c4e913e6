#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_seg.py tests/test_gpu_hostpath.py -m gpu -q -x > $O/t23.txt 2>&1; echo "rc $?" >> $O/t23.txt
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" $O/t23.txt | tail -8
for i in 1 2; do bash tools/seg_ab.sh lib lib_ab; done 2>&1 | tee $O/ab23.txt
