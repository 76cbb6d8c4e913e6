#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_bench_ranks.py tests/test_gpu_reduce.py -m gpu -q 2>&1 | tail -3
