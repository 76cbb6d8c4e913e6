#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
(timeout 120 ./tools/mfma_power 1024 100000; timeout 120 ./tools/mfma_power 2048 50000) 2>&1 | grep -v amdgpu > $O/mfma_power16.txt; cat $O/mfma_power16.txt
