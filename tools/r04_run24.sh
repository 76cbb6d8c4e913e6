#!/bin/bash
# fresh seeds of the three shape fuzzers on the round's final kernels
O=gpurun_out/r04; mkdir -p $O
( timeout 500 python tools/fuzz_more.py 5000 400; timeout 400 python tools/fuzz_tpb.py 2000 800; timeout 300 python tools/fuzz_intstat.py 3000 600 ) 2>&1 | grep -v "amdgpu.ids" | tee $O/fuzz24.txt | tail -30
