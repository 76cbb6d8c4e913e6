#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
for v in "" _u8 _u8d _u8b _u8e _u8_nofetch; do echo "=== tpb_prof$v"; timeout 120 ./tools/tpb_prof$v 8192 96000 2>&1 | grep -v "amdgpu.ids\|shader cycles\| wave "; done > $O/tpb_u8_17.txt 2>&1
cat $O/tpb_u8_17.txt
MTR_LIB=$PWD/meters.lv2_amd/lib_u8/libmtr_engine.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_lv2_plugin.py tests/test_lv2_dr14.py tests/test_gpu_hostpath.py -m gpu -q -k "ballistics or dBTP or dr14 or TPnRMS or host" 2>&1 | tail -3
MTR_LIB=$PWD/meters.lv2_amd/lib_u8/libmtr_engine.so timeout 600 python tools/fuzz_tpb.py 0 300 2>&1 | tail -2
MTR_LIB=$PWD/meters.lv2_amd/lib_u8/libmtr_engine.so timeout 300 bash tools/tpb_ab.sh lib_u8 2>&1 | grep k_tpb
