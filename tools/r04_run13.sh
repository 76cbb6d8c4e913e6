#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
python - <<PY 2>&1 | grep -v amdgpu > $O/seg44_13.txt
import sys; sys.path[:0]=['.','tests','tools']
import align_probe as ap, meters.lv2_amd as M
both = M.METER_EBU | M.METER_TRUEPEAK
for rep in range(3):
    ap.run(48000.0, 0, both); ap.run(44100.0, 0, both)
PY
cat $O/seg44_13.txt
for fs in 48000 44100; do
  MTR_LIB=$PWD/meters.lv2_amd/lib_prof/libmtr_engine.so timeout 300 python tools/seg_prof.py ebu+tp $fs 2>&1 | grep -v amdgpu
done > $O/seg_prof13.txt; cat $O/seg_prof13.txt
timeout 600 python -m pytest tests/test_gpu_seg.py tests/test_gpu_parity.py -m gpu -q -k "seg or full_size or 44" 2>&1 | tail -3
