#!/bin/bash
# k_seg takes the call's last fragment at 44.1 / 88.2 kHz too: tests, fuzz, and the same-box A/B against HEAD (lib_ab)
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_seg.py tests/test_gpu_hostpath.py tests/test_gpu_layout6.py tests/test_gpu_fuzz.py -m gpu -q -x > $O/t25.txt 2>&1; echo "rc $?" >> $O/t25.txt
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" $O/t25.txt | tail -8
timeout 400 python tools/fuzz_more.py 7000 300 2>&1 | grep -v amdgpu | tail -5 | tee $O/fuzz25.txt
for L in lib lib_ab lib lib_ab; do
echo "== $L"
MTR_LIB=$PWD/meters.lv2_amd/$L/libmtr_engine.so python - <<PY 2>&1 | grep -v amdgpu
import sys; sys.path[:0]=['.','tests','tools']
import align_probe as ap, meters.lv2_amd as M
both = M.METER_EBU | M.METER_TRUEPEAK
ap.run(44100.0, 0, both); ap.run(44100.0, 0, M.METER_TRUEPEAK); ap.run(88200.0, 0, both, S=4096)
PY
done | tee $O/ab25.txt
