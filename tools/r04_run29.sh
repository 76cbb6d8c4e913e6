#!/bin/bash
# unaligned kernels: waves of one class of lanes, started on a line — tests, fuzz, same-box A/B against HEAD (lib_ab)
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_seg.py tests/test_gpu_hostpath.py tests/test_gpu_fuzz.py -m gpu -q -x > $O/t29.txt 2>&1; echo "rc $?" >> $O/t29.txt
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" $O/t29.txt | tail -8
timeout 400 python tools/fuzz_unaligned.py 5000 400 2>&1 | grep -v amdgpu | tail -6 | tee $O/fuzz29.txt
for L in lib lib_ab lib lib_ab; do
echo "== $L"
MTR_LIB=$PWD/meters.lv2_amd/$L/libmtr_engine.so python - <<PY 2>&1 | grep -v amdgpu
import sys; sys.path[:0]=['.','tests','tools']
import torch, meters.lv2_amd as M
def run(fs, meters, S=8192, steps=6, pad=0):
    T = int(fs) * 10
    stride = T + pad
    flat = torch.empty(S * stride * 2 + 64, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    ptr = flat.data_ptr()
    M.synth_fill_device(ptr, S, T, stride, 777, fs, 1, st)
    with M.Engine(S, fs, meters) as e:
        if meters & M.METER_EBU: e.integr_start()
        e.process_device(ptr, T, stride, st); torch.cuda.synchronize()
        e.timing_enable(True)
        for _ in range(steps): e.process_device(ptr, T, stride, st)
        torch.cuda.synchronize()
        pc = e.timing_calls()
        ms = float(sorted(pc[:, 0])[len(pc) // 2])
        print("fs %6.0f stride T+%d %-7s kernel median %.3f ms (min %.3f)  %.1f %% of 8 TB/s  seg %s" % (
            fs, pad, "ebu+tp" if meters & M.METER_EBU else "tp", ms, pc[:, 0].min(), 100 * S * T * 8 / (ms * 1e-3) / 8e12, e.seg_stats()), flush=True)
both = M.METER_EBU | M.METER_TRUEPEAK
run(44100.0, both); run(44100.0, M.METER_TRUEPEAK); run(88200.0, both, S=4096); run(44100.0, both, pad=3)
PY
done | tee $O/ab29.txt
