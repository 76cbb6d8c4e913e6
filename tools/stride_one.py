"""tools/stride_one.py <pad> — k_seg at 44.1 kHz with stride T + pad, 4 calls (for rocprofv3)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import meters.lv2_amd as M
pad = int(sys.argv[1]); S = 8192; fs, T = 44100.0, 441000
stride = T + pad
flat = torch.empty(S * stride * 2 + 64, dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
M.synth_fill_device(flat.data_ptr(), S, T, stride, 777, fs, 1, st)
with M.Engine(S, fs, M.METER_EBU | M.METER_TRUEPEAK) as e:
    e.integr_start()
    for _ in range(4): e.process_device(flat.data_ptr(), T, stride, st)
    torch.cuda.synchronize()
