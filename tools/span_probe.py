"""tools/span_probe.py — k_seg at 44.1 kHz on fewer, longer streams (the same 29 GB): how far apart may the lanes of a wave be?  GPU box only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import meters.lv2_amd as M
fs = 44100.0
for S, secs, pad in ((8192, 10, 8), (4096, 20, 0), (2048, 40, 0), (1024, 80, 0), (512, 160, 0), (2048, 40, 8)):
    T = int(fs) * secs; stride = T + pad
    flat = torch.empty(S * stride * 2 + 64, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    M.synth_fill_device(flat.data_ptr(), S, T, stride, 777, fs, 1, st)
    with M.Engine(S, fs, M.METER_EBU | M.METER_TRUEPEAK) as e:
        e.integr_start()
        e.process_device(flat.data_ptr(), T, stride, st); torch.cuda.synchronize()
        e.timing_enable(True)
        for _ in range(4): e.process_device(flat.data_ptr(), T, stride, st)
        torch.cuda.synchronize()
        pc = e.timing_calls()
        print("S %5d x %3d s stride T+%d (%.1f MiB): kernel median %.3f ms  seg %s" % (S, secs, pad, stride * 8 / 2**20, float(sorted(pc[:, 0])[len(pc) // 2]), e.seg_stats()), flush=True)
    del flat
