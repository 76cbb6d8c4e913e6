"""tools/stride_probe.py <out.npz> [S] — k_seg at 44.1 kHz over strides T + 0, 1, 2, 3, 4, 8 frames (how the unaligned kernels sort their lanes
into waves depends on stride mod 16): kernel ms, and every stream's results into <out.npz> for a comparison between two builds.  GPU box only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import meters.lv2_amd as M
out = {}
S = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
fs, T = 44100.0, 441000
for pad in (0, 1, 2, 3, 4, 8):
    stride = T + pad
    flat = torch.empty(S * stride * 2 + 64, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    M.synth_fill_device(flat.data_ptr(), S, T, stride, 777, fs, 1, st)
    with M.Engine(S, fs, M.METER_EBU | M.METER_TRUEPEAK) as e:
        e.integr_start()
        e.process_device(flat.data_ptr(), T, stride, st); torch.cuda.synchronize()
        out["o9_%d" % pad] = e.out9(); out["tp_%d" % pad] = e.truepeak(); out["fr_%d" % pad] = e.fragment_powers()[::64]
        e.timing_enable(True)
        for _ in range(4): e.process_device(flat.data_ptr(), T, stride, st)
        torch.cuda.synchronize()
        pc = e.timing_calls()
        print("stride T+%d: kernel median %.3f ms  seg %s" % (pad, float(sorted(pc[:, 0])[len(pc) // 2]), e.seg_stats()), flush=True)
np.savez(sys.argv[1], **out)
