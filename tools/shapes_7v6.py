"""tools/shapes_7v6.py — big shapes through the default engine (layout 7: k_seg where the call fits) and through layout 6 (k_kwtp16):
peaks within 2e-6 relative, M / S / integrated within 1e-3 / 0.01 dB, the same histogram counts.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import meters.lv2_amd as M
for S, T, fs in ((65536, 48000, 48000.0), (8193, 480000, 48000.0), (64, 3600 * 48000, 48000.0), (1000, 60 * 96000, 96000.0), (5000, 441000 * 2, 44100.0), (8192, 100001, 88200.0)):
    buf = torch.empty((S, T, 2), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    M.synth_fill_device(buf.data_ptr(), S, T, T, 5 + S, fs, 1, st)
    out = {}
    for lay in (7, 6):
        with M.Engine(S, fs, M.METER_EBU | M.METER_TRUEPEAK, tune_layout=lay) as e:
            e.integr_start()
            e.process_device(buf.data_ptr(), T, T, st); torch.cuda.synchronize()
            hm, hs = e.histograms()
            out[lay] = (e.out9(), e.truepeak(), hm.sum(), hs.sum(), e.seg_stats())
    a, b = out[7], out[6]
    rel = np.abs(a[1].astype(np.float64) / b[1] - 1).max()
    d = np.abs(a[0][:, :4].astype(np.float64) - b[0][:, :4]).max()
    di = np.abs(a[0][:, 4].astype(np.float64) - b[0][:, 4]).max()
    ok = rel <= 2e-6 and d <= 1e-3 and di <= 0.01 and a[2] == b[2] and a[3] == b[3]
    print("S=%d T=%d fs=%g: seg_stats %s  peaks rel %.1e  M/S %.1e dB  I %.1e dB  points %d/%d  %s" % (S, T, fs, a[4], rel, d, di, a[2], a[3], "ok" if ok else "MISMATCH"), flush=True)
    del buf
    torch.cuda.empty_cache()
