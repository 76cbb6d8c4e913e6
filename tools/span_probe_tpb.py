"""tools/span_probe_tpb.py — do the other kernels mind how far apart the streams of a workgroup lie?  The same 31.5 GB as fewer, longer
streams: true-peak ballistics (k_tpb: a workgroup reads 32 streams), DR-14, K-meter, the 30-band bank (4096 x 10 s shape).  GPU box only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import meters.lv2_amd as M
fs = 48000.0
for name, meters, shapes in (("tpb", M.METER_TPBALLIST, ((8192, 10), (2048, 40), (1024, 80), (512, 160), (256, 320))),
                             ("dr14", M.METER_DR14, ((8192, 10), (1024, 80), (256, 320))),
                             ("kmeter", M.METER_KMETER, ((8192, 10), (1024, 80), (256, 320))),
                             ("ebu", M.METER_EBU, ((8192, 10), (256, 320))),
                             ("bank", M.METER_SPECTR30, ((2048, 10), (256, 80), (64, 320)))):
    for S, secs in shapes:
        T = int(fs) * secs
        buf = torch.empty((S, T, 2), dtype=torch.float32, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        M.synth_fill_device(buf.data_ptr(), S, T, T, 777, fs, 1, st)
        with M.Engine(S, fs, meters) as e:
            if meters & M.METER_EBU: e.integr_start()
            e.process_device(buf.data_ptr(), T, T, st); torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(2): e.process_device(buf.data_ptr(), T, T, st)
            ev[1].record(); torch.cuda.synchronize()
            ms = ev[0].elapsed_time(ev[1]) / 2
        print("%-6s S %5d x %3d s (stream %.1f MiB): %.3f ms per pass, %.2f TB/s" % (name, S, secs, T * 8 / 2**20, ms, S * T * 8 / ms / 1e9), flush=True)
        del buf
