"""tools/power_probe.py — is k_kwtp16 power-limited?  The same launch on zeros, on small-amplitude noise and on the
bench programme: identical instruction streams, different operand toggling (MI355X_MICROARCH.md, DVFS give-back)."""
import sys
import torch
import meters.lv2_amd as M
S, T, fs = 8192, 480000, 48000.0
buf = torch.zeros((S, T, 2), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for what in ("zeros", "programme", "uniform noise", "constant 0.5"):
    if what == "programme":
        M.synth_fill_device(buf.data_ptr(), S, T, T, 777, fs, 1, st)
    elif what == "uniform noise":
        M.synth_fill_device(buf.data_ptr(), S, T, T, 777, fs, 0, st)
    elif what == "constant 0.5":
        buf.fill_(0.5)
    torch.cuda.synchronize()
    for meters, name in ((M.METER_EBU | M.METER_TRUEPEAK, "ebu+tp"), (M.METER_TRUEPEAK, "tp")):
        with M.Engine(S, fs, meters) as e:
            if meters & M.METER_EBU:
                e.integr_start()
            e.process_device(buf.data_ptr(), T, T, st)
            torch.cuda.synchronize()
            e.timing_enable(True)
            for _ in range(4):
                e.process_device(buf.data_ptr(), T, T, st)
            torch.cuda.synchronize()
            q = e.timing_query()
            print("%-14s %-7s %7.3f ms" % (what, name, q["ms_fused"] / q["calls"]), flush=True)
