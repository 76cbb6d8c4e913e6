"""tools/bank_mono_probe.py — would config 4 (EBU R128 + true peak + 30-band bank) gain from ONE pass over the audio?
The bank (k_bank) works on the mono mix (L + R) / 2 (src/spectrumlv2.c:216).  A one-pass form would have k_seg write that
mix into a [S][T] f32 side buffer (15.7 GB written, 15.7 GB read back: the same HBM traffic as the 31.5 GB re-read) and
k_bank read it.  What k_bank would then save is the second channel's bytes and one add per frame — this probe measures
exactly that: k_bank on the stereo batch against k_bank on the pre-mixed mono batch of the same streams (a mono engine,
n_channels = 1: the reference's spectr30mono), same box, alternating.  GPU box only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import meters.lv2_amd as M

S, fs = 8192, 48000.0
T = 480000
buf = torch.empty((S, T, 2), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
M.synth_fill_device(buf.data_ptr(), S, T, T, 777, fs, 1, st)
mono = torch.empty((S, T), dtype=torch.float32, device="cuda")
for s0 in range(0, S, 512):                                          # (L + R) / 2 exactly as spectrumlv2.c:216 forms it
    mono[s0:s0 + 512] = (buf[s0:s0 + 512, :, 0] + buf[s0:s0 + 512, :, 1]) / 2.0
torch.cuda.synchronize()

def run(ptr, chn, stride, reps=2):
    with M.Engine(S, fs, M.METER_SPECTR30, n_channels=chn) as e:
        e.process_device(ptr, T, stride, st); torch.cuda.synchronize()
        e.timing_enable(True)
        for _ in range(reps): e.process_device(ptr, T, stride, st)
        torch.cuda.synchronize()
        pc = e.timing_calls()
        sp = e.spectrum(0, 4)
        return float(np.median(pc[:, 2])), sp["val_db"]

for rep in range(2):
    a, va = run(buf.data_ptr(), 2, T)
    b, vb = run(mono.data_ptr(), 1, T)
    print("k_bank, 8192 streams x 10 s:  stereo input %.2f ms (%.1f GB read)   pre-mixed mono input %.2f ms (%.1f GB read)   band levels equal: %s" % (
        a, S * T * 8 / 1e9, b, S * T * 4 / 1e9, bool(np.array_equal(va, vb))), flush=True)
