"""tools/seg_try.py — first contact of layout 7 (k_seg) with the hardware: a small parity check against layout 6 and the
oracle, then kernel milliseconds of layouts 6 and 7 on the bench shape (8192 streams x 10 s).  GPU box only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np, torch
import meters.lv2_amd as M
from make_golden import tri_noise
from _oracle import Oracle

def small():
    T, S = 2400 * 40 + 1234, 6
    x = np.stack([tri_noise(T, 700 + s, 2.0 ** -(s % 3), period=72000) for s in range(S)])
    orc = Oracle()
    res = {}
    for lay, segs in ((6, 0), (7, 1), (7, 3)):
        with M.Engine(S, 48000.0, M.METER_EBU | M.METER_TRUEPEAK, tune_layout=lay, tune_segments=segs) as e:
            e.integr_start()
            e.process(x)
            res[(lay, segs)] = (e.out9(), e.truepeak(), e.fragment_powers(), e.seg_stats())
    for k, (o9, tp, fr, st) in res.items():
        print("layout/segs", k, "seg_stats", st)
        for s in range(S):
            ref = orc.ebu(x[s], 48000.0, 2400, want_frag=True)
            rtp = orc.tp(x[s], 48000.0, 8192)
            print("  s%d frag rel %.2e  M %.4f/%.4f I %.3f/%.3f  tp rel %.2e %.2e" % (
                s, np.abs(fr[s] / ref["frag_power"] - 1).max(), o9[s, 0], ref["out9"][0], o9[s, 4], ref["out9"][4],
                abs(tp[s, 0] / rtp[0] - 1), abs(tp[s, 1] / rtp[1] - 1)))

def big(S=8192, secs=10.0, meters=None, label=""):
    fs = 48000.0
    T = int(secs * fs)
    meters = meters or (M.METER_EBU | M.METER_TRUEPEAK)
    buf = torch.empty((S, T, 2), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    M.synth_fill_device(buf.data_ptr(), S, T, T, 777, fs, 1, st)
    out = {}
    for lay in (6, 7, 6, 7):
        with M.Engine(S, fs, meters, tune_layout=lay) as e:
            if meters & M.METER_EBU: e.integr_start()
            e.process_device(buf.data_ptr(), T, T, st); torch.cuda.synchronize()
            e.timing_enable(True)
            for _ in range(5): e.process_device(buf.data_ptr(), T, T, st)
            torch.cuda.synchronize()
            q = e.timing_query()
            ms = q["ms_fused"] / q["calls"]
            print("%s layout %d: kernel %.3f ms  (%.1f %% of 8 TB/s)  gate %.3f  seg_stats %s" % (
                label, lay, ms, 100 * S * T * 8 / (ms * 1e-3) / 8e12, q["ms_gate"] / q["calls"], e.seg_stats()), flush=True)
            out[lay] = (e.out9() if meters & M.METER_EBU else None, e.truepeak())
    if out[6][0] is not None:
        d = np.abs(out[6][0][:, :5].astype(np.float64) - out[7][0][:, :5])
        print("  max |d| M maxM S maxS I between layouts:", d.max(0))
    r = np.abs(out[6][1].astype(np.float64) / out[7][1] - 1)
    print("  max rel dev of peaks:", r.max())

if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("all", "small"): small()
    if what in ("all", "big"):
        big(label="ebu+tp")
        big(meters=M.METER_TRUEPEAK, label="tp")
