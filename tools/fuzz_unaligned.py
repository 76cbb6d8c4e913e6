"""tools/fuzz_unaligned.py [first_seed] [count] — calls that END on or near a fragment boundary at 44.1 / 88.2 / 22.05 kHz (fragments of 2205 /
4410 / 1102 frames: not whole 16-frame steps), through k_seg with 1..6 segments, EBU R128 + true peak and true peak alone: the lanes of a
stream's last segment stop inside the launch's last step (mtr_seg.hip).  A spike is planted in the last frames of some streams; the next
stream opens with a large value.  Against the oracle, per call."""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np
import meters.lv2_amd as M
import _signals as sig
from _oracle import Oracle
from test_gpu_parity import _check_ebu

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
orc = Oracle()
bad = took = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(52000 + seed)
    fs = float(rng.choice([44100.0, 88200.0, 22050.0]))
    fragm = int(fs) // 20
    S = int(rng.integers(1, 7))
    calls = []
    for _ in range(int(rng.integers(1, 4))):
        calls.append(int(rng.integers(6, 40)) * fragm + int(rng.choice([0, 0, 0, 1, 5, 15, 16, 17, fragm - 1, -1, -7, -16])))
    T = sum(calls)
    x = np.stack([sig.lcg_noise(T, 97 * seed + s, float(rng.uniform(0.05, 0.5))) for s in range(S)])
    pos = 0
    for n in calls:
        pos += n
        for s in range(S):
            if rng.integers(3) == 0:
                x[s, pos - int(rng.integers(1, 40))] = (rng.uniform(0.7, 0.99), -rng.uniform(0.7, 0.99))
            if rng.integers(3) == 0 and pos < T:
                x[s, pos:pos + 8] = rng.choice([30.0, 3.0, -2.5])       # ... and right behind the call's end
    tp_only = bool(seed % 3 == 0)
    meters = M.METER_TRUEPEAK if tp_only else (M.METER_EBU | M.METER_TRUEPEAK)
    kw = dict(tune_segments=1 + seed % 6, tune_layout=7)
    try:
        with M.Engine(S, fs, meters, **kw) as e:
            if not tp_only: e.integr_start()
            pos, frags, percall = 0, [], []
            for n in calls:
                e.process(x[:, pos:pos + n]); pos += n
                if not tp_only: frags.append(e.fragment_powers())
                percall.append(np.array([[r.truepeak_call[0], r.truepeak_call[1]] for r in e.results()], np.float32))
            tp = e.truepeak()
            took += e.seg_stats()[0] > 0
            if not tp_only:
                out9 = e.out9(); hm, hs = e.histograms(); frag = np.concatenate(frags, 1)
        with M.Engine(S, fs, M.METER_TRUEPEAK, tune_layout=6) as e:      # per-call peaks: the wave-per-segment kernel's (held to process_max block by block elsewhere)
            pos, percall6 = 0, []
            for n in calls:
                e.process(x[:, pos:pos + n]); pos += n
                percall6.append(np.array([[r.truepeak_call[0], r.truepeak_call[1]] for r in e.results()], np.float32))
        for s in range(S):
            assert np.allclose(tp[s], orc.tp(x[s], fs, 4096), rtol=2e-6), ("tp", s, tp[s], orc.tp(x[s], fs, 4096))
            for c in range(len(calls)): assert np.allclose(percall[c][s], percall6[c][s], rtol=2e-6), ("call", c, s, percall[c][s], percall6[c][s])
            if not tp_only:
                o = orc.ebu(x[s], fs, 1024, want_frag=True)
                _check_ebu(out9[s], (hm[s], hs[s]), o["out9"], (o["hist_M"], o["hist_S"]), None, frag[s], o["frag_power"])
    except (AssertionError, M.EngineError) as ex:
        bad += 1
        print("FAIL seed", seed, fs, S, calls, kw, "tp" if tp_only else "ebu+tp", str(ex)[:300], flush=True)
        print("   ", " | ".join(l.strip() for l in traceback.format_exc().splitlines()[-4:-1])[:400], flush=True)
        if not tp_only and "frag" in traceback.format_exc():
            d = np.abs(frag[s].astype(np.float64) / np.maximum(o["frag_power"], 1e-300) - 1)
            print("    stream", s, "worst fragments", np.argsort(d)[-4:], d[np.argsort(d)[-4:]], "frag_power there", o["frag_power"][np.argsort(d)[-4:]], flush=True)
print("seeds %d..%d: %d failed, %d used k_seg" % (first, first + count - 1, bad, took))
