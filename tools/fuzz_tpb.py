"""tools/fuzz_tpb.py [first_seed] [count] — shape fuzz of TruePeakdsp::process (k_tpb) against the oracle's object: sample rate, mono /
stereo, streams, length, how the stream is cut into calls (every call = one process () + read (m, p))."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np
import meters.lv2_amd as M
import _signals as sig
from _oracle import Oracle, MoTp

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
orc = Oracle()
TOL = 4e-6          # level and peak, RELATIVE TO THE VALUE (the levels drawn below go down to 2^-13: a bound on max (1, value) would say nothing)
FLOOR = 4e-7        # ... plus this much of the largest INPUT magnitude inside the interpolator's window of the call: an output that is the
                    # cancelling tail of a 48-tap sum (the first frames behind silence, e.g.) is far smaller than its terms, and any two correct
                    # f32 evaluations of it — the reference's fmaf chain is one — differ by ~2^-24 x sum |g| |x| = 1.5e-7 of that magnitude
bad = 0
worst = {}
for seed in range(first, first + count):
    rng = np.random.default_rng(7000 + seed)
    fs = float(rng.choice([44100.0, 48000.0, 88200.0, 96000.0, 192000.0]))
    chn = int(rng.choice([1, 2]))
    S = int(rng.choice([1, 2, 7, 31, 32, 33, 64, 65, 97]))
    T = int(rng.integers(1, 9000)) if seed % 3 else int(rng.integers(1, 70))
    ncall = int(rng.integers(1, 5))
    cuts = np.sort(rng.integers(1, max(T, 2), size=ncall - 1)) if ncall > 1 and T > 1 else np.array([], np.int64)
    calls = [int(c) for c in np.diff(np.concatenate([[0], cuts, [T]])) if c > 0]
    x = np.stack([sig.lcg_noise(T, 13 * seed + s, float(2.0 ** -rng.integers(3, 14))) for s in range(S)])
    if seed % 5 == 0:
        x[:, T // 2:] *= np.float32(2.0 ** 6)                        # a level jump: the column scales move (levels stay under the clamp at
                                                                     # 20, which the oracle's 8192-frame blocks would apply inside a call)
    if seed % 7 == 0:
        x[:, :T // 3] = 0.0                                            # digital silence first: the column scales start at their ceiling
    if seed % 11 == 0:
        x[:, 2 * T // 3:] *= np.float32(2.0 ** -18)                   # a drop the hysteresis has to follow (the scale grows back)
    if seed % 13 == 0 and T > 100:
        x[:, T // 4::max(T // 9, 70), 0] = np.float32(0.9)            # lone loud samples: shrink for 64 frames, grow again
    if chn == 1:
        x = np.ascontiguousarray(x[:, :, :1])
    try:
        with M.Engine(S, fs, M.METER_TPBALLIST, n_channels=chn) as e:
            got, pos = [], 0
            for n in calls:
                e.process(x[:, pos:pos + n] if chn == 2 else np.ascontiguousarray(x[:, pos:pos + n, 0]))
                r = e.results()
                got.append([[(r[s].tpb_level[c], r[s].tpb_peak[c]) for c in range(chn)] for s in range(S)])
                pos += n
        for s in sorted(set([0, S // 2, S - 1])):
            for c in range(chn):
                ch = np.ascontiguousarray(x[s, :, c])
                t = MoTp(); orc.lib.mo_tp_init(C.byref(t), fs)
                m, p = C.c_float(), C.c_float()
                pos = 0
                for i, n in enumerate(calls):
                    mm = pp = 0.0
                    for o in range(pos, pos + n, 8192):
                        seg = np.ascontiguousarray(ch[o:min(o + 8192, pos + n)])
                        orc.lib.mo_tp_process(C.byref(t), seg, seg.size); orc.lib.mo_tp_read2(C.byref(t), C.byref(m), C.byref(p))
                        mm, pp = max(mm, m.value), max(pp, p.value)
                    wmax = float(np.abs(ch[max(pos - 47, 0):pos + n]).max())
                    for k, want in ((0, mm), (1, pp)):
                        dev = abs(got[i][s][c][k] - want)
                        if dev > FLOOR * wmax:
                            worst[fs] = max(worst.get(fs, 0.0), dev / max(1e-37, want))
                        assert dev <= TOL * want + FLOOR * wmax + 1e-37, ("mp"[k], s, c, i, got[i][s][c][k], want, wmax)
                    pos += n
    except AssertionError as ex:
        bad += 1
        print("FAIL seed", seed, fs, chn, S, T, calls, str(ex)[:200], flush=True)
print("seeds %d..%d: %d failed; largest deviation by rate:" % (first, first + count - 1, bad), {k: "%.2e" % v for k, v in sorted(worst.items())})
