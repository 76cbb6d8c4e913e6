"""tools/fuzz_more.py [first_seed] [count] — the shape fuzz of tests/test_gpu_fuzz.py on more seeds, with the lane = time segment kernel
forced (tune_segments 1..5, no pruning): sample rate, streams, length, how the stream is cut into calls.  Prints what failed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np
import meters.lv2_amd as M
from _oracle import Oracle
from test_gpu_fuzz import _case
from test_gpu_parity import _check_ebu

first = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
orc = Oracle()
bad = took = 0
for seed in range(first, first + count):
    fs, x, calls, kw = _case(seed)
    kw = dict(tune_segments=1 + seed % 5, tune_prune=0)
    S = x.shape[0]
    try:
        with M.Engine(S, fs, M.METER_EBU | M.METER_TRUEPEAK, **kw) as e:
            e.integr_start()
            pos, frags = 0, []
            for n in calls:
                e.process(x[:, pos:pos + n]); frags.append(e.fragment_powers()); pos += n
            out9, tp = e.out9(), e.truepeak()
            hm, hs = e.histograms()
            frag = np.concatenate(frags, 1)
            took += e.seg_stats()[0] > 0
        for s in range(S):
            o = orc.ebu(x[s], fs, 1024, want_frag=True)
            _check_ebu(out9[s], (hm[s], hs[s]), o["out9"], (o["hist_M"], o["hist_S"]), None, frag[s], o["frag_power"])
            assert np.allclose(tp[s], orc.tp(x[s], fs, 4096), rtol=2e-6), ("tp", s, tp[s])
    except AssertionError as ex:
        bad += 1
        print("FAIL seed", seed, fs, S, x.shape[1], calls, kw, str(ex)[:300], flush=True)
print("seeds %d..%d: %d failed, %d used k_seg" % (first, first + count - 1, bad, took))
