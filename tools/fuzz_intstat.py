"""tools/fuzz_intstat.py [first_seed] [count] — more seeds of the integer-statistics fuzz (k_bitstats, k_sigdist incl. the reference's
mean / variance recurrence once samples were out of range) and of the filter bank, against the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np
import meters.lv2_amd as M
import _signals as sig
from _oracle import Oracle
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
orc = Oracle()
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(4000 + seed)
    S = int(rng.integers(1, 10)); T = int(rng.integers(50, 70000))
    x = np.stack([sig.lcg_noise(T, 1200 + 13 * seed + s, float(2.0 ** -int(rng.integers(0, 30))))[:, 0] for s in range(S)])
    if S > 1: x[1, ::int(rng.integers(2, 9))] = 0.0
    cuts = np.sort(rng.integers(1, T, size=int(rng.integers(0, 3))))
    calls = [int(c) for c in np.diff(np.concatenate([[0], cuts, [T]])) if c > 0]
    try:
        with M.Engine(S, 48000.0, M.METER_BITSTATS, n_channels=1) as e:
            pos = 0
            for n in calls: e.process(np.ascontiguousarray(x[:, pos:pos + n])); pos += n
            got = e.bitstats()
        for s in range(S):
            want = orc.bitstats(x[s])
            assert np.array_equal(got["hist"][s], want["hist"]) and np.array_equal(got["counters"][s], want["counters"]), ("bit", s)
        y = (x * np.float32(rng.uniform(0.5, 2.0))).astype(np.float32)
        if seed % 2:                                                  # out-of-range samples (|v| > 1.2), NaN: the skipped-sample regime
            for s in range(S):
                idx = rng.integers(0, T, size=int(rng.integers(1, 6)))
                y[s, idx] = np.float32(rng.choice([1.5, -3.0, np.nan, 1.21]))
        with M.Engine(S, 48000.0, M.METER_SIGDIST, n_channels=1) as e:
            pos = 0
            for n in calls: e.process(np.ascontiguousarray(y[:, pos:pos + n])); pos += n
            got = e.sigdist()
        for s in range(S):
            want = orc.sigdist(y[s])
            assert np.array_equal(got["bins"][s], want["bins"]), ("bins", s)
            assert got["peak_cnt"][s] == want["peak_cnt"] and got["peak_bin"][s] == want["peak_bin"] and got["count"][s] == want["count"], ("peak", s)
            for k in ("var_m", "var_s"):
                if k in got and k in want:
                    a, b = float(got[k][s]), float(want[k])
                    assert abs(a - b) <= 1e-10 * max(1.0, abs(b)), (k, s, a, b)
    except AssertionError as ex:
        bad += 1; print("FAIL seed", seed, S, T, calls, str(ex)[:200], flush=True)
print("seeds %d..%d: %d failed" % (first, first + count - 1, bad))
