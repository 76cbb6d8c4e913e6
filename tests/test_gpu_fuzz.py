"""Seeded shape fuzz of the default EBU R128 + true-peak path against the oracle.

The hand-picked cases of test_gpu_parity.py / test_gpu_layout6.py fix the sizes that matter by construction (tile and
block boundaries, LV2-sized calls, the rates of the reference's own plugins); this file draws the rest: sample rate,
number of streams, stream length, how the stream is cut into process() calls, time segments, exact pruning level, and
programme-like level changes — 96 + 32 + 32 seeded cases, and the bank and the integer
statistics kernels get 16 + 12 of their own, each checked per stream with the tolerances stated at the top of
test_gpu_parity.py (M / S 1e-3 dB, true peak 2e-6 relative, fragment powers 2e-5 relative, histograms <= 2 moved points)."""
import numpy as np
import pytest

import _signals as sig
from test_gpu_parity import _check_ebu, M  # noqa: F401  (M: the module fixture)

pytestmark = pytest.mark.gpu

RATES = (44100.0, 48000.0, 88200.0, 96000.0, 192000.0)


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    fs = RATES[int(rng.integers(len(RATES)))]
    S = int(rng.integers(1, 9))
    T = int(rng.integers(100, int(2.6 * fs))) if seed % 4 else int(rng.integers(100, 6000))   # up to ~50 fragments; every fourth case shorter than three
    ncall = int(rng.integers(1, 6))
    cuts = np.sort(rng.integers(1, T, size=ncall - 1)) if ncall > 1 else np.array([], np.int64)
    calls = np.diff(np.concatenate([[0], cuts, [T]])).astype(int)
    calls = [int(c) for c in calls if c > 0]
    x = np.empty((S, T, 2), np.float32)
    for s in range(S):
        n = sig.lcg_noise(T, 31 * seed + s, float(rng.uniform(0.05, 0.9)))
        # two or three level steps and a tone on one channel: peaks move, pruning gets something to do
        env = np.ones(T, np.float32)
        for _ in range(int(rng.integers(0, 4))):
            a = int(rng.integers(0, T))
            env[a:] *= np.float32(rng.choice([0.125, 0.5, 2.0]))
        n *= env[:, None]
        t = np.arange(T, dtype=np.float64) / fs
        n[:, s & 1] += (0.3 * np.sin(2 * np.pi * float(rng.uniform(50.0, 0.45 * fs)) * t)).astype(np.float32)
        x[s] = n
    kw = dict(tune_segments=int(rng.choice([0, 0, 2, 5])), tune_prune=int(rng.choice([0, 1, 2])))
    return fs, x, calls, kw


@pytest.mark.parametrize("seed", range(96))
def test_fuzzed_shapes_against_oracle(M, oracle, seed):  # noqa: F811
    fs, x, calls, kw = _case(seed)
    S, T = x.shape[0], x.shape[1]
    with M.Engine(S, fs, M.METER_EBU | M.METER_TRUEPEAK, **kw) as e:
        assert e.layout() == (7 if kw["tune_prune"] == 0 else 6)      # 7: calls that fit take the lane = segment kernel (forced by tune_segments)
        e.integr_start()
        pos, frags = 0, []
        for n in calls:
            e.process(x[:, pos:pos + n])
            frags.append(e.fragment_powers())
            pos += n
        out9, tp = e.out9(), e.truepeak()
        hm, hs = e.histograms()
        frag = np.concatenate(frags, 1)
    for s in range(S):
        o = oracle.ebu(x[s], fs, 1024, want_frag=True)
        _check_ebu(out9[s], (hm[s], hs[s]), o["out9"], (o["hist_M"], o["hist_S"]), None, frag[s], o["frag_power"])
        assert np.allclose(tp[s], oracle.tp(x[s], fs, 4096), rtol=2e-6), (seed, s, fs, calls, kw, tp[s])


@pytest.mark.parametrize("seed", range(32))
def test_fuzzed_truepeak_only(M, oracle, seed):  # noqa: F811
    fs, x, calls, kw = _case(100 + seed)
    with M.Engine(x.shape[0], fs, M.METER_TRUEPEAK, **kw) as e:
        pos = 0
        for n in calls:
            e.process(x[:, pos:pos + n])
            pos += n
        tp = e.truepeak()
    for s in range(x.shape[0]):
        assert np.allclose(tp[s], oracle.tp(x[s], fs, 4096), rtol=2e-6), (seed, s, fs, calls, kw)


@pytest.mark.parametrize("seed", range(32))
def test_fuzzed_ebu_only(M, oracle, seed):  # noqa: F811
    """k_kw, the K-weighting-only kernel (layout 4): its own tiling (39-frame runs through LDS) and pass 2."""
    fs, x, calls, kw = _case(200 + seed)
    S = x.shape[0]
    with M.Engine(S, fs, M.METER_EBU, tune_segments=kw["tune_segments"]) as e:
        assert e.layout() == 4
        e.integr_start()
        pos, frags = 0, []
        for n in calls:
            e.process(x[:, pos:pos + n])
            frags.append(e.fragment_powers())
            pos += n
        out9 = e.out9()
        hm, hs = e.histograms()
        frag = np.concatenate(frags, 1)
    for s in range(S):
        o = oracle.ebu(x[s], fs, 1024, want_frag=True)
        _check_ebu(out9[s], (hm[s], hs[s]), o["out9"], (o["hist_M"], o["hist_S"]), None, frag[s], o["frag_power"])


@pytest.mark.parametrize("seed", range(16))
def test_fuzzed_filter_bank(M, oracle, seed):  # noqa: F811
    """k_bank: stereo and mono engines, odd stream counts (lanes = (stream, band) pairs packed across streams), ragged
    calls; 1e-3 dB above -90 dB and 1e-4 relative on the linear levels, as in test_gpu_parity.py."""
    rng = np.random.default_rng(3000 + seed)
    fs = (44100.0, 48000.0, 96000.0)[int(rng.integers(3))]
    S = int(rng.integers(1, 24))
    T = int(rng.integers(300, 30000))
    mono = bool(seed & 1)
    x = np.stack([sig.lcg_noise(T, 900 + 17 * seed + s, float(rng.uniform(0.05, 0.8))) for s in range(S)])
    # equal blocks and a ragged tail, as a host would call run(): spectrum_run adds 1e-20f to the levels once per call
    # (spectrumlv2.c:236), so the oracle has to see the same blocks
    blk = int(rng.integers(max(T // 6, 1), T + 1))
    calls = [blk] * (T // blk) + ([T % blk] if T % blk else [])
    feed = np.ascontiguousarray(x[:, :, 0]) if mono else x
    with M.Engine(S, fs, M.METER_SPECTR30, n_channels=1 if mono else 2) as e:
        pos = 0
        for n in calls:
            e.process(np.ascontiguousarray(feed[:, pos:pos + n]))
            pos += n
        r = e.spectrum()
    for s in sorted({0, S // 2, S - 1}):
        ref_in = np.repeat(x[s, :, :1], 2, axis=1) if mono else x[s]        # (L + L) / 2 = L
        o = oracle.spectr(ref_in, fs, blk)
        assert np.allclose(r["val"][s], o["val"], rtol=1e-4, atol=1e-30), (seed, s, fs, mono, calls)
        live = o["val_db"] > -90
        assert np.allclose(r["val_db"][s][live], o["val_db"][live], atol=1e-3), (seed, s)


@pytest.mark.parametrize("seed", range(12))
def test_fuzzed_integer_statistics(M, oracle, seed):  # noqa: F811
    """k_bitstats / k_sigdist: bit-exact histograms and counters for ragged lengths, several calls, scaled and sparse data."""
    rng = np.random.default_rng(4000 + seed)
    S = int(rng.integers(1, 10))
    T = int(rng.integers(50, 70000))
    x = np.stack([sig.lcg_noise(T, 1200 + 13 * seed + s, float(2.0 ** -int(rng.integers(0, 30))))[:, 0] for s in range(S)])
    if S > 1:
        x[1, ::int(rng.integers(2, 9))] = 0.0
    cuts = np.sort(rng.integers(1, T, size=int(rng.integers(0, 3))))
    calls = [int(c) for c in np.diff(np.concatenate([[0], cuts, [T]])) if c > 0]
    with M.Engine(S, 48000.0, M.METER_BITSTATS, n_channels=1) as e:
        pos = 0
        for n in calls:
            e.process(np.ascontiguousarray(x[:, pos:pos + n]))
            pos += n
        got = e.bitstats()
    for s in range(S):
        want = oracle.bitstats(x[s])
        assert np.array_equal(got["hist"][s], want["hist"]) and np.array_equal(got["counters"][s], want["counters"]), (seed, s)
        assert got["vmin"][s] == want["vmin"] and got["vmax"][s] == want["vmax"]
    y = (x * np.float32(rng.uniform(0.5, 2.0))).astype(np.float32)
    with M.Engine(S, 48000.0, M.METER_SIGDIST, n_channels=1) as e:
        pos = 0
        for n in calls:
            e.process(np.ascontiguousarray(y[:, pos:pos + n]))
            pos += n
        got = e.sigdist()
    for s in range(S):
        want = oracle.sigdist(y[s])
        assert np.array_equal(got["bins"][s], want["bins"]), (seed, s)
        assert got["peak_cnt"][s] == want["peak_cnt"] and got["peak_bin"][s] == want["peak_bin"] and got["count"][s] == want["count"]
