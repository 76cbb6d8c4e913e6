"""Layout 6 (mtr_fused4.hip): K-weighting as k_kw + the 4x interpolator on the matrix pipe at f32 grade — samples
and taps as two f16 halves each, three partial products, f32 accumulation (mtr_mfma16_fir.h).

The bar is the one the exact-f32 VALU interpolator is held to (tests/test_gpu_parity.py): true peaks within
2e-6 relative of the oracle (= the reference's Resampler + TruePeakdsp::process_max), per call and held;
loudness is the same arithmetic as in every other layout.  Measured worst case over this file: 3e-7."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_golden import tri_noise  # noqa: E402

TP_RTOL = 2e-6


@pytest.fixture(scope="module")
def M():
    import meters.lv2_amd as m
    return m


def _rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), 1e-300)


def _run(M, x, calls, fs=48000.0, meters=None, **kw):
    meters = meters if meters is not None else (M.METER_EBU | M.METER_TRUEPEAK)
    with M.Engine(x.shape[0], fs, meters, **kw) as e:
        if meters & M.METER_EBU:
            e.integr_start()
        pos, per_call = 0, []
        for n in calls:
            e.process(x[:, pos:pos + n])
            per_call.append(np.array([[r.truepeak_call[0], r.truepeak_call[1]] for r in e.results()], np.float32))
            pos += n
        o9 = e.out9() if meters & M.METER_EBU else None
        hist = e.histograms() if meters & M.METER_EBU else None
        return o9, e.truepeak(), np.stack(per_call), hist


@pytest.mark.parametrize("fs", [48000.0, 44100.0, 96000.0, 192000.0])
@pytest.mark.parametrize("segs", [0, 3])
def test_layout6_matches_f32_layout_and_oracle(M, oracle, fs, segs):
    T = int(fs) * 6 + 1
    calls = [1001, int(fs) * 3, 47, T - 1001 - int(fs) * 3 - 47]
    x = np.stack([tri_noise(T, 500 + s, 2.0 ** -(s % 3), period=72000) for s in range(3)])
    o3, p3, c3, h3 = _run(M, x, calls, fs, tune_segments=segs, tune_layout=3)
    o6, p6, c6, h6 = _run(M, x, calls, fs, tune_segments=segs, tune_layout=6)
    # loudness: the K-filter is the same arithmetic in every layout (summation grouping differs: 38 vs 39 frame runs)
    assert np.allclose(o6[:, :4], o3[:, :4], atol=1e-3)
    assert np.all(np.abs(o6[:, 4] - o3[:, 4]) <= 0.01)
    for a, b in zip(h6, h3):
        assert a.sum() == b.sum() and np.abs(a - b).sum() // 2 <= 2
    # peaks: per call against the exact-f32 layout, held against the oracle
    assert _rel(c6, c3).max() <= TP_RTOL, _rel(c6, c3).max()
    for s in range(3):
        tp = oracle.tp(x[s], fs, 8192)
        assert _rel(p6[s], tp).max() <= TP_RTOL, (s, p6[s], tp)


def test_layout6_edge_signals(M, oracle):
    """Impulses next to tile and call boundaries, the full-scale fs/4 pattern (+3.01 dBTP), silence, streams from
    2^-60 to 1e30 (the per-tile scale keeps 22+ bits at any level), a stream whose level jumps by 2^40 mid-tile."""
    import _signals as sig
    T = 48000 * 3
    spike = np.zeros((T, 2), np.float32)
    spike[2399, 0] = 1.0; spike[2400, 1] = -1.0; spike[T - 1, 0] = 0.5; spike[50000, 1] = 0.25
    g3 = np.tile(np.array([1, 1, -1, -1], np.float32), T // 4)[:, None].repeat(2, 1)
    n = sig.lcg_noise(T, 11, 1.0).astype(np.float32)
    quiet, tiny, hot, huge = (n * np.float32(2.0 ** -20), n * np.float32(2.0 ** -60), n * np.float32(8.0), n * np.float32(1e30))
    jump = n.copy()
    jump[:70000] *= np.float32(2.0 ** -40)
    lr = n.copy()
    lr[:, 1] *= np.float32(2.0 ** -30)                      # channels 180 dB apart: the scales are per channel
    x = np.stack([spike, g3, np.zeros((T, 2), np.float32), quiet, tiny, hot, huge, jump, lr])
    calls = [2399, 1, 100000, T - 102400]
    _, p3, c3, _ = _run(M, x, calls, tune_segments=0, tune_layout=3)
    _, p6, c6, _ = _run(M, x, calls, tune_segments=0, tune_layout=6)
    assert np.all(p6[2] == 0.0)
    m = c3 > 0
    assert np.all((c6 > 0) == m)
    assert _rel(c6[m], c3[m]).max() <= TP_RTOL, _rel(c6[m], c3[m]).max()
    for s in (0, 1, 3, 4, 5, 6, 7, 8):
        assert _rel(p6[s], oracle.tp(x[s], 48000.0, 8192)).max() <= TP_RTOL, s
    assert abs(20 * np.log10(p6[1, 0]) - 3.1056) < 1e-3            # 1.429816


def test_layout6_short_calls(M, oracle):
    """LV2-sized blocks: every call is its own first and last tile, shorter than the interpolator's memory; phase 0
    (|x[n - 24]|) must come from the right side of each call boundary."""
    sizes = [1, 7, 23, 24, 25, 47, 48, 49, 64, 100, 255, 256, 257, 1024, 2399, 2401, 5000]
    T = sum(sizes)
    x = np.stack([tri_noise(T, 40 + s, 0.7, period=900) for s in range(2)])
    x[0, 30, 0] = 0.99; x[1, 180, 1] = -0.98                    # isolated peaks that phase 0 sees 24 frames later
    _, p3, c3, _ = _run(M, x, sizes, tune_layout=3)
    _, p6, c6, _ = _run(M, x, sizes, tune_layout=6)
    assert _rel(c6, c3).max() <= TP_RTOL, _rel(c6, c3).max()
    assert _rel(p6, p3).max() <= TP_RTOL
    # and the per-call peaks are the reference's: TruePeakdsp::process_max block by block
    for s in range(2):
        for c in range(2):
            pos, orc = 0, oracle.tp_stream(48000.0)
            for i, nfr in enumerate(sizes):
                want = orc.process(x[s, pos:pos + nfr, c])
                assert _rel(c6[i, s, c], want) <= TP_RTOL, (s, c, i, c6[i, s, c], want)
                pos += nfr


def test_layout6_truepeak_only(M, oracle):
    T = 48000 * 2 + 333
    x = np.stack([tri_noise(T, 900 + s, 0.5, period=30000) for s in range(5)])
    _, p6, _, _ = _run(M, x, [T], meters=M.METER_TRUEPEAK, tune_layout=6)
    for s in range(5):
        assert _rel(p6[s], oracle.tp(x[s], 48000.0, 8192)).max() <= TP_RTOL, s


def test_layout6_nonfinite(M):
    x = np.zeros((3, 2400 * 4, 2), np.float32)
    x[0, :, 0] = np.nan                                     # NaN never wins a max, never sticks in state
    x[0, :, 1] = 0.5
    x[1, 7, 1] = np.inf
    x[1, :, 0] = 0.25
    x[2, 5000, 0] = np.nan                                  # one NaN poisons 48 outputs, no more
    x[2, :5000, 0] = 0.125
    with M.Engine(3, 48000.0, M.METER_EBU | M.METER_TRUEPEAK, tune_layout=6) as e, \
         M.Engine(3, 48000.0, M.METER_EBU | M.METER_TRUEPEAK, tune_layout=3) as e3:
        for eng in (e, e3):
            eng.integr_start()
            eng.process(x)
            eng.process(np.full((3, 2400 * 8, 2), 0.25, np.float32))
        r, r3 = e.results(), e3.results()
    assert r[0].truepeak[0] < 1.0 and np.isfinite(r[0].loudness_M)
    assert r[1].truepeak[1] == np.inf
    for s in range(3):
        for c in range(2):
            a, b = r[s].truepeak[c], r3[s].truepeak[c]
            assert (a == b) or abs(a - b) <= TP_RTOL * abs(b), (s, c, a, b)


def test_layout6_exact_pruning_changes_nothing_but_time(M):
    import _signals as sig
    T = 48000 * 8
    loud_then_quiet = sig.lcg_noise(T, 5, 0.5)
    loud_then_quiet[48000:] *= np.float32(0.125)
    ramp_up = (sig.lcg_noise(T, 6, 0.5) * np.linspace(0.05, 1.0, T, dtype=np.float32)[:, None]).astype(np.float32)
    steady = sig.lcg_noise(T, 7, 0.5)
    spike = np.zeros((T, 2), np.float32)
    spike[T // 2, 0] = 1.0
    spike[T - 30, 1] = -0.75
    x = np.stack([loud_then_quiet, ramp_up, steady, spike])
    res = {}
    for prune in (0, 1):
        for segs in (0, 4):
            with M.Engine(4, 48000.0, M.METER_EBU | M.METER_TRUEPEAK, tune_layout=6, tune_prune=prune, tune_segments=segs) as e:
                e.integr_start()
                for a, b in ((0, 100000), (100000, T)):
                    e.process(x[:, a:b])
                res[prune, segs] = (e.truepeak(), e.out9(), e.prune_stats())
    for segs in (0, 4):
        assert np.array_equal(res[0, segs][0], res[1, segs][0])
        assert np.array_equal(res[0, segs][1], res[1, segs][1])
    considered, skipped = res[1, 0][2]
    assert considered > 0 and skipped > 0.2 * considered          # streams 0 and 3 are mostly prunable


def test_layout6_block_refinement_is_bit_identical(M):
    """tune_prune = 2: every 256-frame block is screened with the first f16 product and completed only when it can
    still hold the maximum.  Same peaks to the bit, on steady noise (where the tile-level bound never prunes), on a
    programme with level changes, on tiny and on non-finite samples, EBU + true peak and true peak alone."""
    import _signals as sig
    T = 48000 * 6
    steady = sig.lcg_noise(T, 17, 0.5)
    prog = (sig.lcg_noise(T, 18, 0.5) * (0.1 + 0.9 * np.abs(np.sin(np.arange(T, dtype=np.float32) / 20000.0)))[:, None]).astype(np.float32)
    tiny = (sig.lcg_noise(T, 19, 0.5) * np.float32(1e-30)).astype(np.float32)
    tiny[T // 2:] *= np.float32(1e-9)                                  # below 2^-97: the clamped-scale tiles
    sine = np.zeros((T, 2), np.float32)
    n = np.arange(T, dtype=np.float64)
    sine[:, 0] = (0.7 * np.sin(2 * np.pi * 997.0 / 48000.0 * n)).astype(np.float32)
    sine[:, 1] = (0.7 * np.sin(2 * np.pi * 11999.0 / 48000.0 * n + 0.3)).astype(np.float32)
    bad = sig.lcg_noise(T, 20, 0.25)
    bad[1000, 0] = np.nan
    bad[T - 5000, 1] = np.inf
    x = np.stack([steady, prog, tiny, sine, bad])
    for meters in (M.METER_EBU | M.METER_TRUEPEAK, M.METER_TRUEPEAK):
        res = {}
        for prune in (0, 2):
            for segs in (0, 3):
                with M.Engine(5, 48000.0, meters, tune_layout=6, tune_prune=prune, tune_segments=segs) as e:
                    if meters & M.METER_EBU:
                        e.integr_start()
                    for a, b in ((0, 77777), (77777, 77777 + 1023), (77777 + 1023, T)):
                        e.process(x[:, a:b])
                    res[prune, segs] = (e.truepeak(), e.out9() if meters & M.METER_EBU else None, e.refine_stats())
        for segs in (0, 3):
            assert np.array_equal(res[0, segs][0], res[2, segs][0], equal_nan=True), (meters, segs, res[0, segs][0], res[2, segs][0])
            if meters & M.METER_EBU:
                assert np.array_equal(res[0, segs][1], res[2, segs][1], equal_nan=True)
        screened, completed = res[2, 0][2]
        assert screened > 0 and completed < 0.5 * screened, (screened, completed)
        assert res[0, 0][2] == (0, 0)


def test_layout7_is_the_default_for_true_peak(M):
    # 7 = layout 6's kernel plus the lane = segment kernel for the calls that fit it (tests/test_gpu_seg.py)
    with M.Engine(1, 48000.0, M.METER_EBU | M.METER_TRUEPEAK) as e:
        assert e.layout() == 7
    with M.Engine(1, 48000.0, M.METER_TRUEPEAK) as e:
        assert e.layout() == 7
    with M.Engine(1, 48000.0, M.METER_TRUEPEAK, tune_prune=1) as e:
        assert e.layout() == 6                                   # pruning is layout 6's
    with M.Engine(1, 48000.0, M.METER_EBU) as e:
        assert e.layout() == 4
