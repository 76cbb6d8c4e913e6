"""Layout 5 (mtr_fused3.hip): K-weighting as k_kw + the 4x interpolator on the matrix pipe with the samples
split into two f16 halves.  Loudness must be what the other layouts give; true peaks may differ from the
f32 interpolator by the tap rounding only: bound 2^-12 * L1 = 0.0055 dB, stated tolerance +-0.01 dB."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_golden import tri_noise  # noqa: E402

DB_BOUND = 0.0056      # worst case of f16 taps (mtr_mfma_fir.h); the parity clause allows 0.01


@pytest.fixture(scope="module")
def M():
    import meters.lv2_amd as m
    return m


def _db(a, b):
    a = np.maximum(np.asarray(a, np.float64), 1e-30)
    b = np.maximum(np.asarray(b, np.float64), 1e-30)
    return 20 * np.log10(a / b)


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
        return o9, e.truepeak(), np.stack(per_call)


@pytest.mark.parametrize("fs", [48000.0, 44100.0, 96000.0])
@pytest.mark.parametrize("segs", [0, 3])
@pytest.mark.parametrize("run", [39, 19])
def test_mfma_layout_matches_f32_layout_and_oracle(M, oracle, fs, segs, run):
    T = int(fs) * 6 + 1
    calls = [1001, int(fs) * 3, 47, T - 1001 - int(fs) * 3 - 47]
    x = np.stack([tri_noise(T, 500 + s, 2.0 ** -(s % 3), period=72000) for s in range(3)])
    o3, p3, c3 = _run(M, x, calls, fs, tune_segments=segs)
    o5, p5, c5 = _run(M, x, calls, fs, tune_segments=segs, tune_layout=5, tune_run=run)
    # loudness: the K-filter is the same arithmetic in every layout
    assert np.allclose(o5[:, :4], o3[:, :4], atol=1e-3)
    assert np.all(np.abs(o5[:, 4] - o3[:, 4]) <= 0.01)
    # peaks: against the exact-f32 layout, per call and held, and against the oracle
    assert np.all(np.abs(_db(c5, c3)) <= DB_BOUND), np.abs(_db(c5, c3)).max()
    assert np.all(np.abs(_db(p5, p3)) <= DB_BOUND)
    for s in range(3):
        tp = oracle.tp(x[s], fs, 8192)
        assert np.all(np.abs(_db(p5[s], tp)) <= 0.01), (s, p5[s], tp)


def test_mfma_layout_edge_signals(M, oracle):
    """Impulses next to tile and call boundaries, a full-scale +1/-1 pattern (the worst inter-sample peak),
    silence, a very quiet stream (f16 subnormal halves) and a hot one (|x| up to 8)."""
    import _signals as sig
    T = 48000 * 3
    spike = np.zeros((T, 2), np.float32)
    spike[2399, 0] = 1.0; spike[2400, 1] = -1.0; spike[T - 1, 0] = 0.5; spike[50000, 1] = 0.25
    g3 = np.tile(np.array([1, 1, -1, -1], np.float32), T // 4)[:, None].repeat(2, 1)
    quiet = sig.lcg_noise(T, 11, 1.0) * np.float32(2.0 ** -20)
    hot = sig.lcg_noise(T, 12, 1.0) * np.float32(8.0)
    x = np.stack([spike, g3, np.zeros((T, 2), np.float32), quiet.astype(np.float32), hot.astype(np.float32)])
    calls = [2399, 1, 100000, T - 102400]
    _, p3, c3 = _run(M, x, calls, tune_segments=0)
    _, p5, c5 = _run(M, x, calls, tune_segments=0, tune_layout=5)
    assert np.all(p5[2] == 0.0)
    nz = [0, 1, 3, 4]
    assert np.all(np.abs(_db(p5[nz], p3[nz])) <= DB_BOUND), _db(p5[nz], p3[nz])
    m = c3 > 0
    assert np.all((c5 > 0) == m)
    assert np.all(np.abs(_db(c5[m], c3[m])) <= DB_BOUND)
    for s in nz:
        assert np.all(np.abs(_db(p5[s], oracle.tp(x[s], 48000.0, 8192))) <= 0.01), s


def test_mfma_layout_truepeak_only(M, oracle):
    T = 48000 * 2 + 333
    x = np.stack([tri_noise(T, 900 + s, 0.5, period=30000) for s in range(5)])
    _, p5, _ = _run(M, x, [T], meters=M.METER_TRUEPEAK, tune_layout=5)
    for s in range(5):
        assert np.all(np.abs(_db(p5[s], oracle.tp(x[s], 48000.0, 8192))) <= 0.01), s


def test_mfma_layout_exact_pruning_changes_nothing_but_time(M):
    """tune_prune in layout 5 skips the products of tiles whose L1 * max|x| cannot beat the running peak: the
    peaks must be bit-identical to the unpruned layout-5 run, on signals that prune a lot and nothing."""
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
            with M.Engine(4, 48000.0, M.METER_EBU | M.METER_TRUEPEAK, tune_layout=5, tune_prune=prune, tune_segments=segs) as e:
                e.integr_start()
                for a, b in ((0, 100000), (100000, T)):
                    e.process(x[:, a:b])
                res[prune, segs] = (e.truepeak(), e.out9(), e.prune_stats())
    for segs in (0, 4):
        assert np.array_equal(res[0, segs][0], res[1, segs][0])
        assert np.array_equal(res[0, segs][1], res[1, segs][1])
    considered, skipped = res[1, 0][2]
    assert considered > 0 and skipped > 0.2 * considered          # streams 0 and 3 are mostly prunable
