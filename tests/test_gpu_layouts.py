"""Every K-weighting + true-peak kernel the library ships — layout 3 (the exact-f32 VALU interpolator, mirror-symmetric
and dense form), layout 6 (matrix pipe at f32 grade, wave per segment) and layout 7 (the same with lane = segment for the
calls that fit it) — must give the same record, and each must match the oracle within the tolerances of
tests/test_gpu_parity.py.  Layouts 1, 2 and 5 of rounds 1-2 are gone: asking for them is an argument error."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_golden import tri_noise  # noqa: E402

VARIANTS = [dict(tune_layout=3), dict(tune_layout=3, tune_fir=1), dict(tune_layout=6), dict(tune_layout=7)]


def test_retired_layouts_are_argument_errors(M):
    for kw in (dict(tune_layout=1), dict(tune_layout=2), dict(tune_layout=5), dict(tune_layout=8), dict(tune_fir=3),
               dict(tune_run=13)):
        with pytest.raises(M.EngineError):
            M.Engine(1, 48000.0, M.METER_EBU | M.METER_TRUEPEAK, **kw)


@pytest.fixture(scope="module")
def M():
    import meters.lv2_amd as m
    return m


def _run(M, x, calls, **kw):
    with M.Engine(x.shape[0], 48000.0, M.METER_EBU | M.METER_TRUEPEAK, **kw) as e:
        e.integr_start()
        pos, frags = 0, []
        for n in calls:
            e.process(x[:, pos:pos + n])
            frags.append(e.fragment_powers())
            pos += n
        hm, hs = e.histograms()
        return e.out9(), e.truepeak(), np.concatenate(frags, 1), hm, hs


@pytest.mark.parametrize("segs", [0, 5])
@pytest.mark.parametrize("calls", [[48000 * 7], [1001, 48000 * 3, 47, 48000 * 4 - 1048]])
def test_variants_agree_and_match_oracle(M, oracle, segs, calls):
    T = sum(calls)
    x = np.stack([tri_noise(T, 200 + s, 2.0 ** -(s % 3), period=72000) for s in range(3)])
    ref = [oracle.ebu(x[s], 48000.0, 2400, want_frag=True) for s in range(3)]
    tp = [oracle.tp(x[s], 48000.0, 8192) for s in range(3)]
    for kw in VARIANTS:
        o9, pk, fr, hm, hs = _run(M, x, calls, tune_segments=segs, **kw)
        for s in range(3):
            assert np.allclose(fr[s], ref[s]["frag_power"], rtol=2e-5), (kw, s)
            assert np.allclose(o9[s, :4], ref[s]["out9"][:4], atol=1e-3), (kw, s)
            assert abs(o9[s, 4] - ref[s]["out9"][4]) <= 0.01, (kw, s)
            assert np.allclose(pk[s], tp[s], rtol=2e-6), (kw, s)
            assert np.abs(hm[s] - ref[s]["hist_M"]).sum() <= 4, (kw, s)


@pytest.mark.parametrize("fs", [48000.0, 44100.0])
@pytest.mark.parametrize("segs", [0, 3])
def test_kweighting_only_kernel(M, oracle, fs, segs):
    """EBU without true peak runs k_kw (layout 4: one wave per segment, single LDS buffer, DPP scan), with
    39- or 19-frame lane runs; same record as the fused kernel's K-filter role and as the oracle.  44.1 kHz
    fragments are 2205 frames, so tiles start on odd frames; odd-sized calls end on odd frames."""
    T = int(fs) * 6 + 1
    calls = [1001, int(fs) * 3, 47, T - 1001 - int(fs) * 3 - 47]
    x = np.stack([tri_noise(T, 300 + s, 2.0 ** -(s % 3), period=72000) for s in range(3)])
    ref = [oracle.ebu(x[s], fs, 2400, want_frag=True) for s in range(3)]
    for kw in (dict(tune_layout=3), dict(tune_layout=4), dict(tune_layout=4, tune_run=19), dict()):
        with M.Engine(3, fs, M.METER_EBU, tune_segments=segs, **kw) as e:
            e.integr_start()
            pos, frags = 0, []
            for n in calls:
                e.process(x[:, pos:pos + n])
                frags.append(e.fragment_powers())
                pos += n
            o9, fr = e.out9(), np.concatenate(frags, 1)
            hm, _ = e.histograms()
        for s in range(3):
            assert np.allclose(fr[s], ref[s]["frag_power"], rtol=2e-5), (kw, s)
            assert np.allclose(o9[s, :4], ref[s]["out9"][:4], atol=1e-3), (kw, s)
            assert abs(o9[s, 4] - ref[s]["out9"][4]) <= 0.01, (kw, s)
            assert np.abs(hm[s] - ref[s]["hist_M"]).sum() <= 4, (kw, s)


def test_exact_peak_pruning_changes_nothing_but_time(M, oracle):
    """tune_prune skips interpolator tiles whose L1 * max|x| bound cannot beat the running peak:
    peaks must be bit-identical to the dense run, on signals that prune a lot and that prune nothing."""
    import _signals as sig
    T = 48000 * 8
    loud_then_quiet = sig.lcg_noise(T, 5, 0.5)
    loud_then_quiet[48000:] *= np.float32(0.125)             # everything after 1 s is 18 dB down: prunable
    ramp_up = sig.lcg_noise(T, 6, 0.5) * np.linspace(0.05, 1.0, T, dtype=np.float32)[:, None]   # peak keeps rising
    steady = sig.lcg_noise(T, 7, 0.5)                        # stationary: nothing to prune
    spike = np.zeros((T, 2), np.float32)
    spike[T // 2, 0] = 1.0                                   # one impulse in silence, mid-tile
    spike[T - 30, 1] = -0.75                                 # and one whose ringing crosses the last tiles
    x = np.stack([loud_then_quiet, ramp_up.astype(np.float32), steady, spike])
    res = {}
    for prune in (0, 1):
        for segs in (0, 4):
            # (layout 6 for both: pruning is its feature, and with tune_segments the dense engine would otherwise take k_seg)
            with M.Engine(4, 48000.0, M.METER_EBU | M.METER_TRUEPEAK, tune_prune=prune, tune_segments=segs, tune_layout=6) as e:
                e.integr_start()
                for a, b in ((0, 100000), (100000, T)):          # two calls: history + running state carry over
                    e.process(x[:, a:b])
                res[prune, segs] = (e.truepeak(), e.out9(), e.prune_stats())
    for segs in (0, 4):
        assert np.array_equal(res[0, segs][0], res[1, segs][0])           # identical peaks, bit for bit
        assert np.array_equal(res[0, segs][1], res[1, segs][1])           # loudness untouched
        done, skipped = res[1, segs][2]
        assert done > 0 and skipped > 0.2 * done, (done, skipped)
        assert res[0, segs][2] == (0, 0)
    for s in range(4):
        assert np.allclose(res[1, 0][0][s], oracle.tp(x[s], 48000.0, 8192), rtol=2e-6), s
