"""Every fused-kernel variant (layout 1: wave per segment; 2: wave-specialised; 3: wave-specialised with
rotating roles; dense vs mirror-symmetric interpolator) must give the same record, and each must match
the oracle within the tolerances of tests/test_gpu_parity.py."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_golden import tri_noise  # noqa: E402

VARIANTS = [dict(tune_layout=1, tune_run=13), dict(tune_layout=1, tune_run=39), dict(tune_layout=2),
            dict(tune_layout=2, tune_fir=1), dict(tune_layout=3)]


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
