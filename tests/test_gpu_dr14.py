"""DR-14 for a batch of tracks (MTR_METER_DR14, mtr_dr14.hip) against the restatement of src/dr14.c
(oracle mo_dr14_run, pinned to the LV2 plugin's behaviour by tests/test_lv2_dr14.py).

The reference adds the squares of a 3 s window sequentially in f32; the kernel reduces in double, so a
window's RMS can fall into the neighbouring 0.01 dB histogram bin: scores agree to +-0.01 dB (one bin; DR_TOL below), the window
count and the peak are exact."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
DR_TOL = 0.01 + 1e-5      # the contract: one bin of the reference's 0.01 dB histogram (measured worst case in this file: 0.01)


class Ports(C.Structure):
    _fields_ = [("v_rms", C.c_float * 2), ("v_peak", C.c_float * 2), ("m_rms", C.c_float * 2), ("m_peak", C.c_float * 2),
                ("dr", C.c_float * 2), ("dr_total", C.c_float), ("block_count", C.c_float)]


@pytest.fixture(scope="module")
def M():
    import meters.lv2_amd as m
    return m


def programme(n, seed, fs):
    """Noise under a slow envelope with a silent stretch (dropped windows) and a few loud hits (the peaks)."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(-1, 1, (n, 2)).astype(np.float32)
    t = np.arange(n) / fs
    env = (0.02 + 0.5 * (0.5 + 0.5 * np.sin(2 * np.pi * t / 11.0 + seed))).astype(np.float32)
    env[int(6.5 * fs):int(10.2 * fs)] = 0.0                     # more than one whole silent window
    x *= env[:, None]
    for k in range(5):
        x[(seed * 7919 + k * 104729) % n, k & 1] = np.float32(0.9 - 0.1 * k)
    x[:, 1] *= np.float32(0.5)
    return x


def ref_dr14(oracle, x, fs, chn, calls):
    lib = oracle.lib
    lib.mo_dr14_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double]
    lib.mo_dr14_run.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(Ports)]
    state = C.create_string_buffer(1 << 17)
    lib.mo_dr14_init(state, chn, 1, float(fs))
    want = Ports()
    pos = 0
    for n in calls:
        for q in range(pos, pos + n, 8192):                     # TruePeakdsp::process asserts n <= 8192
            m = min(8192, pos + n - q)
            chans = [np.ascontiguousarray(x[q:q + m, c]) for c in range(chn)]
            ptrs = (C.c_void_p * 2)(*[ch.ctypes.data for ch in chans], *([None] * (2 - chn)))
            lib.mo_dr14_run(state, ptrs, m, C.byref(want))
        pos += n
    return want


@pytest.mark.parametrize("fs", [48000.0, 44100.0])
@pytest.mark.parametrize("chn", [2, 1])
def test_dr14_batch_matches_the_restatement(M, oracle, fs, chn):
    T = int(fs * 31) + 17
    calls = [int(fs * 2.5), int(fs * 3) + 1, 1, int(fs * 12), T - int(fs * 2.5) - int(fs * 3) - 2 - int(fs * 12)]
    S = 3
    x = np.stack([programme(T, 40 + s, fs) for s in range(S)])
    if chn == 1:
        x = x[:, :, :1]
    with M.Engine(S, fs, M.METER_DR14, n_channels=chn) as e:
        pos = 0
        for n in calls:
            e.process(x[:, pos:pos + n] if chn == 2 else x[:, pos:pos + n, 0])
            pos += n
        got = e.dr14()
        for s in range(S):
            want = ref_dr14(oracle, x[s], fs, chn, calls)
            assert got[s].block_count == want.block_count, (s, got[s].block_count, want.block_count)
            assert want.block_count >= 3 * 8                    # ten windows, at least one of them silent
            for c in range(chn):
                assert abs(got[s].m_rms[c] - want.m_rms[c]) <= DR_TOL, (s, c, got[s].m_rms[c], want.m_rms[c])
                # (the plugin's m_peak PORT shows the true-peak maximum; the engine's m_peak is the second-highest
                # window peak that enters dr — checked through dr)
                assert abs(got[s].dr[c] - want.dr[c]) <= DR_TOL, (s, c, got[s].dr[c], want.dr[c])
            if chn == 2:
                assert abs(got[s].dr_total - want.dr_total) <= DR_TOL, s
        # reset_peaks
        e.dr14_reset()
        r = e.dr14()[0]
        assert r.block_count == 0 and r.dr[0] == 21 and r.m_rms[0] == -81


def test_dr14_batch_beside_the_other_meters(M, oracle):
    """DR-14 rides in the same engine as EBU R128 + true peak; neither disturbs the other."""
    fs, T = 48000.0, 48000 * 10
    x = np.stack([programme(T, 90 + s, fs) for s in range(2)])
    with M.Engine(2, fs, M.METER_EBU | M.METER_TRUEPEAK | M.METER_DR14) as e, M.Engine(2, fs, M.METER_EBU | M.METER_TRUEPEAK) as e0:
        for eng in (e, e0):
            eng.integr_start()
            eng.process(x)
        assert np.array_equal(e.out9(), e0.out9()) and np.array_equal(e.truepeak(), e0.truepeak())
        got = e.dr14()
    for s in range(2):
        want = ref_dr14(oracle, x[s], fs, 2, [T])
        assert got[s].block_count == want.block_count and want.block_count >= 3 * 2
        for c in range(2):
            assert abs(got[s].m_rms[c] - want.m_rms[c]) <= DR_TOL or (got[s].m_rms[c] == want.m_rms[c])


def test_dr14_known_answers(M):
    """Hand-derived: the DR-14 RMS is sqrt (2 mean x^2), so a full-scale sine scores 0 dB against a peak of
    0 dB (DR clamps at 1) and uniform noise of amplitude a scores 20 log10 sqrt (2/3) = -1.76 dB below its
    peak; nothing is reported before the third window."""
    fs, T = 48000.0, 48000 * 13
    t = np.arange(T) / fs
    sine = np.sin(2 * np.pi * 997.0 * t).astype(np.float32)
    rng = np.random.default_rng(5)
    noise = (0.5 * rng.uniform(-1, 1, T)).astype(np.float32)
    x = np.stack([np.stack([sine, sine], 1), np.stack([noise, noise], 1)])
    with M.Engine(2, fs, M.METER_DR14) as e:
        e.process(x[:, :int(fs * 6.5)])
        early = e.dr14()
        assert early[0].block_count == 6 and early[0].dr[0] == 21      # two windows: not enough yet
        e.process(x[:, int(fs * 6.5):])
        r = e.dr14()
    assert r[0].block_count == 12
    assert abs(r[0].m_rms[0]) <= 0.011 and abs(r[0].m_peak[0]) <= 1e-3 and r[0].dr[0] == 1.0 and r[0].dr_total == 1.0
    want = -20 * np.log10(np.sqrt(2.0 / 3.0))
    assert abs(r[1].dr[0] - want) <= 0.03 and abs(r[1].dr_total - want) <= 0.03, (r[1].dr[0], want)
