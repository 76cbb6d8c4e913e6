"""dr14 / TPnRMS plugins through the LV2 ABI against the oracle's restatement of src/dr14.c (mo_dr14_*).  The plugin
is a thin client of the batch engine (TPBALLIST + KMETER + DR14 meters, all on the GPU), so every port carries
a stated tolerance:
  * true-peak bars (TruePeakdsp::process):       2e-5 dB
  * K-meter RMS bar (Kmeterdsp, f32 two-pole):   2e-4 dB (the engine sums the squares in double: 1e-5 relative)
  * K-meter held peak (TPnRMS m_rms port):       1e-5 relative in dB (an ulp of the fall-back factor)
  * DR-14 score, DR per channel, DR total:       0.01 dB = the contract = ONE bin of the reference's 0.01 dB histogram:
    a window's RMS is binned after summing 144 001 squares — sequentially in f32 in the reference, in double on
    the GPU — and the two sums differ by ~2e-5 relative (1e-4 dB), which moves a window across a bin edge once
    in ~100 windows; the score is an average over bins, so it moves by at most that one bin.  Measured here: 0.
  * window count: exact."""
import ctypes as C

import numpy as np
import pytest

import _signals as sig
from _lv2host import Host, Instance, MTR_URI, forge_object, forge_sequence

pytestmark = pytest.mark.gpu
K = MTR_URI
F = C.c_float
DR_TOL = 0.01 + 1e-5        # one 0.01 dB histogram bin (module docstring), plus float slack


class Ports(C.Structure):
    _fields_ = [("v_rms", F * 2), ("v_peak", F * 2), ("m_rms", F * 2), ("m_peak", F * 2), ("dr", F * 2),
                ("dr_total", F), ("block_count", F)]


@pytest.fixture(scope="module")
def host():
    return Host()


def _f(v=0.0):
    return np.array([v], np.float32)


def programme(n, seed):
    """Bursts of different level every 1.5 s so that the 3 s windows differ; power-of-two gains only."""
    x = sig.lcg_noise(n, seed, 1.0)
    lev = np.array([0.5, 0.0625, 0.25, 0.5, 0.015625, 0.125, 0.25, 0.03125], np.float32)
    env = lev[(np.arange(n) // 72000) % lev.size]
    return (x[:, 0] * env).astype(np.float32), (x[:, 1] * env * np.float32(0.5)).astype(np.float32)


@pytest.mark.parametrize("name,chn,dr_mode", [("dr14stereo", 2, 1), ("dr14mono", 1, 1), ("TPnRMSstereo", 2, 0), ("TPnRMSmono", 1, 0)])
def test_dr14_against_the_restatement(host, oracle, name, chn, dr_mode):
    lib = oracle.lib
    lib.mo_dr14_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double]
    lib.mo_dr14_reset.argtypes = [C.c_void_p]
    lib.mo_dr14_run.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(Ports)]
    state = C.create_string_buffer(1 << 17)                   # sizeof (mo_dr14) is ~ 65 KB
    lib.mo_dr14_init(state, chn, dr_mode, 48000.0)

    B, fs = 1024, 48000.0
    xl, xr = programme(int(fs * 14), 77)
    inst = Instance(host, name, rate=fs)
    assert inst.ok()
    follow, reset, blk = _f(1.0), _f(0.0), _f()
    ports = {k: [_f() for _ in range(2)] for k in ("v_peak", "m_peak", "v_rms", "m_rms", "dr")}
    total = _f()
    inst.connect(1, follow); inst.connect(2, reset); inst.connect(3, blk)
    for c in range(chn):
        b = 4 + 7 * c
        for off, k in ((2, "v_peak"), (3, "m_peak"), (4, "v_rms"), (5, "m_rms"), (6, "dr")):
            inst.connect(b + off, ports[k][c])
    if chn == 2:
        inst.connect(18, total)
    empty = forge_sequence(host, [])
    want = Ports()
    n_win, worst = 0, 0.0
    for i, q in enumerate(range(0, xl.size - B + 1, B)):
        chans = [xl[q:q + B].copy(), xr[q:q + B].copy()][:chn]
        ctl = empty
        if i == 300:                                          # the GUI's reset message
            ctl = forge_sequence(host, [forge_object(host, K + "dr14reset", [])])
            lib.mo_dr14_reset(state)
        reset[0] = 1.0 if i == 500 else 0.0                   # the reset button (a control port)
        if i == 500:
            lib.mo_dr14_reset(state)
        inst.connect(0, ctl)
        for c in range(chn):
            inst.connect(4 + 7 * c, chans[c]); inst.connect(5 + 7 * c, chans[c])
        inst.run(B)
        ptrs = (C.c_void_p * 2)(*[ch.ctypes.data for ch in chans], *([None] * (2 - chn)))
        lib.mo_dr14_run(state, ptrs, B, C.byref(want))
        for c in range(chn):
            assert abs(ports["v_rms"][c][0] - want.v_rms[c]) <= 2e-4, (i, c, ports["v_rms"][c][0], want.v_rms[c])
            assert abs(ports["v_peak"][c][0] - want.v_peak[c]) <= 2e-5 * max(1.0, abs(want.v_peak[c])), (i, c, ports["v_peak"][c][0], want.v_peak[c])
            assert abs(ports["m_peak"][c][0] - want.m_peak[c]) <= 2e-5 * max(1.0, abs(want.m_peak[c])), (i, c)
            if dr_mode:
                worst = max(worst, abs(ports["m_rms"][c][0] - want.m_rms[c]), abs(ports["dr"][c][0] - want.dr[c]))
                assert abs(ports["m_rms"][c][0] - want.m_rms[c]) <= DR_TOL, (i, c, ports["m_rms"][c][0], want.m_rms[c])
                assert abs(ports["dr"][c][0] - want.dr[c]) <= DR_TOL, (i, c, ports["dr"][c][0], want.dr[c])
            else:
                assert abs(ports["m_rms"][c][0] - want.m_rms[c]) <= 1e-5 * max(1.0, abs(want.m_rms[c])), (i, c, ports["m_rms"][c][0], want.m_rms[c])
        if chn == 2 and dr_mode:
            assert abs(total[0] - want.dr_total) <= DR_TOL, i
        assert blk[0] == np.float32(want.block_count), i
        n_win = max(n_win, int(want.block_count) // 3)
    if dr_mode:
        assert n_win >= 2                                     # windows were completed between the resets
        print("worst DR-14 port deviation [dB]:", worst)
    # the GUI attaching: marker values that force a change on the ports (src/dr14.c:455-466)
    inst.connect(0, forge_sequence(host, [forge_object(host, K + "meteron", [])]))
    inst.run(B)
    assert blk[0] <= -1 and ports["m_peak"][0][0] == -100 and ports["m_rms"][0][0] == -100
    inst.cleanup()
