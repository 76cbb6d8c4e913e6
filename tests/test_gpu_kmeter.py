"""Kmeterdsp for a batch (MTR_METER_KMETER, mtr_kmeter.hip) against the restatement of jmeters/kmeterdsp.cc
(oracle mo_kmeter_*, itself bit-identical to the reference object: tests/test_needle_oracle_vs_ref.py).

One engine call = one Kmeterdsp::process () per channel.  The filter sums are re-associated (pieces of 4096
frames combined with powers of the transition matrix): 1e-5 relative on the RMS; the digital peak and its
hold / fall-back bookkeeping are exact."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
F = C.c_float


class Kmeter(C.Structure):
    _fields_ = [("z1", F), ("z2", F), ("rms", F), ("peak", F), ("cnt", C.c_int), ("fpp", C.c_int), ("fall", F),
                ("flag", C.c_int), ("hold", C.c_int), ("fsamp", F), ("omega", F)]


@pytest.fixture(scope="module")
def M():
    import meters.lv2_amd as m
    return m


def signal(n, seed, fs):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / fs
    env = (0.05 + 0.6 * (0.5 + 0.5 * np.sin(2 * np.pi * t * 0.7 + seed))).astype(np.float32)
    x = (rng.uniform(-1, 1, (n, 2)).astype(np.float32) * env[:, None])
    x[n // 3, 0] = 0.97                                         # a peak that is then held and falls back
    x[:, 1] *= np.float32(0.25)
    return x


@pytest.mark.parametrize("fs", [48000.0, 44100.0])
@pytest.mark.parametrize("chn", [2, 1])
def test_kmeter_batch_matches_the_restatement(M, oracle, fs, chn):
    lib = oracle.lib
    lib.mo_kmeter_init.argtypes = [C.POINTER(Kmeter), F]
    lib.mo_kmeter_process.argtypes = [C.POINTER(Kmeter), C.POINTER(F), C.c_int]
    lib.mo_kmeter_read.argtypes = [C.POINTER(Kmeter), C.POINTER(F), C.POINTER(F)]
    S = 5
    # call sizes: below one piece, not a multiple of four, several pieces, a repeat (fall-back multiplier
    # kept), a one-frame call (no group at all), a long one
    calls = [1024, 1023, 4096 * 3 + 6, 4096 * 3 + 6, 1, 3, int(fs * 2) + 1, 512, 512, 512]
    T = sum(calls)
    x = np.stack([signal(T, 300 + s, fs) for s in range(S)])
    read_after = {2, 3, 6, 9}                                   # the host reads now and then, not after every call
    with M.Engine(S, fs, M.METER_KMETER, n_channels=chn) as e:
        ks = [[Kmeter() for _ in range(chn)] for _ in range(S)]
        for s in range(S):
            for c in range(chn):
                lib.mo_kmeter_init(C.byref(ks[s][c]), fs)
        pos = 0
        for i, n in enumerate(calls):
            blk = x[:, pos:pos + n]
            e.process(blk if chn == 2 else np.ascontiguousarray(blk[:, :, 0]))
            for s in range(S):
                for c in range(chn):
                    ch = np.ascontiguousarray(blk[s, :, c])
                    lib.mo_kmeter_process(C.byref(ks[s][c]), ch.ctypes.data_as(C.POINTER(F)), n)
            pos += n
            if i in read_after:
                rms, peak = e.kmeter_read()
                for s in range(S):
                    for c in range(chn):
                        a, b = F(), F()
                        lib.mo_kmeter_read(C.byref(ks[s][c]), C.byref(a), C.byref(b))
                        assert abs(rms[s, c] - a.value) <= 1e-5 * max(a.value, 1e-3), (i, s, c, rms[s, c], a.value)
                        assert peak[s, c] == np.float32(b.value), (i, s, c, peak[s, c], b.value)
        e.kmeter_reset()
        rms, peak = e.kmeter_read()
        assert not rms.any() and not peak.any()


def test_kmeter_known_answer(M):
    """A full-scale sine reads 1.0 = 0 dB on a K-meter's RMS scale (the detector's sqrt (2 z2)) and peak 1.0."""
    fs, T = 48000.0, 48000 * 2
    t = np.arange(T) / fs
    x = np.sin(2 * np.pi * 1000.0 * t).astype(np.float32)
    with M.Engine(1, fs, M.METER_KMETER, n_channels=1) as e:
        e.process(x[None, :])
        rms, peak = e.kmeter_read()
    assert abs(rms[0, 0] - 1.0) < 1e-3 and abs(peak[0, 0] - 1.0) < 1e-4
