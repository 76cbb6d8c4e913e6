"""The needle-meter restatements (oracle/mtr_oracle.c: IEC 268-10 type I / II PPM, M/S PPM, stereo correlation,
K-meter) against the reference's own objects (oracle/_ref, built from jmeters/*.cc where they lie): every
value read after every block must be BIT-IDENTICAL, at several block sizes and sample rates."""
import ctypes as C

import numpy as np
import pytest

import _signals as sig

pytestmark = pytest.mark.ref
F = C.c_float


class Ppm(C.Structure):
    _fields_ = [("z1", F), ("z2", F), ("m", F), ("res", C.c_int), ("w1", F), ("w2", F), ("w3", F), ("g", F)]


class MsPpm(C.Structure):
    _fields_ = [("p", Ppm), ("db", F), ("mv", F)]


class Stcorr(C.Structure):
    _fields_ = [("zl", F), ("zr", F), ("zlr", F), ("zll", F), ("zrr", F), ("w1", F), ("w2", F)]


class Kmeter(C.Structure):
    _fields_ = [("z1", F), ("z2", F), ("rms", F), ("peak", F), ("cnt", C.c_int), ("fpp", C.c_int), ("fall", F),
                ("flag", C.c_int), ("hold", C.c_int), ("fsamp", F), ("omega", F)]


def fp(a):
    return a.ctypes.data_as(C.POINTER(F))


def noise(n, seed, gain=1.0):
    x = sig.lcg_noise(n, seed, gain)
    env = (np.arange(n) % 20000 < 6000).astype(np.float32) * np.float32(0.9) + np.float32(0.1)   # bursts: attack and decay
    return (x[:, 0] * env).copy(), (x[:, 1] * env[::-1]).copy()


@pytest.fixture(scope="module")
def libs(oracle, reference):
    o, r = oracle.lib, reference.lib
    for f in ("ref_ppm_new", "ref_msppm_new", "ref_stcorr_new", "ref_kmeter_new"):
        getattr(r, f).restype = C.c_void_p
    r.ref_ppm_new.argtypes = [C.c_int, F]
    r.ref_msppm_new.argtypes = [F, F]
    r.ref_stcorr_new.argtypes = [C.c_int, F, F]
    r.ref_kmeter_new.argtypes = [F]
    for f in ("ref_ppm_read", "ref_msppm_read", "ref_stcorr_read"):
        getattr(r, f).restype = F
        getattr(r, f).argtypes = [C.c_void_p]
    r.ref_ppm_process.argtypes = [C.c_void_p, C.POINTER(F), C.c_int]
    r.ref_msppm_process.argtypes = [C.c_void_p, C.POINTER(F), C.POINTER(F), C.c_int, C.c_int]
    r.ref_msppm_set_gain.argtypes = [C.c_void_p, F]
    r.ref_stcorr_process.argtypes = [C.c_void_p, C.POINTER(F), C.POINTER(F), C.c_int]
    r.ref_kmeter_process.argtypes = [C.c_void_p, C.POINTER(F), C.c_int]
    r.ref_kmeter_read.argtypes = [C.c_void_p, C.POINTER(F), C.POINTER(F)]
    for f in ("ref_ppm_free", "ref_msppm_free", "ref_stcorr_free", "ref_kmeter_free", "ref_kmeter_reset"):
        getattr(r, f).argtypes = [C.c_void_p]
    for f in ("mo_ppm_read", "mo_msppm_read", "mo_stcorr_read"):
        getattr(o, f).restype = F
    o.mo_ppm_init_iec1.argtypes = o.mo_ppm_init_iec2.argtypes = [C.POINTER(Ppm), F]
    o.mo_ppm_process.argtypes = [C.POINTER(Ppm), C.POINTER(F), C.c_int]
    o.mo_ppm_read.argtypes = [C.POINTER(Ppm)]
    o.mo_msppm_init.argtypes = [C.POINTER(MsPpm), F, F]
    o.mo_msppm_set_gain.argtypes = [C.POINTER(MsPpm), F]
    o.mo_msppm_process.argtypes = [C.POINTER(MsPpm), C.POINTER(F), C.POINTER(F), C.c_int, C.c_int]
    o.mo_msppm_read.argtypes = [C.POINTER(MsPpm)]
    o.mo_stcorr_init.argtypes = [C.POINTER(Stcorr), C.c_int, F, F]
    o.mo_stcorr_process.argtypes = [C.POINTER(Stcorr), C.POINTER(F), C.POINTER(F), C.c_int]
    o.mo_stcorr_read.argtypes = [C.POINTER(Stcorr)]
    o.mo_kmeter_init.argtypes = [C.POINTER(Kmeter), F]
    o.mo_kmeter_process.argtypes = [C.POINTER(Kmeter), C.POINTER(F), C.c_int]
    o.mo_kmeter_read.argtypes = [C.POINTER(Kmeter), C.POINTER(F), C.POINTER(F)]
    o.mo_kmeter_reset.argtypes = [C.POINTER(Kmeter)]
    return o, r


BLOCKS = (64, 1000, 1024, 4099)


@pytest.mark.parametrize("kind", [1, 2])
@pytest.mark.parametrize("fs", [44100.0, 48000.0, 96000.0])
def test_iec_ppm(libs, kind, fs):
    o, r = libs
    x, _ = noise(60000, 5 + kind)
    for B in BLOCKS:
        h = r.ref_ppm_new(kind, fs)
        p = Ppm()
        (o.mo_ppm_init_iec1 if kind == 1 else o.mo_ppm_init_iec2)(C.byref(p), fs)
        for i, q in enumerate(range(0, x.size - B + 1, B)):
            blk = x[q:q + B].copy()
            r.ref_ppm_process(h, fp(blk), B)
            o.mo_ppm_process(C.byref(p), fp(blk), B)
            if i % 3 != 2:                                       # the maximum is held over unread blocks
                a, b = r.ref_ppm_read(h), o.mo_ppm_read(C.byref(p))
                assert np.float32(a).tobytes() == np.float32(b).tobytes(), (kind, fs, B, i, a, b)
        r.ref_ppm_free(h)


def test_ms_ppm(libs):
    o, r = libs
    fs = 48000.0
    xl, xr = noise(60000, 9)
    for B in BLOCKS:
        for side, db in ((0, -6.0), (1, -6.0), (1, 14.0)):
            h = r.ref_msppm_new(fs, -6.0)
            p = MsPpm()
            o.mo_msppm_init(C.byref(p), fs, -6.0)
            for i, q in enumerate(range(0, xl.size - B + 1, B)):
                if i == 5:                                       # the "+20 dB" switch of the S needle (bbcm_run)
                    r.ref_msppm_set_gain(h, db)
                    o.mo_msppm_set_gain(C.byref(p), db)
                bl, br = xl[q:q + B].copy(), xr[q:q + B].copy()
                r.ref_msppm_process(h, fp(bl), fp(br), B, side)
                o.mo_msppm_process(C.byref(p), fp(bl), fp(br), B, side)
                a, b = r.ref_msppm_read(h), o.mo_msppm_read(C.byref(p))
                assert np.float32(a).tobytes() == np.float32(b).tobytes(), (B, side, db, i, a, b)
            r.ref_msppm_free(h)


@pytest.mark.parametrize("fs", [44100, 48000, 96000])
def test_stereo_correlation(libs, fs):
    o, r = libs
    xl, xr = noise(50000, 13)
    xr[:20000] = xl[:20000]                                      # +1, then decorrelated, then inverted
    xr[35000:] = -xl[35000:]
    xl[40000], xr[40001] = np.inf, np.nan                        # the non-finite scrub at block ends
    for B in BLOCKS:
        h = r.ref_stcorr_new(fs, 2e3, 0.3)
        c = Stcorr()
        o.mo_stcorr_init(C.byref(c), fs, 2e3, 0.3)
        for i, q in enumerate(range(0, xl.size - B + 1, B)):
            bl, br = xl[q:q + B].copy(), xr[q:q + B].copy()
            r.ref_stcorr_process(h, fp(bl), fp(br), B)
            o.mo_stcorr_process(C.byref(c), fp(bl), fp(br), B)
            a, b = r.ref_stcorr_read(h), o.mo_stcorr_read(C.byref(c))
            assert np.float32(a).tobytes() == np.float32(b).tobytes(), (fs, B, i, a, b)
        r.ref_stcorr_free(h)


@pytest.mark.parametrize("fs", [44100.0, 48000.0, 96000.0])
def test_kmeter(libs, fs):
    o, r = libs
    x, _ = noise(120000, 17, 0.8)
    x[70000:] *= np.float32(0.05)                                # quiet tail: peak hold (0.5 s) then fallback
    for B in BLOCKS:
        h = r.ref_kmeter_new(fs)
        k = Kmeter()
        o.mo_kmeter_init(C.byref(k), fs)
        ra, rp, oa, op = F(), F(), F(), F()
        for i, q in enumerate(range(0, x.size - B + 1, B)):
            n = B if i % 7 else B - 3                            # a change of period size recomputes the fallback factor
            blk = x[q:q + n].copy()
            r.ref_kmeter_process(h, fp(blk), n)
            o.mo_kmeter_process(C.byref(k), fp(blk), n)
            if i % 2 == 0:
                r.ref_kmeter_read(h, C.byref(ra), C.byref(rp))
                o.mo_kmeter_read(C.byref(k), C.byref(oa), C.byref(op))
                assert np.float32(ra.value).tobytes() == np.float32(oa.value).tobytes(), (fs, B, i, ra.value, oa.value)
                assert np.float32(rp.value).tobytes() == np.float32(op.value).tobytes(), (fs, B, i, rp.value, op.value)
            if i == 40:
                r.ref_kmeter_reset(h)
                o.mo_kmeter_reset(C.byref(k))
        r.ref_kmeter_free(h)
