"""Needle meters where /root/reference does not exist: the oracle (and, through the LV2 ABI, the plugins of
lib/meters_amd.so — CPU plumbing like VU, no GPU involved) against tests/golden/golden_needle_v1.npz,
which tests/golden/make_golden_needle.py generated from the reference build.  Bit-exact."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_golden_needle import B, signal  # noqa: E402
from test_needle_oracle_vs_ref import Kmeter, MsPpm, Ppm, Stcorr, fp  # noqa: E402

F = C.c_float
G = np.load(os.path.join(HERE, "golden", "golden_needle_v1.npz"))
RATES = (44100.0, 48000.0, 96000.0)


def bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def o(oracle):
    lib = oracle.lib
    for f in ("mo_ppm_read", "mo_msppm_read", "mo_stcorr_read"):
        getattr(lib, f).restype = F
    lib.mo_ppm_init_iec1.argtypes = lib.mo_ppm_init_iec2.argtypes = [C.POINTER(Ppm), F]
    lib.mo_ppm_process.argtypes = [C.POINTER(Ppm), C.POINTER(F), C.c_int]
    lib.mo_ppm_read.argtypes = [C.POINTER(Ppm)]
    lib.mo_msppm_init.argtypes = [C.POINTER(MsPpm), F, F]
    lib.mo_msppm_process.argtypes = [C.POINTER(MsPpm), C.POINTER(F), C.POINTER(F), C.c_int, C.c_int]
    lib.mo_msppm_read.argtypes = [C.POINTER(MsPpm)]
    lib.mo_stcorr_init.argtypes = [C.POINTER(Stcorr), C.c_int, F, F]
    lib.mo_stcorr_process.argtypes = [C.POINTER(Stcorr), C.POINTER(F), C.POINTER(F), C.c_int]
    lib.mo_stcorr_read.argtypes = [C.POINTER(Stcorr)]
    lib.mo_kmeter_init.argtypes = [C.POINTER(Kmeter), F]
    lib.mo_kmeter_process.argtypes = [C.POINTER(Kmeter), C.POINTER(F), C.c_int]
    lib.mo_kmeter_read.argtypes = [C.POINTER(Kmeter), C.POINTER(F), C.POINTER(F)]
    return lib


@pytest.mark.parametrize("fs", RATES)
def test_oracle_matches_the_reference_vectors(o, fs):
    xl, xr = signal()
    tag = str(int(fs))
    blocks = range(0, xl.size - B + 1, B)
    for kind in (1, 2):
        p = Ppm()
        (o.mo_ppm_init_iec1 if kind == 1 else o.mo_ppm_init_iec2)(C.byref(p), fs)
        seq = []
        for q in blocks:
            blk = xl[q:q + B].copy()
            o.mo_ppm_process(C.byref(p), fp(blk), B)
            seq.append(o.mo_ppm_read(C.byref(p)))
        assert np.array_equal(bits(seq), bits(G[f"iec{kind}_{tag}"])), kind
    for side in (0, 1):
        m = MsPpm()
        o.mo_msppm_init(C.byref(m), fs, -6.0)
        seq = []
        for q in blocks:
            bl, br = xl[q:q + B].copy(), xr[q:q + B].copy()
            o.mo_msppm_process(C.byref(m), fp(bl), fp(br), B, side)
            seq.append(o.mo_msppm_read(C.byref(m)))
        assert np.array_equal(bits(seq), bits(G[f"msppm{'MS'[side]}_{tag}"])), side
    c = Stcorr()
    o.mo_stcorr_init(C.byref(c), int(fs), 2e3, 0.3)
    seq = []
    for q in blocks:
        bl, br = xl[q:q + B].copy(), xr[q:q + B].copy()
        o.mo_stcorr_process(C.byref(c), fp(bl), fp(br), B)
        seq.append(o.mo_stcorr_read(C.byref(c)))
    assert np.array_equal(bits(seq), bits(G[f"stcorr_{tag}"]))
    k = Kmeter()
    o.mo_kmeter_init(C.byref(k), fs)
    a, pk, seq = F(), F(), []
    for q in blocks:
        blk = xl[q:q + B].copy()
        o.mo_kmeter_process(C.byref(k), fp(blk), B)
        o.mo_kmeter_read(C.byref(k), C.byref(a), C.byref(pk))
        seq.append((a.value, pk.value))
    assert np.array_equal(bits(seq), bits(G[f"kmeter_{tag}"]))


# ---- the plugins (CPU plumbing, src/meters.cc:298-331 run, :333-412 kmeter_run, :566-588 cor_run,
#      :603-637 bbcm_run): the same vectors through instantiate / connect_port / run ----

def _f(v=0.0):
    return np.array([v], np.float32)


@pytest.fixture(scope="module")
def host():
    from _lv2host import Host
    return Host()


@pytest.mark.parametrize("fs", RATES)
def test_plugins_match_the_reference_vectors(host, fs):
    from _lv2host import Instance
    xl, xr = signal()
    tag = str(int(fs))
    blocks = list(range(0, xl.size - B + 1, B))
    ref = _f(-18.0)                                            # reference level -18 -> rlgain 1 (src/meters.cc:303-306)
    for name, key in (("DINmono", f"iec1_{tag}"), ("NORmono", f"iec1_{tag}"), ("BBCmono", f"iec2_{tag}"), ("EBUmono", f"iec2_{tag}")):
        inst = Instance(host, name, rate=fs)
        assert inst.ok(), name
        lv = _f()
        inst.connect(0, ref); inst.connect(3, lv)
        seq = []
        for q in blocks:
            blk = xl[q:q + B].copy()
            inst.connect(1, blk); inst.connect(2, blk)
            inst.run(B)
            seq.append(lv[0])
        assert np.array_equal(bits(seq), bits(G[key])), name
        inst.cleanup()
    # stereo variant: two independent needles
    inst = Instance(host, "EBUstereo", rate=fs)
    l0, l1 = _f(), _f()
    inst.connect(0, ref); inst.connect(3, l0); inst.connect(6, l1)
    seq = []
    for q in blocks:
        bl, br = xl[q:q + B].copy(), xl[q:q + B].copy()
        for port, arr in ((1, bl), (2, bl), (4, br), (5, br)):
            inst.connect(port, arr)
        inst.run(B)
        seq.append((l0[0], l1[0]))
    seq = np.array(seq, np.float32)
    assert np.array_equal(bits(seq[:, 0]), bits(G[f"iec2_{tag}"])) and np.array_equal(bits(seq[:, 1]), bits(G[f"iec2_{tag}"]))
    inst.cleanup()
    # correlation
    inst = Instance(host, "COR", rate=fs)
    lv = _f()
    inst.connect(0, ref); inst.connect(3, lv)
    seq = []
    for q in blocks:
        bl, br = xl[q:q + B].copy(), xr[q:q + B].copy()
        for port, arr in ((1, bl), (2, bl), (4, br), (5, br)):
            inst.connect(port, arr)
        inst.run(B)
        seq.append(lv[0])
    assert np.array_equal(bits(seq), bits(G[f"stcorr_{tag}"]))
    inst.cleanup()
    # BBC M/S: level0 = M needle, level1 = S needle (port 7 <= 0.5: S at -6 dB)
    inst = Instance(host, "BBCM6", rate=fs)
    l0, l1, s20 = _f(), _f(), _f(0.0)
    inst.connect(0, ref); inst.connect(3, l0); inst.connect(6, l1); inst.connect(7, s20)
    seq = []
    for q in blocks:
        bl, br = xl[q:q + B].copy(), xr[q:q + B].copy()
        for port, arr in ((1, bl), (2, bl), (4, br), (5, br)):
            inst.connect(port, arr)
        inst.run(B)
        seq.append((l0[0], l1[0]))
    seq = np.array(seq, np.float32)
    assert np.array_equal(bits(seq[:, 0]), bits(G[f"msppmM_{tag}"])) and np.array_equal(bits(seq[:, 1]), bits(G[f"msppmS_{tag}"]))
    inst.cleanup()
    # K-meter, mono: level on port 3, peak on port 4 (the unused input pointer), hold on port 5
    inst = Instance(host, "K20mono", rate=fs)
    lv, pk, hold = _f(), _f(), _f()
    kref = _f(-18.0)
    inst.connect(0, kref); inst.connect(3, lv); inst.connect(4, pk); inst.connect(5, hold)
    seq, first = [], True
    want = G[f"kmeter_{tag}"]
    hmax = 0.0
    for i, q in enumerate(blocks):
        blk = xl[q:q + B].copy()
        inst.connect(1, blk); inst.connect(2, blk)
        inst.run(B)
        if first:
            # |ref| >= 3 and != p_refl: p_refl latches, no reset, normal read (src/meters.cc:341-356)
            first = False
        seq.append((lv[0], pk[0]))
        hmax = max(hmax, float(pk[0]))
        assert hold[0] == np.float32(hmax)
    assert np.array_equal(bits(seq), bits(want))
    # the reset handshake: |ref| < 3 resets the DSP and scribbles a negative marker on the hold port
    kref[0] = 1.0
    blk = xl[:B].copy()
    inst.connect(1, blk); inst.connect(2, blk)
    inst.run(B)
    assert hold[0] <= -1.0
    inst.cleanup()


@pytest.mark.parametrize("chn", [3, 5, 8])
def test_surround_meters_match_the_reference_vectors(host, chn):
    """surroundN (src/surmeter.c): a K-meter per channel and correlation meters over selectable channel pairs —
    the same DSP objects as K20mono and COR, so the same golden sequences must come out of the ports."""
    from _lv2host import Instance
    fs, tag = 48000.0, "48000"
    xl, xr = signal()
    blocks = list(range(0, xl.size - B + 1, B))
    inst = Instance(host, "surround%d" % chn, rate=fs)
    assert inst.ok()
    inst.connect(0, _f(-18.0))
    cors = 4 if chn > 3 else 3
    sel_a, sel_b, cor = [_f(0.0) for _ in range(4)], [_f(1.0) for _ in range(4)], [_f() for _ in range(4)]
    sel_a[1][0], sel_b[1][0] = 1.0, 99.0                      # second meter: (R, clamped to the last channel)
    for c in range(4):
        inst.connect(1 + 3 * c, sel_a[c]); inst.connect(2 + 3 * c, sel_b[c]); inst.connect(3 + 3 * c, cor[c])
    lv, pk = [_f() for _ in range(chn)], [_f() for _ in range(chn)]
    for c in range(chn):
        inst.connect(13 + 4 * c + 2, lv[c]); inst.connect(13 + 4 * c + 3, pk[c])
    seq_k, seq_c = [], []
    for q in blocks:
        bufs = [(xl if c % 2 == 0 else xr)[q:q + B].copy() for c in range(chn)]   # even channels = L, odd = R
        for c in range(chn):
            inst.connect(13 + 4 * c, bufs[c]); inst.connect(13 + 4 * c + 1, bufs[c])
        inst.run(B)
        seq_k.append((lv[0][0], pk[0][0]))
        seq_c.append(cor[0][0])
        assert lv[2][0] == lv[0][0] and pk[2][0] == pk[0][0]   # channel 2 carries L as well
    assert np.array_equal(bits(seq_k), bits(G[f"kmeter_{tag}"]))
    assert np.array_equal(bits(seq_c), bits(G[f"stcorr_{tag}"]))   # meter 0 = (channel 0, channel 1) = (L, R)
    assert cors == 3 or np.isfinite(cor[3][0])
    inst.cleanup()
