"""The LV2 plugin surface lib/meters_amd.so driven by a ctypes mini host: the reference's
descriptor / port / run() contract.  src/meters.cc itself cannot be compiled here
(no LV2 SDK headers), so run() semantics are checked against the oracle's DSP plus the
known-answer values the survey probed through the reference's real run()."""
import numpy as np
import pytest

import _signals as sig
from _lv2host import MTR_URI, Host, Instance, arm_notify, forge_object, forge_sequence, notify_buffer, parse_sequence

IN_SCOPE = ["VUmono", "VUstereo", "EBUr128", "spectr30mono", "dBTPmono", "dBTPstereo", "spectr30stereo",
            "SigDistHist", "bitmeter",
            "BBCmono", "BBCstereo", "EBUmono", "EBUstereo", "DINmono", "DINstereo", "NORmono", "NORstereo", "COR", "BBCM6",
            "K12mono", "K14mono", "K20mono", "K12stereo", "K14stereo", "K20stereo",
            "dr14mono", "dr14stereo", "TPnRMSmono", "TPnRMSstereo",
            "surround8", "surround7", "surround6", "surround5", "surround4", "surround3"]


@pytest.fixture(scope="module")
def host():
    return Host()


def _f(v=0.0):
    return np.array([v], np.float32)


def test_descriptor_enumeration(host):
    ds = host.descriptors()
    assert [d.URI.decode() for d in ds] == [MTR_URI + n for n in IN_SCOPE]
    for d in ds:   # 8 positional members; activate / deactivate are NULL as in the reference
        assert d.instantiate and d.connect_port and d.run and d.cleanup and d.extension_data
        assert not d.activate and not d.deactivate
        assert not d.extension_data(b"http://example.org/none")
    assert not host.lib.lv2_descriptor(len(ds))


def test_vu_mono_config0(host, oracle):
    """BASELINE configs[0]: mono VU on 1 s of 48 kHz sine through instantiate/connect_port/run/cleanup (CPU)."""
    x = sig.g0(48000)[:, 0].copy()
    inst = Instance(host, "VUmono")
    assert inst.ok()
    ref, level, out = _f(-22.0), _f(), np.zeros(48000, np.float32)
    inst.connect(0, ref); inst.connect(1, x); inst.connect(2, out); inst.connect(3, level)
    inst.run(48000)
    assert abs(float(level[0]) - 0.637553) < 2e-6            # SURVEY.md §8c probe of the reference's run()
    assert np.array_equal(out, x)                            # pass-through copy
    rlgain = np.float32(10.0) ** np.float32(0.05 * (-22.0 + 18.0))
    assert np.float32(level[0]) == np.float32(rlgain * oracle.vu(x)[0])
    inst.cleanup()
    # block-wise: max over 1024-frame blocks and the last block (same probe)
    inst = Instance(host, "VUmono")
    inst.connect(0, ref); inst.connect(3, level)
    seen = []
    for p in range(0, 48000 - 1023, 1024):
        blk = x[p:p + 1024].copy()
        inst.connect(1, blk); inst.connect(2, blk)           # in-place is allowed
        inst.run(1024)
        seen.append(float(level[0]))
    want = rlgain * oracle.vu(x[:len(seen) * 1024].copy(), 48000.0, 1024)
    assert np.array_equal(np.array(seen, np.float32), want.astype(np.float32))
    assert abs(max(seen) - 0.637553) < 2e-6
    inst.cleanup()


def test_vu_stereo_and_reference_level(host, oracle):
    x = sig.lcg_noise(4800, 11)
    L, R = x[:, 0].copy(), x[:, 1].copy()
    inst = Instance(host, "VUstereo")
    ref, l0, l1 = _f(-18.0), _f(), _f()
    oL, oR = np.zeros_like(L), np.zeros_like(R)
    for port, arr in ((0, ref), (1, L), (2, oL), (3, l0), (4, R), (5, oR), (6, l1)):
        inst.connect(port, arr)
    inst.run(4800)
    assert np.float32(l0[0]) == oracle.vu(L)[0] and np.float32(l1[0]) == oracle.vu(R)[0]   # ref -18 -> gain 1
    assert np.array_equal(oR, R)
    inst.cleanup()


def test_gpu_plugins_refuse_to_start_without_a_gpu(host):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert not Instance(host, "EBUr128", with_urid_map=False).ok()      # missing urid:map -> NULL (ebulv2.cc:140-144)
    for name in ("EBUr128", "dBTPstereo", "spectr30stereo"):
        assert not Instance(host, name).ok()                            # no GPU -> NULL, never a CPU fallback


@pytest.mark.gpu
def test_dbtp_stereo(host, oracle):
    x = sig.sine(48000, 1000.0, 1.0, amp_r=0.5)
    L, R = x[:, 0].copy(), x[:, 1].copy()
    inst = Instance(host, "dBTPstereo")
    assert inst.ok()
    ctl, lv0, lv1, pk0, pk1 = _f(0.0), _f(), _f(), _f(), _f()
    for port, arr in ((0, ctl), (3, lv0), (6, lv1), (7, pk0), (8, pk1)):
        inst.connect(port, arr)
    want_l = _dbtp_expected(oracle, L, 1024)
    want_r = _dbtp_expected(oracle, R, 1024)
    ctl[0] = 1.0        # first run: |v| < 3 and != p_refl -> reset handshake: ports get the -500.. marker
    for i, p in enumerate(range(0, 48000 - 1023, 1024)):
        bl, br = L[p:p + 1024].copy(), R[p:p + 1024].copy()
        for port, arr in ((1, bl), (2, bl), (4, br), (5, br)):
            inst.connect(port, arr)
        inst.run(1024)
        if i == 0:
            assert lv0[0] <= -500 and pk1[0] <= -500
            continue
        assert abs(lv0[0] - want_l[i][0]) <= 2e-6 and abs(lv1[0] - want_r[i][0]) <= 2e-6
        assert abs(pk0[0] - want_l[i][1]) <= 2e-6 and abs(pk1[0] - want_r[i][1]) <= 2e-6
    assert abs(lv0[0] - 0.998895) < 2e-5 and abs(lv1[0] - 0.499448) < 2e-5      # SURVEY.md §8c probe
    assert abs(pk0[0] - 1.0) < 2e-6 and abs(pk1[0] - 0.5) < 2e-6
    inst.cleanup()


def _dbtp_expected(oracle, x, block):
    """What dbtp_run leaves on (level, peak) after each block, emulated with the oracle's TruePeakdsp:
    the first run is a reset run (port 0 differs from p_refl = -9999): reset(), process(), early return
    without read(); later runs process() then read(m, p) and max-hold p (src/meters.cc:446-507)."""
    import ctypes as C
    from _oracle import MoTp
    t = MoTp()
    oracle.lib.mo_tp_init(C.byref(t), 48000.0)
    out, pm = [], 0.0
    m, p = C.c_float(), C.c_float()
    for i, q in enumerate(range(0, x.size, block)):
        seg = np.ascontiguousarray(x[q:q + block])
        if i == 0:
            oracle.lib.mo_tp_reset(C.byref(t))
        oracle.lib.mo_tp_process(C.byref(t), seg, seg.size)
        if i == 0:
            out.append(None)
            continue
        oracle.lib.mo_tp_read2(C.byref(t), C.byref(m), C.byref(p))
        pm = max(pm, p.value)
        out.append((m.value, pm))
    return out


@pytest.mark.gpu
def test_dbtp_mono(host, oracle):
    x = (sig.lcg_noise(8192, 5)[:, 0] * np.float32(0.5)).copy()
    x[100] = 0.9                                               # a peak inside the unread reset block
    inst = Instance(host, "dBTPmono")
    ctl, lv, pk = _f(0.0), _f(), _f()
    inst.connect(0, ctl); inst.connect(3, lv); inst.connect(4, pk)     # port 4 doubles as the peak output
    want = _dbtp_expected(oracle, x, 2048)
    for i, p in enumerate(range(0, 8192, 2048)):
        blk = x[p:p + 2048].copy()
        inst.connect(1, blk); inst.connect(2, blk)
        inst.run(2048)
        if want[i] is None:
            assert lv[0] <= -500 and pk[0] <= -500               # reset run: marker values force a port change
        else:
            assert abs(lv[0] - want[i][0]) < 2e-6 and abs(pk[0] - want[i][1]) < 2e-6, i
    assert pk[0] >= 0.9
    inst.cleanup()


@pytest.mark.gpu
def test_spectr30_stereo(host, oracle):
    x = sig.sine(48000, 1000.0, 1.0, amp_r=0.5)
    inst = Instance(host, "spectr30stereo")
    assert inst.ok()
    spec = [_f() for _ in range(30)]
    mx = [_f() for _ in range(30)]
    spd, rst, amp, st = _f(1.0), _f(-4.0), _f(0.0), _f(0.0)
    for i in range(30):
        inst.connect(i, spec[i]); inst.connect(30 + i, mx[i])
    for port, arr in ((60, spd), (61, rst), (62, amp), (63, st)):
        inst.connect(port, arr)
    for p in range(0, 48000 - 1023, 1024):
        bl, br = x[p:p + 1024, 0].copy(), x[p:p + 1024, 1].copy()
        for port, arr in ((64, bl), (65, bl), (66, br), (67, br)):
            inst.connect(port, arr)
        inst.run(1024)
    n = (48000 // 1024) * 1024
    want = oracle.spectr(x[:n], 48000.0, 1024)
    got = np.array([s[0] for s in spec]), np.array([m[0] for m in mx])
    assert np.allclose(got[0], want["val_db"], atol=1e-3) and np.allclose(got[1], want["max_db"], atol=1e-3)
    assert abs(got[0][16] + 2.506) < 0.02                     # SURVEY.md §8c probe (after a full second)
    rst[0] = 1.0                                              # peak reset: marker values on the peak ports
    inst.run(0)
    assert all(m[0] <= -500 for m in mx)
    inst.cleanup()


# The EBUr128 plugin has its own file: tests/test_lv2_ebur128.py (the whole UI protocol, message by message).


@pytest.mark.gpu
def test_engine_failure_is_signalled_not_hidden(host, capfd):
    """an engine call that fails inside run() (here: an audio input port left unconnected, which
    the engine refuses) must not leave the last good values on the meter ports: they carry NaN until the engine
    answers again, and the error is reported once per failure streak."""
    inst = Instance(host, "spectr30stereo")
    assert inst.ok()
    spec = [_f() for _ in range(30)]
    mx = [_f() for _ in range(30)]
    spd, rst, amp, st = _f(1.0), _f(-4.0), _f(0.0), _f(0.0)
    for i in range(30):
        inst.connect(i, spec[i]); inst.connect(30 + i, mx[i])
    for port, arr in ((60, spd), (61, rst), (62, amp), (63, st)):
        inst.connect(port, arr)
    x = sig.sine(4096, 1000.0, 0.5)
    bl, br = x[:1024, 0].copy(), x[:1024, 1].copy()
    for port, arr in ((64, bl), (65, bl), (66, br), (67, br)):
        inst.connect(port, arr)
    inst.run(1024)
    assert all(np.isfinite(s[0]) for s in spec)
    inst.connect(64, None); inst.connect(66, None)            # the inputs go away
    inst.run(1024); inst.run(1024)
    assert all(np.isnan(s[0]) for s in spec) and all(np.isnan(m[0]) for m in mx)
    err = capfd.readouterr().err
    assert err.count("meters_amd: spectr30") == 1              # once per streak, not once per block
    inst.connect(64, bl); inst.connect(66, br)
    inst.run(1024)
    assert all(np.isfinite(s[0]) for s in spec)
    inst.cleanup()
