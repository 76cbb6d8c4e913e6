"""The CPU oracle against the committed golden vectors (tests/golden/golden_v1.npz, generated
from the reference's own objects by tests/golden/make_golden.py).  Runs anywhere (no GPU, no
/root/reference): this is what keeps the oracle pinned on the GPU box."""
import os
import sys

import numpy as np
import pytest

import _signals as sig

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_golden import tri_noise  # noqa: E402  (input recipe shared with the generator)

G = np.load(os.path.join(HERE, "golden", "golden_v1.npz"))


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32 if a.dtype == np.float32 else np.uint64)


def test_coefficients(oracle):
    for j, r in enumerate(G["rates"]):
        assert np.array_equal(_bits(oracle.kw_coef(float(r))), _bits(G["kw_coef"][j]))
        assert np.array_equal(_bits(oracle.tp_consts(float(r))), _bits(G["tp_consts"][j]))
        for i in range(30):
            assert np.array_equal(_bits(oracle.band_coef(float(r), i)), _bits(G["band_coef"][j, i])), (r, i)
    assert np.array_equal(_bits(oracle.tp_table()), _bits(G["tp_table"]))
    # SURVEY.md a1 probe values @48k
    k = oracle.kw_coef(48000.0)
    assert np.allclose(k, [1.5351752, -2.69206738, 1.19870186, -1.69091594, 0.732725799,
                           0.00995242409, 2.47953594e-05], rtol=2e-7)


@pytest.mark.parametrize("i", range(5))
def test_ebu_exact_cases(oracle, i):
    T, seed, gain, fs, block = G["ebu_cases"][i]
    x = tri_noise(int(T), int(seed), float(gain))
    r = oracle.ebu(x, float(fs), int(block), want_frag=True)
    assert np.array_equal(_bits(r["out9"]), _bits(G[f"ebu{i}_out9"]))
    assert np.array_equal(r["hist_M"], G[f"ebu{i}_hist_M"])
    assert np.array_equal(r["hist_S"], G[f"ebu{i}_hist_S"])
    assert np.array_equal(r["counts"], G[f"ebu{i}_counts"])
    assert np.array_equal(_bits(r["frag_power"]), _bits(G[f"ebu{i}_frag_power"]))


def test_ebu_known_answers(oracle):
    tol = 1e-4  # dB; sin() may differ by an ulp between libm builds
    a = oracle.ebu(sig.g0(48000 * 4), 48000.0, 1024)["out9"]
    assert np.allclose(a[:4], G["ebu_g0_out9"][:4], atol=tol) and abs(a[0]) < 1e-3      # 0.0 LUFS
    a = oracle.ebu(sig.g1(48000 * 20), 48000.0, 1024)["out9"]
    assert np.allclose(a, G["ebu_g1_out9"], atol=tol)
    assert abs(a[0] + 23.0070) < 1e-3 and abs(a[4] + 23.0328) < 1e-3
    r = oracle.ebu(sig.g2(48000 * 30, 777), 48000.0, 1024)
    assert np.allclose(r["out9"], G["ebu_g2_out9"], atol=tol)
    assert np.array_equal(r["counts"], G["ebu_g2_counts"])
    r = oracle.ebu(sig.dc_plus_quiet(48000 * 6), 48000.0, 1024, want_frag=True)
    assert np.array_equal(_bits(r["out9"]), _bits(G["ebu_dc_out9"]))
    assert np.array_equal(_bits(r["frag_power"]), _bits(G["ebu_dc_frag_power"]))


def test_truepeak(oracle):
    assert np.array_equal(_bits(oracle.tp(sig.lcg_noise(48000 * 3, 1234), 48000.0, 1024)), _bits(G["tp_lcg_peak"]))
    pk = oracle.tp(sig.g3(48000 * 2), 48000.0, 1024)
    assert np.array_equal(_bits(pk), _bits(G["tp_g3_peak"]))
    assert abs(20 * np.log10(pk[0]) - 3.1056) < 1e-3                                   # +3.1056 dBTP
    y = oracle.tp_resample(sig.lcg_noise(2000, 31)[:, 0].copy())
    assert np.array_equal(_bits(y), _bits(G["tp_resample_lcg"]))
    x = (sig.lcg_noise(48000 * 2, 9)[:, 0] * np.float32(0.5)).copy()
    assert np.array_equal(_bits(oracle.tp_process_seq(x, 48000.0, 1024)), _bits(G["tp_ballistics_lcg"]))
    # impulse: interpolator latency is 24 input frames
    imp = np.zeros(64, np.float32)
    imp[0] = 1.0
    assert int(np.argmax(np.abs(oracle.tp_resample(imp)))) == 96


def test_golden_v2_ballistics_and_nan(oracle):
    """golden_v2.npz (round 5, written by make_golden.py v2 from the reference build): TruePeakdsp::process at 44.1 / 96 kHz and at
    -40 / -60 / -80 dBFS, and the 4x stream around a NaN — the restatement reproduces every array bit for bit."""
    from make_golden import nan_cases, tpb_cases
    G2 = np.load(os.path.join(HERE, "golden", "golden_v2.npz"))
    for name, fs, block, x in tpb_cases():
        assert np.array_equal(_bits(oracle.tp_process_seq(x, fs, block)), _bits(G2[name])), name
    for k, x in nan_cases().items():
        y = oracle.tp_resample(x)
        assert np.array_equal(np.isnan(y), np.isnan(G2["tp_nan_%s_out" % k])), k
        ok = ~np.isnan(y)
        assert np.array_equal(_bits(y[ok]), _bits(G2["tp_nan_%s_out" % k][ok])), k
        assert np.array_equal(_bits(oracle.tp(np.stack([x, x], 1), 48000.0, 1024)), _bits(G2["tp_nan_%s_peak" % k])), k


def test_filter_bank(oracle):
    r = oracle.spectr(sig.lcg_noise(48000, 42, 0.5), 48000.0, 1024)
    for k in ("val", "max", "val_db", "max_db"):
        assert np.array_equal(_bits(r[k]), _bits(G[f"spectr_lcg_{k}"])), k
    r = oracle.spectr(sig.g4(48000 * 2, 16), 48000.0, 1024)
    assert np.allclose(r["val_db"], G["spectr_g4_16_val_db"], atol=1e-3)
    assert abs(r["val_db"][16]) < 0.01                                                 # 0 dB in-band
    r = oracle.spectr(sig.lcg_noise(44100, 43, 0.5), 44100.0, 1024)
    assert np.array_equal(_bits(r["val"]), _bits(G["spectr_lcg441_val"]))


def test_vu(oracle):
    x = sig.lcg_noise(48000, 77)[:, 0].copy()
    assert np.array_equal(_bits(oracle.vu(x, 48000.0, 1024)), _bits(G["vu_lcg_block1024"]))
    v = oracle.vu(sig.g0(48000)[:, 0].copy(), 48000.0, None)
    assert np.allclose(v, G["vu_g0_onecall"], rtol=1e-5) and abs(float(v[0]) - 1.010453) < 5e-6
    # n mod 4 trailing samples are dropped (vumeterdsp.cc:54)
    a = oracle.vu(x[:1003].copy(), 48000.0, None)
    b = oracle.vu(x[:1000].copy(), 48000.0, None)
    assert np.array_equal(_bits(a), _bits(b))


def test_lcg_generators_agree(oracle):
    assert np.array_equal(_bits(oracle.fill_lcg(5000, 777, 0.25)), _bits(sig.lcg_noise(5000, 777, 0.25)))


def test_bitstats_known_answers(oracle):
    """float_stats (bitmeter.c:63-105) is pure integer logic; src/bitmeter.c needs LV2 headers and
    cannot be compiled here, so this path is pinned by hand-derived known answers."""
    x = np.array([1.0, -1.0, 0.0, -0.0, np.inf, -np.inf, np.nan, 1.5, 2.0 ** -149, 0.75], np.float32)
    r = oracle.bitstats(x)
    assert list(r["counters"]) == [2, 4, 1, 2, 1]        # zero, pos, nan, inf, denormal
    assert r["vmax"] == np.float32(1.5) and r["vmin"] == np.float32(0.75)
    h = r["hist"]
    # bit position p = exp + k, k = 0..22 mantissa, k = 23 the implicit one (BIM_NHIT = BIM_DHIT + 23):
    # exp 127 (1.0, -1.0, 1.5) covers p = 127..150, exp 126 (0.75) p = 126..149, the denormal p = 1..23
    assert h[150] == 3 and h[149] == 4 and h[127] == 4 and h[126] == 1 and h[1] == 1 and h[23] == 1
    # ones: region offset 280, same positions; 1.5 sets bit 22 (p 149), 0.75 sets bit 22 (p 148)
    assert h[280 + 150] == 3 and h[280 + 149] == 2 and h[280 + 148] == 1 and h[280 + 1] == 1
    assert h[560 + 22] == 2 and h[560 + 0] == 1
    assert h.sum() == 23 * 5 + 2 * 4 + (2 + 2 + 2)       # 5 values x 23 hits + (NHIT+NONE) x 4 + ones
    soup = sig.g5(20000)
    r = oracle.bitstats(soup)
    bits = soup.view(np.uint32)
    ex = (bits >> 23) & 0xFF
    man = bits & 0x7FFFFF
    assert r["counters"][2] == int(((ex == 255) & (man != 0)).sum())
    assert r["counters"][3] == int(((ex == 255) & (man == 0)).sum())
    assert r["counters"][0] == int(((ex == 0) & (man == 0)).sum())
    assert r["counters"][4] == int(((ex == 0) & (man != 0)).sum())
    live = (ex != 255) & ~((ex == 0) & (man == 0))
    assert h.sum() > 0 and r["hist"][0:280].sum() == int(live.sum()) * 23 + int((live & (ex > 0)).sum())
    for k in (0, 7, 22):
        assert r["hist"][560 + k] == int((live & ((man >> k) & 1 == 1)).sum())


def test_sigdist_known_answers(oracle):
    """sdh_run's loop (sigdistlv2.c:303-318): bin = rintf(180 + 150 x); pinned by known answers."""
    x = np.array([0.0, 1.0, -1.0, 1.2, -1.2, 1.21, 0.5 / 150, 1.5 / 150, np.float32(-0.0033333334)], np.float32)
    r = oracle.sigdist(x)
    b = r["bins"]
    assert b[180] == 3 and b[330] == 1 and b[30] == 1 and b[360] == 1 and b[0] == 1   # 0.5/150 -> 180.5 -> 180 (even)
    assert b[182] == 1                                                              # 1.5/150 -> 181.5 -> 182
    assert b.sum() == 8 and r["count"] == 9                                         # 1.21 -> 361.5 dropped
    y = sig.lcg_noise(30000, 3)[:, 0].copy()
    r = oracle.sigdist(y)
    ref_bins = np.bincount(np.rint(np.float32(180.0) + y * np.float32(150.0)).astype(np.int64), minlength=361)
    assert np.array_equal(r["bins"], ref_bins[:361])
    assert abs(r["var_m"] - y.astype(np.float64).mean()) < 1e-12
    assert abs(r["var_s"] / y.size - y.astype(np.float64).var()) < 1e-9
