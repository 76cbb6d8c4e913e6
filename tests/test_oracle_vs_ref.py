"""Pin the CPU restatement (oracle/mtr_oracle.c) bit-for-bit against the reference's own
objects (oracle/_ref, compiled from /root/reference).  Runs only where that tree exists;
on the GPU box these are skipped and tests/test_oracle_golden.py carries the pin."""
import numpy as np
import pytest

import _signals as sig

pytestmark = pytest.mark.ref


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32 if a.dtype == np.float32 else np.uint64)


@pytest.mark.parametrize("fs", [44100.0, 48000.0, 96000.0])
def test_kweight_coefficients_bit_exact(oracle, reference, fs):
    assert np.array_equal(_bits(oracle.kw_coef(fs)), _bits(reference.kw_coef(fs)))


def test_fir_table_bit_exact(oracle, reference):
    a, b = oracle.tp_table(), reference.tp_table()
    assert np.array_equal(_bits(a), _bits(b))
    # row 0 is the identity phase, row 4 ~ 0, rows 1/3 mirror, row 2 sums to 0.5
    assert a[23] == 1.0 and np.all(np.abs(a[:23]) < 1e-15)
    assert np.all(np.abs(a[96:120]) < 1e-15)
    assert abs(a[48:72].sum() - 0.5) < 1e-6


def test_resampler_state_after_init(reference):
    nread, phase, _ = reference.tp_resampler_state()
    assert (nread, phase) == (1, 0)


@pytest.mark.parametrize("block", [64, 1000, 1024, 8192])
def test_ebu_bit_exact(oracle, reference, block):
    x = sig.lcg_noise(48000 * 12, 777, 0.25)
    # triangle envelope from integer counters so gating and LRA are exercised reproducibly
    T = x.shape[0]
    n = np.arange(T)
    env = (np.abs((n % 240000) - 120000).astype(np.float32) / np.float32(120000.0))
    x = (x * env[:, None]).astype(np.float32)
    a = oracle.ebu(x, 48000.0, block, want_frag=True)
    b = reference.ebu(x, 48000.0, block, want_frag=True)
    assert np.array_equal(_bits(a["out9"]), _bits(b["out9"]))
    assert np.array_equal(a["hist_M"], b["hist_M"]) and np.array_equal(a["hist_S"], b["hist_S"])
    assert np.array_equal(a["counts"], b["counts"])
    assert np.array_equal(_bits(a["frag_power"]), _bits(b["frag_power"]))


@pytest.mark.parametrize("fs", [44100.0, 96000.0])
def test_ebu_other_rates_bit_exact(oracle, reference, fs):
    x = sig.lcg_noise(int(fs) * 11, 5, 0.5)
    a = oracle.ebu(x, fs, 1024)
    b = reference.ebu(x, fs, 1024)
    assert np.array_equal(_bits(a["out9"]), _bits(b["out9"]))
    assert np.array_equal(a["hist_M"], b["hist_M"]) and np.array_equal(a["hist_S"], b["hist_S"])


def test_resampler_outputs_bit_exact(oracle, reference):
    x = sig.lcg_noise(20000, 31)[:, 0].copy()
    assert np.array_equal(_bits(oracle.tp_resample(x)), _bits(reference.tp_resample(x)))


@pytest.mark.parametrize("block", [64, 1024, 8192])
def test_truepeak_max_bit_exact(oracle, reference, block):
    x = sig.lcg_noise(48000 * 3, 1234)
    assert np.array_equal(_bits(oracle.tp(x, 48000.0, block)), _bits(reference.tp(x, 48000.0, block)))
    y = sig.g3(48000 * 2)
    pa, pb = oracle.tp(y, 48000.0, block), reference.tp(y, 48000.0, block)
    assert np.array_equal(_bits(pa), _bits(pb))
    assert abs(pa[0] - 1.429816) < 2e-6          # SURVEY.md §4 known answer (+3.1056 dBTP)


@pytest.mark.parametrize("fs", [44100.0, 48000.0, 96000.0])
def test_truepeak_ballistics_bit_exact(oracle, reference, fs):
    assert np.array_equal(_bits(oracle.tp_consts(fs)), _bits(reference.tp_consts(fs)))
    x = (sig.lcg_noise(48000 * 2, 9)[:, 0] * np.float32(0.5)).copy()
    a = oracle.tp_process_seq(x, fs, 1024)
    b = reference.tp_process_seq(x, fs, 1024)
    assert np.array_equal(_bits(a), _bits(b))


@pytest.mark.parametrize("rate", [44100.0, 48000.0, 96000.0])
def test_band_coefficients_bit_exact(oracle, reference, rate):
    for i in range(30):
        assert np.array_equal(_bits(oracle.band_coef(rate, i)), _bits(reference.band_coef(rate, i))), i


@pytest.mark.parametrize("block", [64, 1024, 4800])
def test_spectr_bit_exact(oracle, reference, block):
    x = sig.lcg_noise(48000, 42, 0.5)
    a = oracle.spectr(x, 48000.0, block)
    b = reference.spectr(x, 48000.0, block)
    for k in ("val", "max", "val_db", "max_db"):
        assert np.array_equal(_bits(a[k]), _bits(b[k])), k


def test_vu_bit_exact(oracle, reference):
    x = sig.g0(48000)[:, 0].copy()
    for block in (None, 1024, 1001):
        assert np.array_equal(_bits(oracle.vu(x, 48000.0, block)), _bits(reference.vu(x, 48000.0, block)))
    # SURVEY.md §8c probe: read() = 1.010453 for 1 s of 0 dBFS 1 kHz in one call
    assert abs(float(oracle.vu(x)[0]) - 1.010453) < 2e-6
