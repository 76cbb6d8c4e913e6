"""GPU parity of the integer paths (bit meter table, signal-distribution histogram): BIT-EXACT against
the oracle on the same buffers — every one of the 584 / 361 int32 counters, the special-value
counters, min / max and the peak bin; only the double-precision moments of the SDH carry a
tolerance (1e-12 relative: pairwise instead of sequential Welford)."""
import numpy as np
import pytest

import _signals as sig

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def M():
    import meters.lv2_amd as m
    return m


def _check_bim(got, s, want):
    assert np.array_equal(got["hist"][s], want["hist"]), np.flatnonzero(got["hist"][s] != want["hist"])[:10]
    assert np.array_equal(got["counters"][s], want["counters"])
    assert got["vmin"][s] == want["vmin"] and got["vmax"][s] == want["vmax"]


def test_bitstats_bit_exact(M, oracle):
    S, T = 7, 30011
    x = np.stack([sig.lcg_noise(T, 600 + s, 2.0 ** -(3 * s))[:, 0] for s in range(S)])
    x[1] = sig.g5(T, 4242)                       # bit-pattern soup: NaN / Inf / denormals / +-0, every exponent
    x[2, ::3] = 0.0
    x[3] = np.float32(1e-40) * np.arange(T, dtype=np.float32)   # denormals only
    with M.Engine(S, 48000.0, M.METER_BITSTATS, n_channels=1) as e:
        e.process(x)
        got = e.bitstats()
        for s in range(S):
            _check_bim(got, s, oracle.bitstats(x[s]))
        # accumulates over calls like the reference's averaging mode; reset clears
        e.process(x[:, :1000])
        got = e.bitstats()
        for s in (0, 1):
            _check_bim(got, s, oracle.bitstats(np.concatenate([x[s], x[s, :1000]])))
        e.intstat_reset()
        r = e.bitstats()
        assert r["hist"].sum() == 0 and np.all(np.isinf(r["vmin"])) and np.all(r["vmax"] == 0)


def test_bitstats_class_switches_and_flushes(M, oracle):
    """The positional counters of k_bitstats work inside one class of 32 exponents at a time and are
    unloaded every 255 blocks; everything else goes through the per-exponent path.  Streams that switch
    class, sit on class borders, are integer scaled, hold -0 / inf / nan / denormals in bulk, are not a
    multiple of the block size and are longer than one unload interval must all stay bit-exact."""
    T = 4 * 255 * 1024 + 4 * 1024 + 777                  # > one unload per wave, ragged tail
    rng = np.random.default_rng(77)
    base = sig.lcg_noise(T, 31, 1.0)[:, 0]
    x = np.zeros((8, T), np.float32)
    x[0] = base                                           # class 3 throughout
    x[1] = base * np.float32(2.0 ** 15)                   # "int16 scaled": class 4 with excursions into 3
    x[2] = base * np.float32(2.0 ** -40)                  # tiny: class 2 / 1
    x[2, T // 2:] = base[T // 2:]                         # ... then ordinary audio: the wave must re-pick its class
    x[3] = base
    x[3, ::5] = -0.0
    x[3, 7::1001] = np.inf
    x[3, 11::1003] = -np.inf
    x[3, 13::997] = np.nan
    x[3, 3::17] = np.float32(1e-41)                       # denormals sprinkled over audio
    x[4] = (rng.integers(-2 ** 23, 2 ** 23, T) * 2.0 ** -23).astype(np.float32) * np.float32(2.0 ** -28)   # straddles 2^-31
    x[5] = np.where(np.arange(T) % 2 == 0, base, base * np.float32(2.0 ** 40))    # alternating classes 3 / 4-5
    x[6] = sig.g5(T, 99)                                  # bit soup at length
    x[7, 1000:] = base[1000:] * np.float32(1.9999999)     # starts with digital silence; values up to just under 2.0
    with M.Engine(8, 48000.0, M.METER_BITSTATS, n_channels=1) as e:
        e.process(x)
        got = e.bitstats()
        for s in range(8):
            _check_bim(got, s, oracle.bitstats(x[s]))
    # an odd stride makes every second stream unaligned for 16-byte loads
    y = np.ascontiguousarray(x[:4, :30001])
    with M.Engine(4, 48000.0, M.METER_BITSTATS, n_channels=1) as e:
        e.process(y)
        got = e.bitstats()
        for s in range(4):
            _check_bim(got, s, oracle.bitstats(y[s]))


def test_sigdist_bit_exact_bins(M, oracle):
    S, T = 6, 48000
    x = np.stack([sig.lcg_noise(T, 900 + s, 2.0 ** -s)[:, 0] for s in range(S)])
    x[1] *= np.float32(3.0)                                  # peaks at 1.5: samples beyond +-1.2 are dropped
    x[2, :] = np.float32(0.5 / 150)                          # 180.5 -> rint ties to even (bin 180)
    x[3, 100] = np.nan
    x[3, 200] = np.inf
    x[4] = sig.sine(T, 997.0, 0.9)[:, 0]
    with M.Engine(S, 48000.0, M.METER_SIGDIST, n_channels=1) as e:
        for a, b in ((0, 20000), (20000, 48000)):            # two calls: state carries
            e.process(np.ascontiguousarray(x[:, a:b]))
        got = e.sigdist()
    for s in range(S):
        want = oracle.sigdist(x[s])
        assert np.array_equal(got["bins"][s], want["bins"]), s
        assert got["peak_cnt"][s] == want["peak_cnt"] and got["peak_bin"][s] == want["peak_bin"], s
        assert got["count"][s] == want["count"]
        assert abs(got["avg"][s] - want["avg"]) <= 1e-12 * max(1.0, abs(want["avg"])) * T     # the plain sum: always the reference's
        # var_m / var_s are the reference's accumulators in BOTH regimes: Welford's mean / M2 while every sample is binned,
        # and — once a sample was skipped (|x| > 1.2, NaN / Inf: streams 1 and 3) — the recurrence that keeps dividing by
        # the index among ALL samples (sigdistlv2.c:312-315), which the kernel's second pass reproduces
        assert abs(got["var_m"][s] - want["var_m"]) <= 1e-12 * max(1.0, abs(want["var_m"])), (s, got["var_m"][s], want["var_m"])
        assert abs(got["var_s"][s] - want["var_s"]) <= 1e-12 * max(1.0, abs(want["var_s"])) * 10, (s, got["var_s"][s], want["var_s"])
        if s in (1, 3):
            with np.errstate(invalid="ignore"):
                fb = np.rint(np.float32(180.0) + x[s] * np.float32(150.0))
            kept = x[s][np.isfinite(fb) & (fb >= 0) & (fb < 361)].astype(np.float64)
            assert kept.size == want["bins"].sum() and kept.size < T
            assert abs(want["var_m"] - kept.mean()) > 1e-9          # (the reference's value is indeed no longer the mean)


def test_sigdist_moments_after_a_skipped_sample_across_calls(M, oracle):
    """One out-of-range sample in the FIRST of five uneven calls: every later call of that stream must keep the reference's
    divisor (the raw sample index), also calls that skip nothing themselves; the clean neighbour stays on the fast path."""
    T = 30011
    x = np.stack([sig.lcg_noise(T, 70 + s, 0.4)[:, 0] + np.float32(0.25) for s in range(3)])   # a DC offset: m is not small
    x[1, 17] = np.float32(7.0)
    x[2, 29000] = np.float32(-np.inf)
    cuts = [0, 100, 4099, 4100, 20000, T]
    with M.Engine(3, 48000.0, M.METER_SIGDIST, n_channels=1) as e:
        for a, b in zip(cuts[:-1], cuts[1:]):
            e.process(np.ascontiguousarray(x[:, a:b]))
        got = e.sigdist()
    for s in range(3):
        want = oracle.sigdist(x[s])
        assert np.array_equal(got["bins"][s], want["bins"]) and got["count"][s] == want["count"]
        assert abs(got["var_m"][s] - want["var_m"]) <= 1e-12 * max(1.0, abs(want["var_m"])), (s, got["var_m"][s], want["var_m"])
        assert abs(got["var_s"][s] - want["var_s"]) <= 1e-11 * max(1.0, abs(want["var_s"])), (s, got["var_s"][s], want["var_s"])


def test_peak_bin_tie_break(M, oracle):
    """Two bins end with the same maximal count: the reference's strict `>` keeps the bin that got
    there first."""
    T = 4000
    x = np.zeros((2, T), np.float32)
    x[0, :2000] = 0.5          # bin 255 reaches 2000 first
    x[0, 2000:] = -0.5         # bin 105 reaches 2000 later
    x[1, ::2] = -0.5           # interleaved: bin 105 gets its last hit one sample before bin 255
    x[1, 1::2] = 0.5
    with M.Engine(2, 48000.0, M.METER_SIGDIST, n_channels=1) as e:
        e.process(x)
        got = e.sigdist()
    for s in range(2):
        want = oracle.sigdist(x[s])
        assert (got["peak_cnt"][s], got["peak_bin"][s]) == (want["peak_cnt"], want["peak_bin"]), s
    assert got["peak_bin"][0] == 255 and got["peak_bin"][1] == 105


@pytest.mark.timeout(600)
def test_integer_paths_full_size(M, oracle):
    """4096 mono streams x 10 s (7.9 GB): the sum of all per-stream tables equals the table of the
    concatenated sample (integer counting is order-free), and sampled streams match the oracle."""
    import torch
    S, T = 4096, 480000
    buf = torch.empty((S // 2, T, 2), dtype=torch.float32, device="cuda")     # reuse the stereo synth: [S/2][T][2]
    M.synth_fill_device(buf.data_ptr(), S // 2, T, T, 31, 48000.0, 0)
    mono = buf.view(S // 2, 2 * T)          # each stereo stream read as one mono stream of 2T samples
    with M.Engine(S // 2, 48000.0, M.METER_BITSTATS | M.METER_SIGDIST, n_channels=1) as e:
        e.process_device(mono.data_ptr(), 2 * T)
        b, d = e.bitstats(), e.sigdist()
    for s in (0, S // 4, S // 2 - 1):
        h = mono[s].cpu().numpy()
        wb, wd = oracle.bitstats(h), oracle.sigdist(h)
        assert np.array_equal(b["hist"][s], wb["hist"]) and np.array_equal(b["counters"][s], wb["counters"])
        assert np.array_equal(d["bins"][s], wd["bins"]) and d["peak_bin"][s] == wd["peak_bin"]
    assert d["bins"].sum() == (S // 2) * 2 * T               # uniform [-1, 1): every sample lands in a bin
    # LCG floats are 24-bit integers / 2^23: the lowest mantissa bit of a normalised value is never set
    assert b["hist"][:, 560].sum() == 0 and b["hist"][:, 561:583].sum(0).min() > 0
