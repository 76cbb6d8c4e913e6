"""mtr_engine_process_host: the batch crosses the host link in chunks of streams, chunk k + 1 on a copy stream under the
kernels of chunk k (include/mtr_engine.h).  Chunking must be EXACT: every result of every meter, for every stream, bit for
bit what mtr_engine_process_device gives on the same audio resident in HBM — across several calls of uneven length (the
fragment phase, the interpolator's history, the bank's dither parity, the open DR-14 window move once per call, not once
per chunk), with chunk sizes that do not divide the batch, and whichever kernel the whole batch is routed to."""
import numpy as np
import pytest

import _signals as sig

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def M():
    import meters.lv2_amd as m
    return m


def _records(M, e, meters):
    out = {}
    if meters & (M.METER_EBU | M.METER_TRUEPEAK | M.METER_TPBALLIST):
        r = e.results()
        out["o9"] = e.out9()
        out["counts"] = np.array([[x.hist_M_count, x.hist_S_count] for x in r])
        out["tp"] = np.array([[x.truepeak[0], x.truepeak[1], x.truepeak_call[0], x.truepeak_call[1]] for x in r], np.float32)
        out["tpb"] = np.array([[x.tpb_level[0], x.tpb_level[1], x.tpb_peak[0], x.tpb_peak[1]] for x in r], np.float32)
    if meters & M.METER_EBU:
        out["hm"], out["hs"] = e.histograms()
        out["frag"] = e.fragment_powers()
    if meters & M.METER_SPECTR30:
        sp = e.spectrum()
        out["val"], out["max"] = sp["val"], sp["max"]
    if meters & M.METER_BITSTATS:
        b = e.bitstats()
        out.update({"b_" + k: v for k, v in b.items()})
    if meters & M.METER_SIGDIST:
        d = e.sigdist()
        out.update({"d_" + k: v for k, v in d.items()})
    if meters & M.METER_DR14:
        out["dr"] = np.array([[x.m_rms[0], x.m_rms[1], x.m_peak[0], x.m_peak[1], x.dr[0], x.dr[1], x.dr_total, x.block_count] for x in e.dr14()])
    if meters & M.METER_KMETER:
        out["km_rms"], out["km_peak"] = e.kmeter_read()
    return out


def _both_ways(M, x, calls, meters, fs=48000.0, chunk_streams=5, **kw):
    import torch
    S = x.shape[0]
    C = 1 if x.ndim == 2 else 2
    dev = torch.from_numpy(x).cuda()
    st = torch.cuda.current_stream().cuda_stream
    got = []
    for host in (False, True):
        recs = []
        with M.Engine(S, fs, meters, n_channels=C, **kw) as e:
            if meters & M.METER_EBU:
                e.integr_start()
            pos = 0
            for n in calls:
                if host:
                    e.set_host_chunk_bytes(chunk_streams * (((n + 3) & ~3) if C == 1 else ((n + 1) & ~1)) * C * 4)
                    e.process(np.ascontiguousarray(x[:, pos:pos + n]))
                else:
                    e.process_device(dev.data_ptr() + pos * C * 4, n, x.shape[1], st)
                recs.append(_records(M, e, meters))
                pos += n
            seg = e.seg_stats()
        got.append((recs, seg))
    (res, seg_r), (hst, seg_h) = got
    assert seg_r == seg_h                                              # the same kernels served both
    for a, b in zip(res, hst):
        assert a.keys() == b.keys()
        for k in a:
            if k in ("d_avg", "d_var_m", "d_var_s"):
                # the SDH's double sums are grouped by the call's ALIGNMENT (16-byte loads with a scalar head): the staging buffer of
                # the host path and a view into the resident batch start on different bytes, and the groups differ in the last bits
                assert np.allclose(a[k], b[k], rtol=1e-11, atol=0, equal_nan=True), k
            else:
                assert np.array_equal(a[k], b[k], equal_nan=True), k
    return seg_r


def test_chunked_host_path_is_the_resident_path_bit_for_bit(M):
    T = 2400 * 30 + 1234
    S = 37                                                             # 8 chunks of 5, 5, 5, 5, 5, 5, 5, 2 streams
    x = np.stack([sig.g2(T, 300 + s) * np.float32(2.0 ** -(s % 4)) for s in range(S)])
    calls = [2400 * 11, 2400 * 7 + 777, 1000, T - (2400 * 18 + 1777)]
    meters = M.METER_EBU | M.METER_TRUEPEAK | M.METER_SPECTR30
    assert _both_ways(M, x, calls, meters)[0] == 0                     # routed to the wave-per-segment kernel
    seg = _both_ways(M, x, calls, meters, tune_segments=3)              # ... and through the lane = segment kernel:
    assert seg[0] == 3, seg                                             # every call but the 1000-frame one
    _both_ways(M, x, calls, M.METER_EBU)                                # k_kw
    _both_ways(M, x, calls, M.METER_TPBALLIST | M.METER_DR14 | M.METER_KMETER, chunk_streams=7)
    _both_ways(M, x, calls, M.METER_TRUEPEAK, fs=44100.0, tune_segments=2, chunk_streams=36)   # 36 + 1


def test_chunked_host_path_integer_meters(M):
    T = 50000
    S = 11
    x = np.stack([sig.g5(T, 4242 + s) for s in range(S)])
    x[3, 100] = 1.5                                                     # outside the SDH's bins: the skipped-sample regime
    _both_ways(M, x, [20000, 1, T - 20001], M.METER_BITSTATS | M.METER_SIGDIST, chunk_streams=3)
