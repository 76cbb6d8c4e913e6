"""The deferred tail (include/mtr_engine.h: mtr_engine_set_deferred_tail / mtr_engine_join).

k_gate — the once-per-fragment bookkeeping of Ebu_r128_proc::process (ebumeter/ebu_r128_proc.cc:217-244) — and the job's
reduction may run on an engine-owned side stream beside the NEXT call's fused kernel.  Nothing about the arithmetic changes:
the same kernels see the same inputs, fragments are inserted in fragment order, so EVERY result — and the whole exported
state, byte for byte — must be that of the serial order.  The calls below are queued back to back WITHOUT a host wait in
between (a getter would wait for both streams and hide a missing dependency), with sizes that change the plan from call to
call, start inside fragments and go through both fused kernels."""
import numpy as np
import pytest

import _signals as sig
from test_gpu_hostpath import _records

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def M():
    import meters.lv2_amd as m
    return m


def _run(M, x, calls, meters, mode, fs=48000.0, reduce_every=0, sync_every=0, **kw):
    """Queue `calls` on one engine in tail mode `mode`; returns (record, state blob, aggregates per reduce, deferred calls)."""
    import torch
    S, T = x.shape[0], x.shape[1]
    dev = torch.from_numpy(x).cuda()
    st = torch.cuda.current_stream().cuda_stream
    aggs = []
    with M.Comm(0, 1, M.comm_unique_id(), 0) as comm, M.Engine(S, fs, meters, **kw) as e:
        e.set_deferred_tail(mode)
        if meters & M.METER_EBU:
            e.integr_start()
        pos = 0
        for k, n in enumerate(calls):
            e.process_device(dev.data_ptr() + pos * 8, n, T, st)
            pos += n
            if reduce_every and (k + 1) % reduce_every == 0:
                h = torch.zeros(2 * 751, dtype=torch.int32, device="cuda")
                m = torch.zeros(4, dtype=torch.float32, device="cuda")
                e.reduce(comm, h.data_ptr(), m.data_ptr(), st)
                aggs.append((h, m))
            if sync_every and (k + 1) % sync_every == 0:
                e.sync()
        assert pos <= T
        rec = _records(M, e, meters)                       # (waits for both streams)
        blob = e.state_export()
        nd = e.deferred_calls()
        e.sync()
        aggs = [(h.cpu().numpy(), m.cpu().numpy()) for h, m in aggs]
    return rec, blob, aggs, nd


def _same(a, b):
    assert a.keys() == b.keys()
    for k in a:
        assert np.array_equal(a[k], b[k], equal_nan=True), k


CALLS = [4800, 2400, 1000, 7777, 48000, 1, 2399, 24000, 9600, 123, 14400]          # 154 500 frames, most starting inside a fragment


@pytest.mark.parametrize("meters_name", ["ebu+tp", "ebu", "tp", "ebu+tp+tpb"])
@pytest.mark.parametrize("kw", [{}, {"tune_segments": 3}], ids=["k_kwtp16", "k_seg"])
def test_deferred_tail_is_bit_for_bit_the_serial_order(M, meters_name, kw):
    meters = {"ebu+tp": M.METER_EBU | M.METER_TRUEPEAK, "ebu": M.METER_EBU, "tp": M.METER_TRUEPEAK,
              "ebu+tp+tpb": M.METER_EBU | M.METER_TRUEPEAK | M.METER_TPBALLIST}[meters_name]
    if kw and not meters & M.METER_TRUEPEAK:
        pytest.skip("k_seg is a true-peak kernel")
    S, T = 19, sum(CALLS)
    x = np.stack([sig.lcg_noise(T, 500 + s, 2.0 ** -(s % 4)) * (1.0 + 0.5 * np.sin(np.arange(T) / 9000.0 + s))[:, None].astype(np.float32)
                  for s in range(S)])
    serial = _run(M, x, CALLS, meters, 1, reduce_every=2, **kw)
    deferred = _run(M, x, CALLS, meters, 2, reduce_every=2, **kw)
    assert serial[3] == 0 and deferred[3] == len(CALLS)
    _same(serial[0], deferred[0])
    assert serial[1] == deferred[1]                                    # the whole exported state, byte for byte
    assert len(serial[2]) == len(deferred[2]) == len(CALLS) // 2
    for (h0, m0), (h1, m1) in zip(serial[2], deferred[2]):              # every reduction saw exactly the calls in front of it
        assert np.array_equal(h0, h1) and np.array_equal(m0, m1)
    assert serial[2][-1][0].sum() > 0 or not meters & M.METER_EBU
    assert serial[2][-1][1][:2].max() > 0 or not meters & M.METER_TRUEPEAK


def test_mixed_modes_and_resets(M):
    """Deferred calls, then serial ones, resets of the integration and of the peaks in between (they clear what a deferred
    gate may still be writing), the mode switched on a live engine: the record of the all-serial run."""
    import torch
    S, T, fs = 7, 48000 * 3, 48000.0
    x = np.stack([sig.g2(T, 900 + s) for s in range(S)])
    dev = torch.from_numpy(x).cuda()
    st = torch.cuda.current_stream().cuda_stream
    meters = M.METER_EBU | M.METER_TRUEPEAK
    script = [(2, 30000), (2, 4000), ("integr_reset",), (1, 24000), (2, 10000), ("truepeak_reset",), (2, 20000), (1, 777), (2, 48000), ("reset+start",), (2, 7223)]
    recs = []
    for force_serial in (True, False):
        with M.Engine(S, fs, meters) as e:
            e.integr_start()
            pos = 0
            for step in script:
                if step[0] == "integr_reset":
                    e.integr_reset()
                elif step[0] == "truepeak_reset":
                    e.truepeak_reset()
                elif step[0] == "reset+start":
                    e.reset(); e.integr_start()
                else:
                    e.set_deferred_tail(1 if force_serial else step[0])
                    e.process_device(dev.data_ptr() + pos * 8, step[1], T, st)
                    pos += step[1]
            recs.append((_records(M, e, meters), e.state_export()))
    _same(recs[0][0], recs[1][0])
    assert recs[0][1] == recs[1][1]


def test_auto_defers_batches_only_and_join_orders_a_stream(M):
    """mode 0 defers where it was measured to pay (profiles/r06_tail.md): a batch whose whole fragments go through k_seg, in an
    engine of EBU / TRUEPEAK only — not an LV2-sized call, not the EBU-only kernel (HBM-bound: the gate beside it costs more than
    in front of it), not an engine that also runs the bank.  mtr_engine_join makes the caller's stream wait for the side stream: a
    device-to-device copy of reduce ()'s buffer queued behind it sees the finished sum."""
    import torch
    S, T, fs = 8192, 96000, 48000.0                                    # 786 M stream-frames per call: k_seg's kind of batch
    both = M.METER_EBU | M.METER_TRUEPEAK
    assert M.plan_query(S, T, fs, both)["uses_seg"] == 1
    buf = torch.empty((S, T, 2), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    M.synth_fill_device(buf.data_ptr(), S, T, T, 41, fs, 1, st)
    h = torch.zeros(2 * 751, dtype=torch.int32, device="cuda")
    m = torch.zeros(4, dtype=torch.float32, device="cuda")
    with M.Comm(0, 1, M.comm_unique_id(), 0) as comm, M.Engine(S, fs, both) as e, M.Engine(S, fs, both) as ser:
        ser.set_deferred_tail(1)
        e.set_deferred_tail(0)                                         # auto, whatever MTR_TAIL_MODE says (the suite also runs under MTR_TAIL_MODE=2)
        e.integr_start(); ser.integr_start()
        copies = []
        for _ in range(4):
            e.process_device(buf.data_ptr(), T, T, st)
            e.reduce(comm, h.data_ptr(), m.data_ptr(), st)
            e.join(st)
            copies.append((h.clone(), m.clone()))                      # on `st`, behind the join: no host wait anywhere
        e.process_device(buf.data_ptr(), 1024, T, st)                  # an LV2-sized call on the same engine: serial
        assert e.deferred_calls() == 4
        want = []
        for _ in range(4):
            ser.process_device(buf.data_ptr(), T, T, st)
            hh, mm = torch.zeros_like(h), torch.zeros_like(m)
            ser.aggregate_device(hh.data_ptr(), mm.data_ptr(), st)
            want.append((hh, mm))
        ser.process_device(buf.data_ptr(), 1024, T, st)
        assert ser.deferred_calls() == 0
        torch.cuda.synchronize()
        for (a, b), (c, d) in zip(copies, want):
            assert torch.equal(a, c) and torch.equal(b, d)
        assert int(copies[-1][0].sum()) > int(copies[0][0].sum()) > 0
        _same(_records(M, e, both), _records(M, ser, both))
    for meters in (M.METER_EBU, both | M.METER_SPECTR30):              # the same batch: no deferral in auto, any in mode 2
        with M.Engine(S if meters == M.METER_EBU else 256, fs, meters) as e:
            e.set_deferred_tail(0)
            e.integr_start()
            e.process_device(buf.data_ptr(), T, T, st)
            assert e.deferred_calls() == 0
            e.set_deferred_tail(2)
            e.process_device(buf.data_ptr(), T, T, st)
            assert e.deferred_calls() == 1


def test_deferred_tail_on_two_caller_streams(M):
    """The caller moves to another stream between calls while a deferred gate is still queued."""
    import torch
    S, T, fs = 9, 48000 * 2, 48000.0
    x = np.stack([sig.g2(T, 70 + s) for s in range(S)])
    dev = torch.from_numpy(x).cuda()
    meters = M.METER_EBU | M.METER_TRUEPEAK
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    recs = []
    for mode in (1, 2):
        with M.Engine(S, fs, meters, tune_segments=2) as e:
            e.set_deferred_tail(mode)
            e.integr_start()
            pos = 0
            for k, n in enumerate([24000, 12000, 5000, 31000, 24000]):
                st = (s1, s2)[k & 1].cuda_stream
                e.process_device(dev.data_ptr() + pos * 8, n, T, st)
                pos += n
            recs.append((_records(M, e, meters), e.state_export()))
    _same(recs[0][0], recs[1][0])
    assert recs[0][1] == recs[1][1]


def test_lv2_shaped_blocks_with_a_forced_deferred_tail(M):
    """n_streams = 1 through mtr_engine_process_planar_host (one wait per block, the state copied back with it): forcing
    the tail onto the side stream must not let the snapshot overtake the gate."""
    T, fs = 48000, 48000.0
    x = sig.g2(T, 5)
    recs = []
    for mode in (1, 2):
        with M.Engine(1, fs, M.METER_EBU | M.METER_TRUEPEAK) as e:
            e.set_deferred_tail(mode)
            e.integr_start()
            seen = []
            for p in range(0, T, 1024):
                e.process_planar([x[p:p + 1024, 0], x[p:p + 1024, 1]])
                r = e.results()[0]
                seen.append((r.loudness_M, r.loudness_S, r.truepeak_call[0], r.truepeak[1]))
            recs.append(np.array(seen))
    assert np.array_equal(recs[0], recs[1])


def test_bad_mode_is_an_argument_error(M):
    with M.Engine(1, 48000.0, M.METER_EBU | M.METER_TRUEPEAK) as e:
        with pytest.raises(M.EngineError) as ei:
            e.set_deferred_tail(3)
        assert ei.value.code == M.engine.ERR_ARG
