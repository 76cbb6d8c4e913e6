"""mtr_engine_state_export / mtr_engine_state_import (include/mtr_engine.h; SURVEY.md 5 "checkpoint / resume"; VERDICT r4
item 5): everything a stream carries from call to call travels as one opaque blob — out of an engine, into another engine
of the same configuration with a DIFFERENT number of streams and a DIFFERENT slot, in this process or (below) in another —
and processing continues BIT FOR BIT as if it had never stopped: every result of every meter, for both fused kernels.

The reference has nothing like it (it persists one UI word, src/ebulv2.cc:514-553): this is what lets a multi-hour batch be
checkpointed and the streams of a job be re-sharded between its ranks."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import _signals as sig
from test_gpu_hostpath import _records

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def M():
    import meters.lv2_amd as m
    return m


def _same(a, b):
    assert a.keys() == b.keys()
    for k in a:
        assert np.array_equal(a[k], b[k], equal_nan=True), k


def _start(M, e, meters):
    if meters & M.METER_EBU:
        e.integr_start()


def _straight(M, x, calls, meters, fs, C, **kw):
    """ONE engine over all the streams and all the calls: the record after every call."""
    import torch
    dev = torch.from_numpy(x).cuda()
    st = torch.cuda.current_stream().cuda_stream
    recs = []
    with M.Engine(x.shape[0], fs, meters, n_channels=C, **kw) as e:
        _start(M, e, meters)
        pos = 0
        for n in calls:
            e.process_device(dev.data_ptr() + pos * C * 4, n, x.shape[1], st)
            recs.append(_records(M, e, meters))
            pos += n
        seg = e.seg_stats()[0]
    return recs, seg


def _rows(rec, sel):
    """The streams `sel` of a record (every array is [stream, ...]; fragment powers belong to the last call only)."""
    return {k: v[sel] for k, v in rec.items()}


def _checkpointed(M, x, calls, k_stop, meters, fs, C, **kw):
    """The same job, stopped after call k_stop: its streams leave in two blobs (the batch cut at an odd place), the engine
    is DESTROYED, and two new engines of other sizes take them into other slots and process the rest."""
    import torch
    S = x.shape[0]
    cut = S // 3 + 1
    dev = torch.from_numpy(x).cuda()
    st = torch.cuda.current_stream().cuda_stream
    with M.Engine(S, fs, meters, n_channels=C, **kw) as e:
        _start(M, e, meters)
        pos = 0
        for n in calls[:k_stop]:
            e.process_device(dev.data_ptr() + pos * C * 4, n, x.shape[1], st)
            before = _records(M, e, meters)                                # (read after every call, as the straight run is: Kmeterdsp::read arms a flag)
            pos += n
        assert e.state_bytes(cut) == len(e.state_export(0, cut))
        blob_a, blob_b = e.state_export(0, cut), e.state_export(cut, S - cut)
    # engine A': more streams than the blob, the blob's streams in slots [2, 2 + cut); engine B': exactly the rest, slot 0
    pad = 3
    xa = np.concatenate([np.zeros((2,) + x.shape[1:], np.float32), x[:cut], np.zeros((pad - 2,) + x.shape[1:], np.float32)])
    da, db = torch.from_numpy(xa).cuda(), torch.from_numpy(np.ascontiguousarray(x[cut:])).cuda()
    out = []
    with M.Engine(cut + pad, fs, meters, n_channels=C, **kw) as ea, M.Engine(S - cut, fs, meters, n_channels=C, **kw) as eb:
        assert ea.state_import(blob_a, first=2) == cut and eb.state_import(blob_b) == S - cut
        # what came out is what went in, before anything is processed
        ra, rb = _records(M, ea, meters), _records(M, eb, meters)
        for k, v in before.items():
            if k == "frag":
                continue                                                   # (the last call's fragment powers are a diagnostic of that call)
            assert np.array_equal(np.concatenate([ra[k][2:2 + cut], rb[k]]), v, equal_nan=True), k
        p = pos
        for n in calls[k_stop:]:
            ea.process_device(da.data_ptr() + p * C * 4, n, x.shape[1], st)
            eb.process_device(db.data_ptr() + p * C * 4, n, x.shape[1], st)
            ra, rb = _records(M, ea, meters), _records(M, eb, meters)
            out.append({k: np.concatenate([ra[k][2:2 + cut], rb[k]]) for k in ra})
            p += n
        seg = ea.seg_stats()[0]
    return out, seg


CASES = [
    # (meters, channels, fs, engine options, signal, uneven calls (the checkpoint falls INSIDE a 50 ms fragment, a DR-14 window, a bank toggle))
    ("ebu+tp: k_kwtp16", "EBU|TRUEPEAK", 2, 48000.0, {"tune_layout": 6, "tune_segments": 1}, "g2", [2400 * 9 + 1000, 2400 * 6 + 333, 1500, 2400 * 12 + 67]),
    ("ebu+tp: k_seg", "EBU|TRUEPEAK", 2, 48000.0, {"tune_segments": 3}, "g2", [2400 * 9 + 1000, 2400 * 6 + 333, 1500, 2400 * 12 + 67]),
    ("ebu+tp: k_seg at 44.1 kHz", "EBU|TRUEPEAK", 2, 44100.0, {"tune_segments": 2}, "g2", [2205 * 8, 2205 * 7 + 5, 2205 * 9 + 2200]),
    ("ebu: k_kw", "EBU", 2, 48000.0, {}, "g2", [2400 * 9 + 1000, 2400 * 6 + 333, 1500, 2400 * 12 + 67]),
    ("bank + ballistics + DR-14 + K-meter", "SPECTR30|TPBALLIST|DR14|KMETER", 2, 48000.0, {}, "g2", [30011, 7, 48000 * 3 + 5, 20000]),
    ("mono ballistics + bank", "SPECTR30|TPBALLIST", 1, 48000.0, {}, "mono", [30011, 7, 40000]),
    ("integer meters", "BITSTATS|SIGDIST", 1, 48000.0, {}, "g5", [20000, 1, 29999]),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_export_destroy_import_continue_is_one_run_bit_for_bit(M, case):
    _, names, C, fs, kw, kind, calls = case
    meters = 0
    for n in names.split("|"):
        meters |= getattr(M, "METER_" + n)
    T, S = sum(calls), 13
    if kind == "g2":
        x = np.stack([sig.g2(T, 500 + s, fs) * np.float32(2.0 ** -(s % 3)) for s in range(S)])
    elif kind == "mono":
        x = np.stack([sig.g2(T, 600 + s, fs)[:, 0] * np.float32(2.0 ** -(s % 3)) for s in range(S)])
    else:
        x = np.stack([sig.g5(T, 4300 + s) for s in range(S)])
        x[4, 100] = 1.5                                                    # the SDH's skipped-sample regime, before the checkpoint
    x = np.ascontiguousarray(x, np.float32)
    want, seg_w = _straight(M, x, calls, meters, fs, C, **kw)
    for k_stop in (1, 2):
        got, seg_g = _checkpointed(M, x, calls, k_stop, meters, fs, C, **kw)
        for a, b in zip(want[k_stop:], got):
            _same(a, b)
    if kw.get("tune_segments") and not kw.get("tune_layout"):
        assert seg_w >= 2 and seg_g >= 1                                   # both sides of the checkpoint went through k_seg
    else:
        assert seg_w == 0 and seg_g == 0


def test_import_refuses_what_does_not_fit(M):
    import torch
    fs, T = 48000.0, 5000
    x = np.stack([sig.g2(T, 50 + s) for s in range(4)])
    dev = torch.from_numpy(x).cuda()
    st = torch.cuda.current_stream().cuda_stream
    with M.Engine(4, fs, M.METER_EBU | M.METER_TRUEPEAK) as e:
        e.process_device(dev.data_ptr(), 1000, T, st)
        blob = e.state_export(1, 2)
        assert len(blob) == e.state_bytes(2)
        with pytest.raises(M.EngineError):
            e.state_export(3, 2)                                           # stream range
        # another configuration: meters, rate
        for other in (dict(meters=M.METER_EBU), dict(sample_rate=44100.0)):
            with M.Engine(4, other.get("sample_rate", fs), other.get("meters", M.METER_EBU | M.METER_TRUEPEAK)) as o:
                with pytest.raises(M.EngineError) as ei:
                    o.state_import(blob)
                assert ei.value.code == M.engine.ERR_STATE
        # not a blob, a truncated blob, a blob that does not fit the slots
        with M.Engine(2, fs, M.METER_EBU | M.METER_TRUEPEAK) as o:
            for bad in (b"", b"x" * 100, blob[:-1], bytes(len(blob))):
                with pytest.raises(M.EngineError) as ei:
                    o.state_import(bad)
                assert ei.value.code == M.engine.ERR_STATE
            with pytest.raises(M.EngineError):
                o.state_import(blob, first=1)                              # two streams into slots [1, 3) of two
            assert o.state_import(blob) == 2                               # a fresh engine takes the blob's cursors (1000 frames into a fragment)
            o.process_device(dev.data_ptr() + 1000 * 8, 500, T, st)
            with pytest.raises(M.EngineError) as ei:
                o.state_import(blob)                                       # ... and now stands somewhere else
            assert ei.value.code == M.engine.ERR_STATE
        # the same engine, at the same cursors: allowed (streams swap slots inside a job)
        was = [(r.truepeak[0], r.truepeak[1], r.loudness_M) for r in e.results()]
        e.state_import(blob, first=0)
        now = [(r.truepeak[0], r.truepeak[1], r.loudness_M) for r in e.results()]
        assert now == [was[1], was[2], was[2], was[3]] and len(set(was)) == 4


def test_import_checks_the_blob_before_it_trusts_it(M):
    """a checkpoint that rotted — a flipped bit in the payload, a fragment cursor of 0 or beyond the fragment, an
    open DR-14 window longer than a window — is refused with MTR_ERR_STATE and leaves the engine untouched (still fresh: the
    good blob goes in afterwards and processing continues bit for bit); and the documented asymmetry: a fresh engine on which
    integr_start () was called takes the BLOB's integration flag."""
    import struct
    import torch
    fs, T = 48000.0, 9000
    meters = M.METER_EBU | M.METER_TRUEPEAK
    x = np.stack([sig.g2(T, 80 + s) for s in range(3)])
    dev = torch.from_numpy(x).cuda()
    st = torch.cuda.current_stream().cuda_stream
    with M.Engine(3, fs, meters) as e:
        e.process_device(dev.data_ptr(), 3000, T, st)                      # integration OFF (never started)
        blob = e.state_export()
        e.process_device(dev.data_ptr() + 3000 * 8, 6000, T, st)
        want = _records(M, e, meters)
    # header: magic, version, header_bytes, meters, n_channels, rate, count, per_stream, stream_state_bytes, frcnt, integr, omega, dr_scnt, fnv
    hdr = struct.Struct("<IIIIIfIIIIIfQQ")
    f = list(hdr.unpack_from(blob))
    assert f[1] == 2 and f[9] == 2400 - 3000 % 2400 and f[10] == 0

    def with_field(i, v):
        g = list(f); g[i] = v
        return hdr.pack(*g) + blob[hdr.size:]
    rotten = [bytearray(blob) for _ in range(2)]
    rotten[0][hdr.size + 5] ^= 0x10                                        # one bit of stream 0's K-filter state
    rotten[1][-1] ^= 0x01                                                  # the last byte of the payload
    bad = [bytes(r) for r in rotten] + [with_field(9, 0), with_field(9, 2401), with_field(9, 1 << 31), with_field(10, 7),
                                        with_field(11, 2.0), with_field(12, 1 << 40)]
    with M.Engine(3, fs, meters) as o:
        o.integr_start()                                                   # overridden by the import below: the blob says off
        for b in bad:
            with pytest.raises(M.EngineError) as ei:
                o.state_import(b)
            assert ei.value.code == M.engine.ERR_STATE
        assert o.state_import(blob) == 3                                   # still a fresh engine: nothing above moved it
        o.process_device(dev.data_ptr() + 3000 * 8, 6000, T, st)
        got = _records(M, o, meters)
    for k in want:
        assert np.array_equal(want[k], got[k], equal_nan=True), k
    assert want["counts"].sum() == 0                                       # integration stayed off, as in the blob


WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "tests")]
import meters.lv2_amd as M
from meters.lv2_amd import dist as mdist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
import datetime
dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=120))
torch.cuda.set_device(0)
S, T, fs = 24, 48000 * 2, 48000.0                      # streams per rank; every rank holds the WHOLE job's audio, meters its own shard
st = torch.cuda.current_stream().cuda_stream
full = torch.empty((2 * S, 3 * T, 2), dtype=torch.float32, device="cuda")
M.synth_fill_device(full.data_ptr(), 2 * S, 3 * T, 3 * T, 777, fs, 1, st)
first, count = mdist.shard(2 * S, world, rank)
half = S // 2
own = list(range(first, first + count))               # job-wide stream ids in this engine's slots
def process(e, ids, pos, n):
    # the engine's slots hold arbitrary job streams after the swap: gather them into a contiguous batch for the call
    batch = full[ids, pos:pos + n].contiguous()
    e.process_device(batch.data_ptr(), n, n, st)
    torch.cuda.synchronize()
with M.Engine(S, fs, M.METER_EBU | M.METER_TRUEPEAK, **%(kw)r) as e:
    e.integr_start()
    process(e, own, 0, T + 1000)                      # the job stops INSIDE a fragment
    # mid-job: every rank gives the second half of its streams to the other rank and takes the other's
    mine = e.state_export(half, S - half)
    blobs = [None, None]
    dist.all_gather_object(blobs, mine)
    e.state_import(blobs[1 - rank], first=half)
    own = own[:half] + list(range((1 - rank) * S + half, (1 - rank) * S + S))
    process(e, own, T + 1000, 2 * T - 1000)
    hist = torch.zeros(2 * 751, dtype=torch.int32, device="cuda")
    mx = torch.zeros(4, dtype=torch.float32, device="cuda")
    e.aggregate_device(hist.data_ptr(), mx.data_ptr(), st)
    torch.cuda.synchronize()
    per = {own[i]: (r.integrated, r.truepeak[0], r.truepeak[1], r.hist_M_count, r.loudness_S) for i, r in enumerate(e.results())}
hist, mx = hist.cpu(), mx.cpu()
mdist.all_reduce_aggregate(hist, mx)
allper = [None, None]
dist.all_gather_object(allper, per)
if rank == 0:
    import json
    merged = {}
    for d in allper:
        merged.update(d)
    print("STATE_SWAP " + json.dumps({"programme": mdist.programme_summary(hist, mx), "per": [merged[i] for i in range(2 * S)]}), flush=True)
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("segments", [0, 2], ids=["k_kwtp16", "k_seg"])
def test_two_ranks_swap_half_their_streams_mid_job(M, tmp_path, segments):
    """Two gloo ranks (sharing this box's one GPU) meter a job of 48 streams, 24 each; in mid-job — inside a 50 ms fragment —
    each hands the second half of its streams to the other as a state blob over the control plane and goes on with what it
    received.  The job's programme record (summed histograms -> mtr_hist_loudness, max of peaks) and every stream's own
    results equal ONE engine over all 48 streams fed the same two calls, bit for bit."""
    import torch
    from meters.lv2_amd import dist as mdist
    script = tmp_path / "swap_worker.py"
    # (the segmentation is pinned: left to itself the engine picks it from the batch size, and 24 and 48 streams need not agree)
    kw = {"tune_segments": 2} if segments else {"tune_layout": 6, "tune_segments": 2}
    script.write_text(WORKER % dict(root=ROOT, kw=kw))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(29700 + os.getpid() % 200), str(script)], env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("STATE_SWAP ")]
    assert out.returncode == 0 and len(lines) == 1, (out.stdout[-2000:], out.stderr[-4000:])
    got = json.loads(lines[0][len("STATE_SWAP "):])
    S, T, fs = 48, 48000 * 2, 48000.0
    st = torch.cuda.current_stream().cuda_stream
    full = torch.empty((S, 3 * T, 2), dtype=torch.float32, device="cuda")
    M.synth_fill_device(full.data_ptr(), S, 3 * T, 3 * T, 777, fs, 1, st)
    hist = torch.zeros(2 * 751, dtype=torch.int32, device="cuda")
    mx = torch.zeros(4, dtype=torch.float32, device="cuda")
    with M.Engine(S, fs, M.METER_EBU | M.METER_TRUEPEAK, **kw) as e:
        e.integr_start()
        for pos, n in ((0, T + 1000), (T + 1000, 2 * T - 1000)):
            batch = full[:, pos:pos + n].contiguous()
            e.process_device(batch.data_ptr(), n, n, st)
            torch.cuda.synchronize()
        e.aggregate_device(hist.data_ptr(), mx.data_ptr(), st)
        torch.cuda.synchronize()
        per = [[r.integrated, r.truepeak[0], r.truepeak[1], r.hist_M_count, r.loudness_S] for r in e.results()]
    want = mdist.programme_summary(hist, mx)
    for k, v in want.items():
        g = got["programme"][k]
        assert (list(g) == list(v)) if isinstance(v, tuple) else (g == v), (k, g, v)
    assert got["per"] == per
