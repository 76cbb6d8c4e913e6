"""The N>1 host path on CPU: world_size-2 gloo processes shard the streams, build their per-rank
aggregates (here from the oracle, standing in for mtr_engine_aggregate_device), all-reduce them
with meters.lv2_amd.dist and must arrive at the single-process programme record."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _aggregate(streams):
    """hist[2*751] int32 (sum) and max[4] float32 over a list of stream indices, via the oracle."""
    sys.path[:0] = [ROOT, HERE]
    import _signals as sig
    from _oracle import Oracle
    orc = Oracle()
    hist = np.zeros((2, 751), np.int32)
    mx = np.array([0, 0, -200, -200], np.float32)
    for s in streams:
        x = sig.g2(48000 * 8, 900 + s)
        r = orc.ebu(x, 48000.0, 2400)
        tp = orc.tp(x, 48000.0, 8192)
        hist[0] += r["hist_M"]
        hist[1] += r["hist_S"]
        mx = np.maximum(mx, [tp[0], tp[1], r["out9"][1], r["out9"][3]]).astype(np.float32)
    return hist.reshape(-1), mx


def _worker(rank, world, port, n_total, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path[:0] = [ROOT, HERE]
    from meters.lv2_amd import dist as mdist
    first, count = mdist.shard(n_total, world, rank)
    h, m = _aggregate(range(first, first + count))
    th, tm = torch.from_numpy(h.copy()), torch.from_numpy(m.copy())
    mdist.all_reduce_aggregate(th, tm)
    rec = mdist.programme_summary(th, tm)
    if rank == 0:
        torch.save(dict(rec=rec, hist=th, mx=tm, shard=(first, count)), out)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_partition():
    from meters.lv2_amd import dist as mdist
    for n, w in ((65536, 8), (10, 3), (7, 8), (1, 2)):
        parts = [mdist.shard(n, w, r) for r in range(w)]
        assert sum(c for _, c in parts) == n
        assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(w - 1))
        assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


@pytest.mark.timeout(300)
def test_two_rank_reduce_equals_single_process(tmp_path):
    from meters.lv2_amd import dist as mdist
    n_total, world = 5, 2
    out = str(tmp_path / "r0.pt")
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, n_total, out), nprocs=world, join=True)
    got = torch.load(out, weights_only=False)
    h, m = _aggregate(range(n_total))
    assert np.array_equal(got["hist"].numpy(), h)
    assert np.array_equal(got["mx"].numpy(), m)
    want = mdist.programme_summary(torch.from_numpy(h), torch.from_numpy(m))
    assert got["rec"] == want
    assert got["shard"] == (0, 3)
    assert -30 < want["integrated"] < 0 and want["hist_M_count"] == n_total * 80


class _FakeComm:
    closed = 0
    probed = 0
    probe_fails = False

    def probe(self, timeout_ms=0):
        _FakeComm.probed += 1
        if self.probe_fails:
            raise RuntimeError("mtr_comm_probe: the first all-reduce did not finish within %d ms" % timeout_ms)

    def close(self):
        _FakeComm.closed += 1


def _agree_worker(rank, world, port, scenario, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path[:0] = [ROOT, HERE]
    from meters.lv2_amd import dist as mdist

    def make():
        if scenario == "one-fails" and rank == 1 or scenario == "all-fail":
            raise RuntimeError("ncclCommInitRank: refused (rank %d)" % rank)
        if scenario == "one-sleeps":
            # rank 1 sleeps through the others' deadline; mtr_comm_init_timeout gives rank 0 up after its own (here: 0.5 s),
            # and rank 1, arriving late, finds nobody and times out as well
            import time
            time.sleep(4.0 if rank == 1 else 0.5)
            raise RuntimeError("ncclCommInitRankConfig: no answer from RCCL within 500 ms (communicator aborted)")
        c = _FakeComm()
        c.probe_fails = scenario == "probe-fails" and rank == 1
        return c

    import time
    t0 = time.monotonic()
    comm, group, desc = mdist.agree_on_collective(rank, world, make, allow_nccl=False, probe_timeout_s=2.0)
    took = time.monotonic() - t0
    # whatever was agreed, the ranks reduce the same way: a collective on the default group must still work
    t = torch.tensor([rank + 1], dtype=torch.int32)
    mdist.all_reduce_aggregate(t, torch.zeros(1), group)
    torch.save(dict(comm=comm is not None, group=group is not None, desc=desc, closed=_FakeComm.closed, sum=int(t.item()),
                    probed=_FakeComm.probed, took=took), out + ".%d" % rank)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("scenario", ["all-succeed", "one-fails", "all-fail", "one-sleeps", "probe-fails"])
def test_ranks_agree_on_how_they_reduce(tmp_path, scenario):
    """bench.py's negotiation at N > 1 (meters.lv2_amd.dist.agree_on_collective), on two gloo ranks without a GPU: the
    engine communicator is kept only if EVERY rank built one; a rank whose mtr_comm_init succeeded while another's failed
    closes it and falls back with the others (it would otherwise wait inside ncclAllReduce for a peer that never comes)."""
    out = str(tmp_path / scenario)
    port = 31500 + os.getpid() % 2000
    mp.spawn(_agree_worker, args=(2, port, scenario, out), nprocs=2, join=True)
    r = [torch.load(out + ".%d" % k, weights_only=False) for k in range(2)]
    assert r[0]["sum"] == r[1]["sum"] == 3
    if scenario == "all-succeed":
        assert all(x["comm"] and not x["group"] and "RCCL behind the C ABI" in x["desc"] for x in r)
        assert r[0]["closed"] == r[1]["closed"] == 0
        assert r[0]["probed"] == r[1]["probed"] == 1           # the job's first collective ran before anything else
    elif scenario == "one-sleeps":
        # a rank that sleeps through the communicator's creation: the others time out (mtr_comm_init_timeout), every rank
        # votes, and the job goes on over the fallback — in the sleeper's time, not the control plane's 30 minutes
        assert not any(x["comm"] or x["group"] for x in r)
        assert all("gloo" in x["desc"] and "mtr_comm_init failed on a rank" in x["desc"] and "no answer from RCCL" in x["desc"] for x in r)
        assert all(x["took"] < 30 for x in r)
    elif scenario == "probe-fails":
        # every rank built a communicator, the FIRST collective on it hung on one: all give theirs up and fall back together
        assert not any(x["comm"] or x["group"] for x in r)
        assert all("gloo" in x["desc"] and "first all-reduce on the engine's communicator failed" in x["desc"] for x in r)
        assert r[0]["closed"] == r[1]["closed"] == 1
    else:
        assert not any(x["comm"] or x["group"] for x in r)
        assert all("gloo" in x["desc"] and "mtr_comm_init failed on a rank" in x["desc"] for x in r)
        assert r[0]["closed"] == (1 if scenario == "one-fails" else 0) and r[1]["closed"] == 0   # rank 0 built one, and gave it up
