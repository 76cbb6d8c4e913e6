"""mtr_engine_reduce: the job's one collective, RCCL behind the C ABI (include/mtr_engine.h).

  * one rank (any box): a world-of-one communicator — the reduction must be the identity on the aggregate;
  * two ranks (skipped where the box has one GPU): engine -> aggregate -> ncclAllReduce -> mtr_hist_loudness on
    two devices against ONE engine over all the streams: identical histograms, peaks and programme record."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_world_of_one_is_the_identity():
    import torch
    import meters.lv2_amd as M
    S, T, fs = 16, 48000 * 3, 48000.0
    buf = torch.empty((S, T, 2), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    M.synth_fill_device(buf.data_ptr(), S, T, T, 31, fs, 1, st)
    h0 = torch.zeros(2 * 751, dtype=torch.int32, device="cuda")
    m0 = torch.zeros(4, dtype=torch.float32, device="cuda")
    h1, m1 = torch.zeros_like(h0), torch.zeros_like(m0)
    with M.Comm(0, 1, M.comm_unique_id(), 0) as comm, M.Engine(S, fs, M.METER_EBU | M.METER_TRUEPEAK) as e:
        # what RCCL itself says about the communicator (ncclCommCount / ncclCommCuDevice): bench.py puts it into the job's line
        assert comm.nranks() == 1 and comm.device() == 0
        e.integr_start()
        e.process_device(buf.data_ptr(), T, T, st)
        e.aggregate_device(h0.data_ptr(), m0.data_ptr(), st)
        e.reduce(comm, h1.data_ptr(), m1.data_ptr(), st)
        torch.cuda.synchronize()
    assert int(h0.sum()) > 0 and torch.equal(h0, h1) and torch.equal(m0, m1)
    with pytest.raises(M.EngineError):
        M.Comm(2, 2, M.comm_unique_id(), 0)                   # rank outside the world


def test_world_of_one_with_a_deadline():
    """mtr_comm_init_timeout / mtr_comm_probe / mtr_engine_reduce on a NON-BLOCKING communicator (ncclCommInitRankConfig with
    blocking = 0, every call polled through ncclCommGetAsyncError): the path bench.py takes at N > 1, on the one GPU a box has."""
    import torch
    import meters.lv2_amd as M
    S, T, fs = 16, 48000 * 2, 48000.0
    buf = torch.empty((S, T, 2), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    M.synth_fill_device(buf.data_ptr(), S, T, T, 33, fs, 1, st)
    h0 = torch.zeros(2 * 751, dtype=torch.int32, device="cuda")
    m0 = torch.zeros(4, dtype=torch.float32, device="cuda")
    h1, m1 = torch.zeros_like(h0), torch.zeros_like(m0)
    assert M.engine.rccl_version() >= 21800
    with M.Comm(0, 1, M.comm_unique_id(), 0, timeout_ms=60000) as comm, M.Engine(S, fs, M.METER_EBU | M.METER_TRUEPEAK) as e:
        assert 0 < comm.init_ms < 60000
        assert 0 < comm.probe(30000) < 30000                  # the first collective, bounded
        assert comm.nranks() == 1 and comm.device() == 0
        comm.set_timeout(20000)
        e.integr_start()
        for _ in range(3):
            e.process_device(buf.data_ptr(), T, T, st)
            e.aggregate_device(h0.data_ptr(), m0.data_ptr(), st)
            e.reduce(comm, h1.data_ptr(), m1.data_ptr(), st)
            torch.cuda.synchronize()
            assert int(h0.sum()) > 0 and torch.equal(h0, h1) and torch.equal(m0, m1)
    with M.Comm(0, 1, M.comm_unique_id(), 0) as blocking:
        with pytest.raises(M.EngineError):
            blocking.set_timeout(1000)                        # a blocking communicator has no deadline to set
    closed = M.Comm(0, 1, M.comm_unique_id(), 0, timeout_ms=20000)
    closed.close()                                            # (finalised and destroyed, not aborted: ADVICE r5)
    closed.close()


def test_two_ranks_against_one_engine():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (hipGetDeviceCount() < 2 on this box)")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29631", os.path.join(HERE, "_reduce_worker.py")],
                         env=env, capture_output=True, text=True, timeout=600)
    assert "REDUCE_OK world=2" in out.stdout, (out.stdout[-2000:], out.stderr[-2000:])
