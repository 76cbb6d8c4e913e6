"""Worker of tests/test_gpu_reduce.py: one rank per GPU.  Every rank meters its block of the same seeded batch,
the aggregates are reduced by mtr_engine_reduce (RCCL inside the C ABI), rank 0 compares the programme record with
ONE engine over all the streams.  Launched by torch.distributed.run; prints REDUCE_OK on success."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import meters.lv2_amd as M  # noqa: E402
from meters.lv2_amd import dist as mdist  # noqa: E402


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")                            # carries the RCCL id only
    torch.cuda.set_device(local)
    S_total, T, fs = 96, 48000 * 4, 48000.0
    first, count = mdist.shard(S_total, world, rank)
    buf = torch.empty((count, T, 2), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    M.synth_fill_device(buf.data_ptr(), count, T, T, 900 + first, fs, 1, st)
    hist = torch.zeros(2 * 751, dtype=torch.int32, device="cuda")
    mx = torch.zeros(4, dtype=torch.float32, device="cuda")
    comm = mdist.make_comm(rank, world, local)
    assert comm.nranks() == world and comm.device() == local, (comm.nranks(), comm.device())    # RCCL's own view: N ranks, this rank's device
    with M.Engine(count, fs, M.METER_EBU | M.METER_TRUEPEAK, device=local) as e:
        e.integr_start()
        e.set_deferred_tail(2)                                      # the gate and the all-reduce on the engine's side stream, as a batch job runs them
        e.process_device(buf.data_ptr(), T, T, st)
        e.reduce(comm, hist.data_ptr(), mx.data_ptr(), st)
        e.sync()                                                   # (both streams)
        torch.cuda.synchronize()
    got = mdist.programme_summary(hist, mx)
    comm.close()
    if rank == 0:
        full = torch.empty((S_total, T, 2), dtype=torch.float32, device="cuda")
        M.synth_fill_device(full.data_ptr(), S_total, T, T, 900, fs, 1, st)
        h1 = torch.zeros(2 * 751, dtype=torch.int32, device="cuda")
        m1 = torch.zeros(4, dtype=torch.float32, device="cuda")
        with M.Engine(S_total, fs, M.METER_EBU | M.METER_TRUEPEAK, device=local) as e:
            e.integr_start()
            e.process_device(full.data_ptr(), T, T, st)
            e.aggregate_device(h1.data_ptr(), m1.data_ptr(), st)
            torch.cuda.synchronize()
        want = mdist.programme_summary(h1, m1)
        assert torch.equal(hist, h1), "summed histograms differ"
        assert torch.equal(mx, m1), (mx, m1)
        assert got == want, (got, want)
        print("REDUCE_OK world=%d integrated=%.2f LUFS" % (world, got["integrated"]), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
