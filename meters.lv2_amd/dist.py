"""Multi-GPU host logic: static stream partition + the one collective of the job.

Streams are independent, so rank r of W owns the block [first, first+count) and no audio ever
crosses xGMI.  The only exchange is the final reduction of the per-rank aggregates that
mtr_engine_aggregate_device() leaves in device memory: int32[2][751] loudness histograms (sum)
and float[4] = {tp_L, tp_R, maxloudn_M, maxloudn_S} (max).

On GPUs that reduction is mtr_engine_reduce(): RCCL INSIDE the C ABI (ncclAllReduce x2 on the engine's
own communicator, include/mtr_engine.h) — what a C host calls.  torch.distributed is plumbing here: it
carries the 128-byte RCCL id from rank 0 to the others (make_comm) and provides the barrier of bench.py;
all_reduce_aggregate() is the same reduction on torch tensors, kept for the CPU tests ("gloo").
"""
from . import engine as _engine


def shard(n_total, world_size, rank):
    """Block partition: (first, count) of rank's streams; counts differ by at most one."""
    base, extra = divmod(int(n_total), int(world_size))
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def all_reduce_aggregate(hist, maxv, group=None):
    """In-place all-reduce of the aggregate tensors across ranks. Returns (hist, maxv)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(hist, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(maxv, op=dist.ReduceOp.MAX, group=group)
    return hist, maxv


def make_comm(rank, world, device=0, timeout_ms=0, before_init=None):
    """One engine communicator per rank: rank 0 draws the RCCL id, torch.distributed (any backend) ships it.
    timeout_ms > 0: mtr_comm_init_timeout — a rank that never arrives costs the others that long, not forever.
    before_init (tests): called between the id's arrival and mtr_comm_init — where a rank can be made late for the others."""
    import torch.distributed as dist
    uid = [_engine.comm_unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(uid, src=0)
    if before_init:
        before_init()
    return _engine.Comm(rank, world, uid[0], device, timeout_ms=timeout_ms)


def _abandon_group(dist, group):
    """A NCCL group whose probe timed out holds an unfinished collective: abort it (ProcessGroupNCCL.abort / _shutdown where
    this torch has them) so that nothing at teardown waits for a peer that never answered.  Best effort, never raises."""
    try:
        backend = group._get_backend(__import__("torch").device("cuda"))
    except Exception:                                             # noqa: BLE001
        backend = None
    for obj, name in ((backend, "abort"), (backend, "_shutdown"), (group, "abort"), (group, "_shutdown")):
        fn = getattr(obj, name, None) if obj is not None else None
        if callable(fn):
            try:
                fn()
                return True
            except Exception:                                     # noqa: BLE001
                continue
    return False


def agree_on_collective(rank, world, make, device=None, allow_nccl=True, log=None, probe_timeout_s=60.0):
    """How the ranks of a job reduce — decided TOGETHER, so that no rank ever sits alone inside a collective.

    `make()` builds this rank's engine communicator (mtr_comm_init_timeout behind it: itself a collective call, bounded by
    its deadline) or raises.  Every rank is here before anyone starts (a barrier on both sides); then the ranks vote (MIN
    over "mine succeeded", on CPU tensors: the process group is the control plane, gloo).  All succeeded -> the job's FIRST
    collective is probed the same way (`comm.probe`: one 4-byte all-reduce polled to `probe_timeout_s`, then a vote) ->
    (comm, None, "RCCL behind the C ABI ...").  Otherwise every rank closes what it built and the ranks fall back, in
    order: a torch.distributed NCCL (= RCCL) group — voted on right after it is created, before any traffic on it, then
    probed with one ASYNCHRONOUS all-reduce on `device` that is waited for `probe_timeout_s` at most and voted on again
    (skipped where allow_nccl is False: two ranks on one GPU, or no GPU at all) — then the default group: gloo, device
    buffers through the host.  Returns (comm, group, description); the description names the fallback and why.
    A rank that is lost altogether surfaces as the control plane's own timeout (init_process_group(timeout=...))."""
    import time

    import torch
    import torch.distributed as dist

    def agreed(ok):
        if world == 1:
            return bool(ok)
        t = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t.item()))

    comm, err = None, None
    if world > 1:
        dist.barrier()
    try:
        comm = make()
    except Exception as ex:                                       # noqa: BLE001 — reported in the description, never silent
        err = ex
    if world == 1:
        if err:
            raise err
        return comm, None, "RCCL behind the C ABI (mtr_engine_reduce)"
    why = None
    if agreed(err is None):
        # every rank holds a communicator: the first collective on it, bounded, before anything is timed
        try:
            if hasattr(comm, "probe"):
                comm.probe_ms = comm.probe(int(1e3 * probe_timeout_s))
        except Exception as ex:                                   # noqa: BLE001
            err = ex
        if agreed(err is None):
            dist.barrier()
            return comm, None, "RCCL behind the C ABI (mtr_engine_reduce)"
        why = "the first all-reduce on the engine's communicator failed on a rank: %s" % (err or "another rank")
    else:
        why = "mtr_comm_init failed on a rank: %s" % (err or "another rank")
    if comm is not None:
        comm.close()
    group, ok = None, False
    if allow_nccl:
        try:
            group = dist.new_group(backend="nccl")               # "nccl" is RCCL on ROCm
            ok = True
        except Exception as ex:                                   # noqa: BLE001
            why += "; torch NCCL group: %r" % (ex,)
        if agreed(ok):                                            # nobody touches the group unless everybody has one
            ok = False
            try:
                probe = torch.ones(1, dtype=torch.int32, device=device)
                work = dist.all_reduce(probe, group=group, async_op=True)
                t_end = time.monotonic() + probe_timeout_s
                while not work.is_completed() and time.monotonic() < t_end:
                    time.sleep(0.01)
                if work.is_completed():
                    work.wait()
                    ok = int(probe.item()) == world
                else:
                    why += "; torch NCCL group: the probe all-reduce did not finish within %g s" % probe_timeout_s
                    # the all-reduce is still outstanding on this group: it must not be waited for at teardown (destroy_process_group
                    # would sit in it for good) — abort the group's communicators where this torch can, else shut it down
                    _abandon_group(dist, group)
            except Exception as ex:                               # noqa: BLE001
                why += "; torch NCCL group: %r" % (ex,)
            ok = agreed(ok)
        else:
            ok = False
    if ok:
        desc = "torch.distributed all_reduce over a NCCL (= RCCL) group (%s)" % why
    else:
        group = None
        desc = "torch.distributed all_reduce over gloo, device buffers through the host (%s)" % why
    if log:
        log(desc)
    dist.barrier()
    return None, group, desc


def programme_summary(hist, maxv):
    """Programme-level record from the reduced aggregates: integrated loudness and range exactly as
    Ebu_r128_hist::calc_integ / calc_range compute them from a histogram (ebu_r128_proc.cc:105-150)."""
    h = hist.detach().cpu().numpy().reshape(2, _engine.HIST_LEN)
    m = maxv.detach().cpu().numpy()
    integ, integ_thr, rmin, rmax, rthr = _engine.hist_loudness(h[0], h[1])
    return dict(integrated=integ, integ_thr=integ_thr, range_min=rmin, range_max=rmax, range_thr=rthr,
                truepeak=(float(m[0]), float(m[1])), maxloudn_M=float(m[2]), maxloudn_S=float(m[3]),
                hist_M_count=int(h[0].sum()), hist_S_count=int(h[1].sum()))
