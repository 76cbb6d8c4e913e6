"""Wall time of one LV2 run() of the GPU plugins of lib/meters_amd.so, per block size (test infrastructure,
also used by bench.py's `extra.lv2_run_latency`).  The clock is around the plugin's run() only: the copy to the
device, the kernels, the results' way back and the port writes — what a host's audio thread would wait for."""
import time

import numpy as np

from _lv2host import Instance, forge_object, forge_sequence, notify_buffer, arm_notify, MTR_URI


def _f(v=0.0):
    return np.array([v], np.float32)


def _wire(host, name, n, keep, max_block_length=None):
    """Instantiate `name` with every port connected for blocks of n frames; returns (inst, per_block_hook)."""
    rng = np.random.default_rng(7)
    bl = (rng.uniform(-0.5, 0.5, n)).astype(np.float32)
    br = (rng.uniform(-0.5, 0.5, n)).astype(np.float32)
    inst = Instance(host, name, rate=48000.0, max_block_length=max_block_length)
    assert inst.ok(), name
    hook = None
    if name == "EBUr128":
        notify = notify_buffer(16384)
        empty = forge_sequence(host, [])
        first = forge_sequence(host, [forge_object(host, MTR_URI + "meteron", [])]) if keep.get("ui") else empty
        inst.connect(0, first); inst.connect(1, notify)
        for port, arr in ((2, bl), (3, bl), (4, br), (5, br)):
            inst.connect(port, arr)
        keep.update(notify=notify, empty=empty, first=first)
        state = {"i": 0}

        def hook():
            arm_notify(notify)
            if state["i"] == 1:
                inst.connect(0, empty)
            state["i"] += 1
    elif name == "dBTPstereo":
        ports = [_f(1.0), _f(), _f(), _f(), _f()]
        for port, arr in zip((0, 3, 6, 7, 8), ports):
            inst.connect(port, arr)
        for port, arr in ((1, bl), (2, bl), (4, br), (5, br)):
            inst.connect(port, arr)
        keep.update(ports=ports)
    elif name == "spectr30stereo":
        spec = [_f() for _ in range(60)]
        ctl = [_f(1.0), _f(-4.0), _f(0.0), _f(0.0)]
        for i in range(60):
            inst.connect(i, spec[i])
        for port, arr in zip((60, 61, 62, 63), ctl):
            inst.connect(port, arr)
        for port, arr in ((64, bl), (65, bl), (66, br), (67, br)):
            inst.connect(port, arr)
        keep.update(spec=spec, ctl=ctl)
    else:
        raise ValueError(name)
    keep.update(bl=bl, br=br)
    return inst, hook


def run_latency(host, name, n, blocks=300, warm=30, ui=False, max_block_length=None):
    """-> dict(median_us, p99_us, max_us, budget_us) for blocks of n frames at 48 kHz."""
    keep = {"ui": ui}
    inst, hook = _wire(host, name, n, keep, max_block_length)
    t = np.zeros(blocks)
    for i in range(warm + blocks):
        if hook:
            hook()
        t0 = time.perf_counter()
        inst.run(n)
        dt = time.perf_counter() - t0
        if i >= warm:
            t[i - warm] = dt
    inst.cleanup()
    return {"median_us": float(np.median(t) * 1e6), "p99_us": float(np.quantile(t, 0.99) * 1e6),
            "max_us": float(t.max() * 1e6), "budget_us": n / 48000.0 * 1e6,
            "first10_max_us": float(t[:10].max() * 1e6), "over_budget": int((t > n / 48000.0).sum())}
