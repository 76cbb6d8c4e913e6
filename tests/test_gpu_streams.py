"""Stream ordering of the engine's host side: many calls of different sizes queued back to back
on a NON-BLOCKING stream, without a wait in between — every call changes (n_frames, fragment phase), so every call
uploads a new tiling plan while kernels of the previous ones may still be running; then the same audio on a second
stream.  Results must equal the one-call run (loudness) and the oracle (true peak)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_golden import tri_noise  # noqa: E402


def test_back_to_back_calls_on_a_nonblocking_stream(oracle):
    import torch
    import meters.lv2_amd as M
    fs, S = 48000.0, 64
    sizes = [1024, 333, 2400, 1, 4097, 1024, 1024, 977, 5000, 64] * 12
    T = sum(sizes)
    x = np.stack([tri_noise(T, 60 + s, 0.5, period=20000 + 1000 * s) for s in range(S)])
    dev = torch.from_numpy(x).cuda()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()                                # hipStreamNonBlocking
    other = torch.cuda.Stream()
    with M.Engine(S, fs, M.METER_EBU | M.METER_TRUEPEAK) as e, M.Engine(S, fs, M.METER_EBU | M.METER_TRUEPEAK) as ref:
        e.integr_start(); ref.integr_start()
        pos = 0
        for i, n in enumerate(sizes):
            st = side if i < len(sizes) // 2 else other       # half way, the caller moves to another stream
            chunk = dev[:, pos:pos + n]
            # each stream's slice of [S][T][2] is strided: stream s at base + s * T frames
            e.process_device(chunk.data_ptr(), n, T, st.cuda_stream)
            pos += n
        e.sync()
        ref.process_device(dev.data_ptr(), T, T, torch.cuda.current_stream().cuda_stream)
        a, b = e.out9(), ref.out9()
        assert np.allclose(a[:, :4], b[:, :4], atol=1e-3)
        assert np.all(np.abs(a[:, 4] - b[:, 4]) <= 0.01)
        ha, hb = e.histograms(), ref.histograms()
        for u, v in zip(ha, hb):
            assert np.array_equal(u.sum(-1), v.sum(-1))
            assert np.abs(u - v).sum() // 2 <= 2 * S
        tp = e.truepeak()
        for s in (0, 17, 63):
            want = oracle.tp(x[s], fs, 8192)
            assert np.all(np.abs(tp[s] - want) <= 2e-6 * want), (s, tp[s], want)
