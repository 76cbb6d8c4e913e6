"""How the engine tiles and routes a call — the host logic behind mtr_engine_process_* (seg_plan / plan_tiling in
csrc/mtr_engine.hip), through mtr_plan_query: pure arithmetic, runs without a GPU.

The reference has no counterpart (it walks its samples one by one, ebumeter/ebu_r128_proc.cc:217-244); what is held here
is the contract the kernels rely on: tiles cover the call exactly, the lane = time segment kernel only ever gets whole
fragments — all of them, at every rate (no read-ahead behind the call at 44.1 / 88.2 kHz) —, segments are long enough for their warm-up."""
import math
import random

import numpy as np
import pytest

import meters.lv2_amd as M


def q(S, N, fs=48000.0, left=0, **kw):
    return M.plan_query(S, N, fs, frames_left_in_fragment=left, **kw)


def test_headline_shape():
    p = q(8192, 480000)
    assert p["layout"] == 7 and p["uses_seg"] == 1
    assert (p["head_frames"], p["body_fragments"], p["segments"], p["fragments_per_lane"]) == (0, 200, 8, 25)
    assert p["warm_steps"] == 228                                   # 0.075 s of 16-frame steps (225), rounded up to a multiple of 4
    assert p["n_tiles"] == 200 and p["n_fragments_ended"] == 200 and p["frames_left_after"] == 2400
    assert 8192 * p["segments"] == 65536                            # one lane per (stream, segment): 1024 waves, one per SIMD


@pytest.mark.parametrize("fs", [44100.0, 88200.0])
def test_fragments_that_are_not_whole_steps_all_go_to_the_batch_kernel(fs):
    """2205 / 4410 frames are not whole 16-frame steps: the lanes of a stream's last segment stop at the frame their last
    fragment ends on (mtr_seg.hip), so the launch needs no read-ahead behind the call and no fragment is left for a tail launch."""
    fragm = int(fs) // 20
    assert fragm % 16
    p = q(8192, 200 * fragm, fs)                                    # exactly 200 fragments
    assert p["uses_seg"] == 1 and p["body_fragments"] == 200 and p["n_fragments_ended"] == 200 and p["n_tiles"] == 200
    for extra in (1, 15, 16, fragm - 1):
        p = q(8192, 200 * fragm + extra, fs)                        # what is left of the call is the tail kernel's, however little
        assert p["body_fragments"] == 200 and p["n_tiles"] > 200 and p["frames_left_after"] == fragm - extra
    assert p["warm_steps"] % 4 == 0 and p["warm_steps"] * 16 >= 0.075 * fs


def test_a_call_that_starts_inside_a_fragment():
    p = q(8192, 480000, left=1000)
    assert p["uses_seg"] == 1 and p["head_frames"] == 1000 and p["head_tiles"] == 1
    assert p["body_fragments"] == (480000 - 1000) // 2400 == 199
    assert p["n_fragments_ended"] == 200 and p["frames_left_after"] == 2400 - (480000 - 1000) % 2400
    p = q(8192, 900, left=1000)                                     # ends inside the fragment it started in
    assert p["uses_seg"] == 0 and p["n_tiles"] == 1 and p["n_fragments_ended"] == 0 and p["frames_left_after"] == 100
    p = q(8192, 1000, left=1000)                                    # ... or exactly with it
    assert p["uses_seg"] == 0 and p["n_fragments_ended"] == 1 and p["frames_left_after"] == 2400


def test_small_batches_short_calls_and_pruning_stay_with_the_wave_per_segment_kernel():
    assert q(3, 48000 * 4)["uses_seg"] == 0                          # three streams do not fill 65536 lanes
    assert q(3, 48000 * 4, tune_segments=2)["uses_seg"] == 1         # ... unless forced (how the tests reach it)
    assert q(8192, 1024)["uses_seg"] == 0                            # an LV2-sized block holds no whole fragment
    p = q(8192, 480000, tune_prune=1)
    assert p["layout"] == 6 and p["uses_seg"] == 0
    p = q(1, 3600 * 48000)                                           # BASELINE config 1 with a true peak: one stream, many segments
    assert p["uses_seg"] == 0 and p["kw_segments"] > 1000
    p = q(8192, 480000, meters=M.METER_EBU)
    assert p["layout"] == 4 and p["uses_seg"] == 0
    p = q(8192, 480000, meters=M.METER_SPECTR30)                     # no fused kernel at all
    assert p["n_tiles"] == 0


def test_plan_invariants_on_random_calls():
    rng = random.Random(7)
    for _ in range(400):
        fs = rng.choice([32000.0, 44100.0, 48000.0, 88200.0, 96000.0, 192000.0])
        fragm = int(fs) // 20
        S = rng.choice([1, 5, 64, 1000, 8192, 65536])
        N = rng.choice([rng.randrange(1, 5000), rng.randrange(5000, 400000), rng.randrange(400000, 3000000)])
        left = rng.choice([0, rng.randrange(1, fragm + 1)])
        segs = rng.choice([0, 0, 1, 3, 16])
        p = q(S, N, fs, left=left, tune_segments=segs)
        open_ = left or fragm
        # fragment bookkeeping is the reference's, whatever the routing
        ended = (N - open_) // fragm + 1 if N >= open_ else 0
        assert p["n_fragments_ended"] == ended, (fs, S, N, left, p)
        assert p["frames_left_after"] == (open_ - N) % fragm or fragm, (fs, S, N, left, p)
        if p["uses_seg"]:
            head, body = p["head_frames"], p["body_fragments"]
            assert head == (0 if open_ == fragm else open_) and body >= 1
            assert head + body * fragm <= N
            assert body == (N - head) // fragm                          # every whole fragment, at every rate (no read-ahead behind the call)
            assert p["fragments_per_lane"] == math.ceil(body / p["segments"])
            if p["segments"] > 1:                                     # every later segment has its warm-up in front of it, inside the body
                assert (body // p["segments"] - 1) * fragm >= p["warm_steps"] * 16
            assert p["kw_segments"] == 0
        else:
            assert p["head_frames"] == 0 and p["body_fragments"] == 0 and p["kw_segments"] >= 1
        assert p["n_tiles"] >= max(1, ended)


def test_plan_query_rejects_nonsense():
    with pytest.raises(M.EngineError):
        q(0, 1000)
    with pytest.raises(M.EngineError):
        q(8, 0)
    with pytest.raises(M.EngineError):
        q(8, 1000, left=5000)                                        # more frames left than a fragment has
    with pytest.raises(M.EngineError):
        q(8, 1000, tune_layout=5)
