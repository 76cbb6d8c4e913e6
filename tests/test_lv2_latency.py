"""LV2 run() latency of the GPU plugins (reference contract: the plugins are lv2:hardRTCapable,
lv2ttl/meters.lv2.ttl.in:609, and run() is called once per audio block, src/ebulv2.cc:340-367).

One run() = interleave into page-locked memory, one H2D copy, the kernels, one D2H copy of the result, ONE wait
(mtr_engine_process_planar_host); the result getters are served from that snapshot.  The block's real-time budget
is n / fs; the test asserts the median AND the 99th percentile stay below it at 1024 frames, and prints the table
DESIGN.md quotes."""
import pytest

from _lv2host import Host
from _lv2lat import run_latency

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def host():
    return Host()


@pytest.mark.parametrize("name", ["EBUr128", "dBTPstereo", "spectr30stereo"])
def test_run_latency_below_the_block_budget(host, name):
    rows = []
    for n in (64, 256, 1024, 8192):
        r = run_latency(host, name, n, blocks=200, warm=20)
        rows.append((n, r))
        print("%-15s n=%5d  median %8.1f us  p99 %8.1f us  max %9.1f us  budget %9.1f us  (%.1f %% of the budget)"
              % (name, n, r["median_us"], r["p99_us"], r["max_us"], r["budget_us"], 100 * r["median_us"] / r["budget_us"]))
    r1024 = dict(rows)[1024]
    assert r1024["median_us"] < r1024["budget_us"] and r1024["p99_us"] < r1024["budget_us"], r1024
    r8192 = dict(rows)[8192]
    assert r8192["median_us"] < r8192["budget_us"]


def test_run_latency_with_the_ui_attached(host):
    """EBUr128 with a GUI attached also fetches the two histograms when the radar advances: still one block budget."""
    r = run_latency(host, "EBUr128", 1024, blocks=200, warm=20, ui=True)
    print("EBUr128 + UI   n= 1024  median %8.1f us  p99 %8.1f us  max %9.1f us  budget %9.1f us"
          % (r["median_us"], r["p99_us"], r["max_us"], r["budget_us"]))
    assert r["median_us"] < r["budget_us"]


@pytest.mark.parametrize("name", ["EBUr128", "dBTPstereo", "spectr30stereo"])
def test_no_block_misses_its_budget_from_the_first_one(host, name):
    """instantiate() warms the engine up (mtr_engine_prepare_host): the host's FIRST run() — staging buffers, the engine's
    stream, every kernel's code object — and the first blocks in which a fragment ends (plan buffers with head-room) cost
    what any other block costs.  64-frame blocks: 1333 us each; before, the first took 6 ms and five more 0.3 ms."""
    r = run_latency(host, name, 64, blocks=400, warm=0)
    print("%-15s n=   64 from the first block: median %.1f us  first ten <= %.1f us  max %.1f us  budget %.1f us"
          % (name, r["median_us"], r["first10_max_us"], r["max_us"], r["budget_us"]))
    # the claim is about the engine: the first blocks cost what the others do.  (One block in 400 may still meet a scheduler
    # hiccup of the host it shares with others: that is not the engine's, so one miss is tolerated, never among the first ten.)
    assert r["first10_max_us"] < r["budget_us"] and r["p99_us"] < r["budget_us"] and r["over_budget"] <= 1, r


def test_a_host_that_states_its_block_length_gets_it_warmed_up(host):
    """options:options with buf-size:maxBlockLength = 32768: instantiate() prepares the engine for blocks of that size, so the
    first run() of 32768 frames does not pay for page-locked and device staging buffers (without the option the engine is
    warmed for 8192 frames and the first larger block allocates)."""
    # An allocation in the first run() is deterministic, a scheduler hiccup of the shared host is not.  The claim is judged
    # against the UNTOLD instance on the same box, attempt by attempt (no retry-until-green): told must keep its first
    # block within 5 % + 100 us of its own median in at least 2 of 3 fresh instances.
    bound = lambda r: 1.05 * r["median_us"] + 100.0                          # noqa: E731
    tolds = [run_latency(host, "dBTPstereo", 32768, blocks=12, warm=0, max_block_length=32768) for _ in range(3)]
    untold = run_latency(host, "dBTPstereo", 32768, blocks=12, warm=0)
    good = [t for t in tolds if t["first10_max_us"] < bound(t)]
    print("dBTPstereo n=32768: first block %s us told / %.0f us untold, median %.0f us" % ([round(t["first10_max_us"]) for t in tolds], untold["first10_max_us"], tolds[0]["median_us"]))
    assert len(good) >= 2, tolds      # (untold: + ~0.3 ms of hipHostMalloc / hipMalloc, printed above)
