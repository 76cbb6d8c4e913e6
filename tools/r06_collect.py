#!/usr/bin/env python3
"""tools/r06_collect.py — copy what tools/r06_profiles.sh left under gpurun_out/r06/ into the tracked profiles/r06* files
(kernel-trace medians + PMC averages as small markdown; the line, the rehearsals, the suite and fuzz summary as they are)."""
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(ROOT, "gpurun_out", "r06"), os.path.join(ROOT, "profiles")
F = re.compile(r"RCCL|HIP version|ROCm version|Hostname|Librccl|amdgpu\.ids")


def md_from_summary(src, dst, title, command, kernel, bytes_per_launch, note=""):
    txt = open(os.path.join(O, src)).read()
    med = None
    for line in txt.splitlines():
        if kernel in line and not line.startswith("("):
            m = re.search(r"\s(\d+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+\(avg ([0-9.]+)\)", line)
            n, med, mn, mx, avg = int(m.group(1)), float(m.group(2)), float(m.group(3)), float(m.group(4)), float(m.group(5))
            break
    head = ["# " + title, "", "Command: " + command, ""]
    if med:
        tb = bytes_per_launch / (med * 1e-3) / 1e12
        ta = bytes_per_launch / (avg * 1e-3) / 1e12
        head += ["**`%s`: %d dispatches, AVERAGE %.4f ms = %.3f TB/s = %.1f %% of the 8 TB/s HBM roofline; median %.4f ms = %.1f %% (min %.4f, max %.4f: the first dispatch is cold)** "
                 "(%.2f GB of algorithmic bytes per launch)." % (kernel, n, avg, ta, 100 * ta / 8.0, med, 100 * tb / 8.0, mn, mx, bytes_per_launch / 1e9), ""]
    if note:
        head += [note, ""]
    head += ["Units: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES in cycles, GRBM_GUI_ACTIVE summed over the 8 XCDs; "
             "FETCH_SIZE / WRITE_SIZE in KiB — on gfx950 FETCH_SIZE reports half the bytes of a wide stream (MI355X_MICROARCH.md, HBM): x 2.", "", "```"]
    open(os.path.join(P, dst), "w").write("\n".join(head) + "\n" + txt.rstrip() + "\n```\n")
    print("wrote", dst, med)


md_from_summary("r06a_seg_ebu_tp.txt", "r06a_kseg_ebu_tp.md", "rocprofv3 summary of the headline kernel, round 6 (r06a_seg_ebu_tp)",
                "`python bench.py --no-cpu-baseline --no-extra --steps 20 --warmup 5` under `rocprofv3 --kernel-trace --stats` and, in separate runs (`--steps 6 --warmup 1`), "
                "`rocprofv3 --pmc ...` (`tools/prof_seg.sh` via `tools/r06_profiles.sh`; the kernel trace on the driver's flags `--steps 20 --warmup 5`: 25 dispatches of `k_seg`; the PMC passes on 7).",
                "k_seg<true, true>", 8192 * 480000 * 8,
                "`k_seg` is round 4's kernel instruction for instruction; what is new beside it is the DEFERRED TAIL: in the kernel trace `k_gate` (512 workgroups, behind "
                "`k_delay`'s 100 us) runs on the engine's side stream beside the next `k_seg` — its 1.9 ms are elapsed time beside a kernel that owns the SIMDs, not work "
                "(0.13 ms alone) — and the PMC passes serialise the kernels, so their `k_seg` is the undisturbed one.  See r06_tail.md.")
md_from_summary("r06a44_seg_ebu_tp.txt", "r06a44_kseg_ebu_tp_44k1.md", "the same at 44.1 kHz (r06a44_seg_ebu_tp): fragments of 2205 frames end inside k_seg's 16-frame steps",
                "`python bench.py --no-cpu-baseline --no-extra --steps 6 --warmup 1 --fs 44100` (`tools/prof_seg.sh`).", "k_seg<true, false>", 8192 * 441000 * 8)
md_from_summary("r06_tpb.txt", "r06_tpb_pmc.md", "k_tpb<2> (TruePeakdsp::process for a batch), 8192 streams x 10 s: kernel trace + PMC, round 6 (the round-5 kernel: r06_tpb.md)",
                "`python bench.py --no-cpu-baseline --no-extra --steps 4 --warmup 1 --meters tpb` (`tools/prof_seg.sh`).", "k_tpb<2>", 8192 * 480000 * 8)
for src, dst in (("r06_traffic.json", "r06_traffic.json"), ("r06_bench_line.json", "r06_bench_line.json"), ("r06_two_ranks_one_gpu.json", "r06_two_ranks_one_gpu.json"),
                 ("r06_sleeping_rank.json", "r06_sleeping_rank.json")):
    if os.path.exists(os.path.join(O, src)):
        shutil.copy(os.path.join(O, src), os.path.join(P, dst))
out = ["# round 6, final sources, one GPU-box call of tools/r06_profiles.sh: the -m gpu suite (-s: the full-size tests print the histogram flip rates), smoke (), the four fuzzers on fresh seeds",
       "## python -m pytest tests -m gpu -q -s"]
for line in open(os.path.join(O, "gputests.txt")):
    if re.search(r"passed|failed|histogram flips|1 stream x 3600|pytest rc|dBTPstereo n=32768", line) and not F.search(line):
        out.append(line.rstrip())
out += ["## __graft_entry__.smoke ()"] + [l.rstrip() for l in open(os.path.join(O, "smoke.txt")) if not F.search(l)]
out += ["## tools/fuzz_more.py 25000 400; tools/fuzz_unaligned.py 5000 400; tools/fuzz_tpb.py 31000 800 (relative deviation of the ballistics from the oracle, bound 4e-6); tools/fuzz_intstat.py 9000 400"]
out += [l.rstrip() for l in open(os.path.join(O, "fuzz.txt")) if not F.search(l)]
forced = os.path.join(P, "r06_suite_and_fuzz.txt")
keep = ""
if os.path.exists(forced) and "## the same suite, the fuzzers and smoke () with MTR_TAIL_MODE=2" in open(forced).read():
    t = open(forced).read()
    keep = t[t.index("## the same suite, the fuzzers and smoke () with MTR_TAIL_MODE=2"):]       # (its own GPU-box call: kept across re-collections)
open(forced, "w").write("\n".join(out) + "\n" + keep)
d = json.load(open(os.path.join(O, "r06_bench_line.json")))
r = d["roofline"]
print("line: value %.4g, k_seg %.3f ms = %.4f, whole step %.3f ms = %.4f, traffic %s" % (d["value"], r["kernel_ms"], r["frac"], d["ms_per_step"], r["whole_step_frac"], r["traffic"]))
