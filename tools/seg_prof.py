"""tools/seg_prof.py — shader cycles per part of a step of k_seg (layout 7), one wave's view.  Needs a library with the
counters compiled in:
    make -C meters.lv2_amd/csrc OUT=../lib_prof EXTRA_mtr_seg="<the Makefile's EXTRA_mtr_seg> -DMTR_SEG_PROF" ../lib_prof/libmtr_engine.so
    MTR_LIB=meters.lv2_amd/lib_prof/libmtr_engine.so python tools/seg_prof.py [ebu+tp|tp] [sample rate]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import meters.lv2_amd as M
from meters.lv2_amd import engine as E
what = sys.argv[1] if len(sys.argv) > 1 else "ebu+tp"
meters = (M.METER_EBU | M.METER_TRUEPEAK) if what == "ebu+tp" else M.METER_TRUEPEAK
fs = float(sys.argv[2]) if len(sys.argv) > 2 else 48000.0
S, T = 8192, int(fs) * 10
buf = torch.empty((S, T, 2), dtype=torch.float32, device="cuda")
M.synth_fill_device(buf.data_ptr(), S, T, T, 777, fs, 1, torch.cuda.current_stream().cuda_stream)
with M.Engine(S, fs, meters, tune_layout=7) as e:
    if meters & M.METER_EBU:
        e.integr_start()
    for _ in range(2):
        e.process_device(buf.data_ptr(), T, T, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 8)()
    E.lib.mtr_debug_seg_prof(out)
    n = max(out[6], 1)
    names = ["step head: phase 0, scale check, operand fetch 0, 8 loads issued", "chunk 0 (waits for the stream)", "chunks 1-6",
             "chunk 7 (+ ring stores, next operand fetch)", "the recurrence's packed block (EBU), tile bookkeeping", "whole step"]
    print(what, fs, "steps", n, "seg_stats", e.seg_stats())
    for i, nm in enumerate(names):
        print("  %-70s %9.1f cycles / step" % (nm, out[i] / n))
