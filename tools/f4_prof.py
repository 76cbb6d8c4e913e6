"""tools/f4_prof.py — shader cycles per phase of k_kwtp16 (layout 6).
Needs a library built with the counters compiled in:
    make -C meters.lv2_amd/csrc OUT=../lib_prof EXTRA_mtr_fused4="-mllvm -amdgpu-mfma-vgpr-form -DMTR_F4_PROF" ../lib_prof/libmtr_engine.so
    MTR_LIB=meters.lv2_amd/lib_prof/libmtr_engine.so python tools/f4_prof.py [ebu+tp|tp]
The counters are one workgroup's (one wave's) view; with two waves per SIMD a phase's cycles include what the
other wave of the SIMD took from it."""
import ctypes as C, sys
import torch
import meters.lv2_amd as M
from meters.lv2_amd import engine as E
what = sys.argv[1] if len(sys.argv) > 1 else "ebu+tp"
meters = (M.METER_EBU | M.METER_TRUEPEAK) if what == "ebu+tp" else M.METER_TRUEPEAK
S, T = 8192, 48000 * 2
buf = (torch.rand(S, T, 2, device="cuda") - 0.5)
with M.Engine(S, 48000.0, meters, tune_layout=6) as e:
    if meters & M.METER_EBU:
        e.integr_start()
    for _ in range(2):
        e.process_device(buf.data_ptr(), T, T, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 12)()
    E.lib.mtr_debug_f4_prof(out)
    n = out[8]
    names = ["wait for the tile (DMA)", "x -> registers, next halo", "maxima, scales, phase 0", "split + write words",
             "K-filter (pass 1, scan, pass 2)", "MFMA phase", "halo back, DMA issue, zero tail", "total"]
    print(what, "tiles", n)
    for i, nm in enumerate(names):
        print("  %-34s %9.1f cycles / tile" % (nm, out[i] / n))
    print("  %-34s %9.1f cycles / tile" % ("  K-filter: pass 1", out[9] / n))
    print("  %-34s %9.1f cycles / tile" % ("  K-filter: scan", out[10] / n))
