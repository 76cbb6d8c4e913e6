"""tools/f3_prof.py — cycles per phase of k_kwtp (layout 5), from a library built with EXTRA_mtr_fused3+=-DMTR_F3_PROF."""
import ctypes as C, sys
import numpy as np, torch
import meters.lv2_amd as M
from meters.lv2_amd import engine as E
run = int(sys.argv[1]) if len(sys.argv) > 1 else 39
S, T = 8192, 48000 * 2
buf = (torch.rand(S, T, 2, device="cuda") - 0.5)
with M.Engine(S, 48000.0, M.METER_EBU | M.METER_TRUEPEAK, tune_layout=5, tune_run=run) as e:
    e.integr_start()
    for _ in range(2):
        e.process_device(buf.data_ptr(), T, T, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 8)()
    E.lib.mtr_debug_f3_prof(out)
    n = out[7]
    names = ["wait DMA + scan", "x -> regs (+DMA issue)", "split + write words", "K-filter", "MFMA phase", "halo + pass 1", "total"]
    print("run", run, "tiles", n)
    for i, nm in enumerate(names):
        print("  %-24s %9.1f cycles / tile" % (nm, out[i] / n))
