#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
for L in lib lib_ab; do echo "== $L"; MTR_LIB=$PWD/meters.lv2_amd/$L/libmtr_engine.so timeout 600 python tools/stride_probe.py /tmp/sp_$L.npz 2>&1 | grep -v amdgpu; done | tee $O/stride30.txt
python - <<PY | tee -a $O/stride30.txt
import numpy as np
a = np.load("/tmp/sp_lib.npz"); b = np.load("/tmp/sp_lib_ab.npz")
for k in sorted(a.files):
    x, y = a[k].astype(np.float64), b[k].astype(np.float64)
    if k.startswith("o9"): print(k, "max |d| dB", np.nanmax(np.abs(x - y)))
    else: print(k, "max rel", np.nanmax(np.abs(x - y) / np.maximum(np.abs(y), 1e-30)))
PY
