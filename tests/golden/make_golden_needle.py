#!/usr/bin/env python3
"""tests/golden/make_golden_needle.py — golden vectors for the needle meters, FROM THE REFERENCE BUILD
(oracle/_ref/libmeters_ref.so: jmeters/{iec1ppm,iec2ppm,msppm,stcorr,kmeter}dsp.cc compiled where they lie).

Run in the authoring container only (needs /root/reference); writes tests/golden/golden_needle_v1.npz: the
values read() returns after every 1024-frame block of a reproducible burst-noise signal (integer LCG and
power-of-two gains only, so any machine regenerates the same input bits).  Data only — no reference text."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _signals as sig  # noqa: E402
from _oracle import Reference  # noqa: E402

F = C.c_float
B = 1024


def signal(n=48000, seed=4242):
    x = sig.lcg_noise(n, seed, 1.0)
    env = np.where(np.arange(n) % 16384 < 4096, np.float32(0.5), np.float32(0.0625))
    return (x[:, 0] * env).astype(np.float32), (x[:, 1] * env[::-1]).astype(np.float32)


def fp(a):
    return a.ctypes.data_as(C.POINTER(F))


def main():
    r = Reference().lib
    for f in ("ref_ppm_new", "ref_msppm_new", "ref_stcorr_new", "ref_kmeter_new"):
        getattr(r, f).restype = C.c_void_p
    r.ref_ppm_new.argtypes = [C.c_int, F]
    r.ref_msppm_new.argtypes = [F, F]
    r.ref_stcorr_new.argtypes = [C.c_int, F, F]
    r.ref_kmeter_new.argtypes = [F]
    for f in ("ref_ppm_read", "ref_msppm_read", "ref_stcorr_read"):
        getattr(r, f).restype = F
        getattr(r, f).argtypes = [C.c_void_p]
    r.ref_ppm_process.argtypes = [C.c_void_p, C.POINTER(F), C.c_int]
    r.ref_msppm_process.argtypes = [C.c_void_p, C.POINTER(F), C.POINTER(F), C.c_int, C.c_int]
    r.ref_stcorr_process.argtypes = [C.c_void_p, C.POINTER(F), C.POINTER(F), C.c_int]
    r.ref_kmeter_process.argtypes = [C.c_void_p, C.POINTER(F), C.c_int]
    r.ref_kmeter_read.argtypes = [C.c_void_p, C.POINTER(F), C.POINTER(F)]
    xl, xr = signal()
    out = {}
    for fs in (44100.0, 48000.0, 96000.0):
        tag = str(int(fs))
        for kind in (1, 2):
            h = r.ref_ppm_new(kind, fs)
            seq = []
            for q in range(0, xl.size - B + 1, B):
                blk = xl[q:q + B].copy()
                r.ref_ppm_process(h, fp(blk), B)
                seq.append(r.ref_ppm_read(h))
            out[f"iec{kind}_{tag}"] = np.array(seq, np.float32)
        for side in (0, 1):
            h = r.ref_msppm_new(fs, -6.0)
            seq = []
            for q in range(0, xl.size - B + 1, B):
                bl, br = xl[q:q + B].copy(), xr[q:q + B].copy()
                r.ref_msppm_process(h, fp(bl), fp(br), B, side)
                seq.append(r.ref_msppm_read(h))
            out[f"msppm{'MS'[side]}_{tag}"] = np.array(seq, np.float32)
        h = r.ref_stcorr_new(int(fs), 2e3, 0.3)
        seq = []
        for q in range(0, xl.size - B + 1, B):
            bl, br = xl[q:q + B].copy(), xr[q:q + B].copy()
            r.ref_stcorr_process(h, fp(bl), fp(br), B)
            seq.append(r.ref_stcorr_read(h))
        out[f"stcorr_{tag}"] = np.array(seq, np.float32)
        h = r.ref_kmeter_new(fs)
        seq = []
        a, p = F(), F()
        for q in range(0, xl.size - B + 1, B):
            blk = xl[q:q + B].copy()
            r.ref_kmeter_process(h, fp(blk), B)
            r.ref_kmeter_read(h, C.byref(a), C.byref(p))
            seq.append((a.value, p.value))
        out[f"kmeter_{tag}"] = np.array(seq, np.float32)
    np.savez_compressed(os.path.join(HERE, "golden_needle_v1.npz"), **out)
    print("wrote golden_needle_v1.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
