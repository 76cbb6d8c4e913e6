#!/usr/bin/env python3
"""Generate tests/golden/golden_v1.npz from the REFERENCE's own objects.

Runs only in the build container (needs /root/reference -> oracle/_ref/libmeters_ref.so,
built by `make -C oracle ref`).  The outputs are data only: seeds/parameters of the
inputs plus the reference's results.  Inputs are regenerated from the seeds by
tests/_signals.py; the ones tagged "exact" use only exactly reproducible operations
(LCG -> float, power-of-two gains, integer-counter envelopes), so the oracle must
reproduce the stored outputs bit-for-bit on any x86-64 box; "tol" cases involve sin()
and are compared at 1e-4 dB.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import _signals as sig  # noqa: E402
from _oracle import Reference, build_ref  # noqa: E402


def tri_noise(T, seed, gain, period=240000):
    """LCG noise under an integer-counter triangle envelope (exactly reproducible)."""
    x = sig.lcg_noise(T, seed, gain)
    n = np.arange(T)
    env = np.abs((n % period) - period // 2).astype(np.float32) / np.float32(period // 2)
    return (x * env[:, None]).astype(np.float32)


def main():
    build_ref()
    ref = Reference()
    g = {}

    # ---- coefficients -------------------------------------------------------
    rates = np.array([44100.0, 48000.0, 96000.0])
    g["rates"] = rates
    g["kw_coef"] = np.stack([ref.kw_coef(float(r)) for r in rates])
    g["tp_table"] = ref.tp_table()
    g["tp_consts"] = np.stack([ref.tp_consts(float(r)) for r in rates])
    g["band_coef"] = np.stack([np.stack([ref.band_coef(float(r), i) for i in range(30)]) for r in rates])

    # ---- EBU R128, exact cases ----------------------------------------------
    # case id, T, seed, gain, fs, block
    ebu_cases = np.array([
        (48000 * 12, 777, 0.25, 48000, 1024),
        (48000 * 12, 777, 0.25, 48000, 2400),
        (48000 * 5 + 1234, 4711, 0.5, 48000, 8192),
        (44100 * 11, 5, 0.5, 44100, 1024),
        (96000 * 6, 6, 0.125, 96000, 4096),
    ], dtype=np.float64)
    g["ebu_cases"] = ebu_cases
    for i, (T, seed, gain, fs, block) in enumerate(ebu_cases):
        x = tri_noise(int(T), int(seed), float(gain))
        r = ref.ebu(x, float(fs), int(block), want_frag=True)
        g[f"ebu{i}_out9"] = r["out9"]
        g[f"ebu{i}_hist_M"] = r["hist_M"]
        g[f"ebu{i}_hist_S"] = r["hist_S"]
        g[f"ebu{i}_counts"] = r["counts"]
        g[f"ebu{i}_frag_power"] = r["frag_power"]

    # ---- EBU R128, tolerance cases (sin-based) --------------------------------
    g["ebu_g0_out9"] = ref.ebu(sig.g0(48000 * 4), 48000.0, 1024)["out9"]          # 0.0 LUFS
    g["ebu_g1_out9"] = ref.ebu(sig.g1(48000 * 20), 48000.0, 1024)["out9"]         # -23 LUFS
    r = ref.ebu(sig.g2(48000 * 30, 777), 48000.0, 1024)
    g["ebu_g2_out9"], g["ebu_g2_counts"] = r["out9"], r["counts"]
    r = ref.ebu(sig.dc_plus_quiet(48000 * 6), 48000.0, 1024, want_frag=True)      # exact (LCG + 0.25)
    g["ebu_dc_out9"], g["ebu_dc_frag_power"] = r["out9"], r["frag_power"]

    # ---- true peak ------------------------------------------------------------
    g["tp_lcg_peak"] = ref.tp(sig.lcg_noise(48000 * 3, 1234), 48000.0, 1024)       # exact
    g["tp_g3_peak"] = ref.tp(sig.g3(48000 * 2), 48000.0, 1024)                     # exact: 1.429816
    g["tp_resample_lcg"] = ref.tp_resample(sig.lcg_noise(2000, 31)[:, 0].copy())   # exact, 8000 outputs
    x = (sig.lcg_noise(48000 * 2, 9)[:, 0] * np.float32(0.5)).copy()
    g["tp_ballistics_lcg"] = ref.tp_process_seq(x, 48000.0, 1024)                  # exact, [(m,p)] per block

    # ---- filter bank ------------------------------------------------------------
    r = ref.spectr(sig.lcg_noise(48000, 42, 0.5), 48000.0, 1024)                   # exact
    for k in ("val", "max", "val_db", "max_db"):
        g[f"spectr_lcg_{k}"] = r[k]
    r = ref.spectr(sig.g4(48000 * 2, 16), 48000.0, 1024)                           # tol: band 16 -> 0 dB
    g["spectr_g4_16_val_db"] = r["val_db"]
    r = ref.spectr(sig.lcg_noise(44100, 43, 0.5), 44100.0, 1024)                   # exact
    g["spectr_lcg441_val"] = r["val"]

    # ---- VU (config 0) ------------------------------------------------------------
    x = sig.lcg_noise(48000, 77)[:, 0].copy()
    g["vu_lcg_block1024"] = ref.vu(x, 48000.0, 1024)                               # exact
    g["vu_g0_onecall"] = ref.vu(sig.g0(48000)[:, 0].copy(), 48000.0, None)         # tol: 1.010453

    out = os.path.join(HERE, "golden_v1.npz")
    np.savez_compressed(out, **g)
    print("wrote", out, os.path.getsize(out), "bytes,", len(g), "arrays")


def tpb_cases():
    """(name, fs, block, mono float32 input): the ballistics cases of golden_v2 — exactly reproducible inputs (LCG noise, power-of-two
    and decimal-literal gains applied in float32)."""
    out = []
    for fs in (44100.0, 96000.0):
        x = (sig.lcg_noise(int(fs) * 2, 9 + int(fs) % 97)[:, 0] * np.float32(0.5)).copy()
        out.append(("tpb_%d" % int(fs), fs, 1024, x))
    # levels the +-0.01 dB contract must hold at: -40 / -60 / -80 dBFS
    for db, g in ((40, 1e-2), (60, 1e-3), (80, 1e-4)):
        x = (sig.lcg_noise(48000, 21 + db)[:, 0] * np.float32(g)).astype(np.float32)
        out.append(("tpb_m%ddbfs" % db, 48000.0, 8192, x))
    return out


NAN_AT, NAN_T = 2000, 2600


def nan_cases():
    """Three mono signals with ONE NaN at frame NAN_AT in quiet noise (2^-12), 2600 frames.  What the reference's `if (v > m)` loses is
    decided by the 48-tap windows that contain the NaN — output frames NAN_AT .. NAN_AT + 47, all four phases (0 x NaN = NaN) — and
    by nothing else:
      A  full-scale finite samples 30, 50 and 63 frames on either side of the NaN: all of them stay visible;
      B  an fs/4 burst (+3.1 dB between the samples) whose interpolated peaks fall into the 16 output frames BEHIND the NaN's windows;
      C  a 0.9 sample ten frames in front of the NaN: its own output frame (+ 24) lies inside the NaN's windows, the reference loses it."""
    base = (sig.lcg_noise(NAN_T, 5)[:, 0] * np.float32(2.0 ** -12)).copy()
    a, b, c = base.copy(), base.copy(), base.copy()
    for x in (a, b, c):
        x[NAN_AT] = np.nan
    for d, v in ((30, 1.0), (50, -0.875), (63, 0.75)):
        a[NAN_AT - d] = np.float32(v)
        a[NAN_AT + d] = np.float32(-v * 0.9375)
    b[NAN_AT + 26:NAN_AT + 42] = np.tile(np.array([0.5, 0.5, -0.5, -0.5], np.float32), 4)
    c[NAN_AT - 10] = np.float32(0.9)
    return {"A": a, "B": b, "C": c}


def main_v2():
    """tests/golden/golden_v2.npz: TruePeakdsp::process at 44.1 and 96 kHz and at low levels, process_max around a NaN (round 5)."""
    build_ref()
    ref = Reference()
    g = {}
    for name, fs, block, x in tpb_cases():
        g[name] = ref.tp_process_seq(x, fs, block)
    for k, x in nan_cases().items():
        g["tp_nan_%s_out" % k] = ref.tp_resample(x)                     # the 4x stream itself, NaNs where the reference has them
        g["tp_nan_%s_peak" % k] = ref.tp(np.stack([x, x], 1), 48000.0, 1024)
    out = os.path.join(HERE, "golden_v2.npz")
    np.savez_compressed(out, **g)
    print("wrote", out, os.path.getsize(out), "bytes,", len(g), "arrays")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "v2":
        main_v2()
    else:
        main()
