"""tools/split_precision.py — how accurate is the three-product f16 split of the 4x interpolator (mtr_mfma16_fir.h)?
CPU only (numpy): emulates the f32 fmaf chain of the reference's Resampler::process, the split with three and with two
partial products (exact f16 products, wide accumulation, one final rounding to f32 — what the MFMA does), and compares
both with a float64 interpolator on a few signals.  Output quoted in DESIGN.md 3.1."""
import numpy as np
import meters.lv2_amd as M

tab = np.zeros(120, np.float32)
M.lib.mtr_fir_table(tab.ctypes.data)
g = np.zeros((3, 48), np.float32)
for ph in (1, 2, 3):
    for i in range(48):
        g[ph - 1, i] = tab[24 * ph + i] if i < 24 else tab[24 * (4 - ph) + (47 - i)]
rng = np.random.default_rng(1)


def fir_f32(x):
    n = len(x) - 47
    out = np.zeros((3, n), np.float32)
    for p in range(3):
        s = np.zeros(n, np.float32)
        for i in range(48):
            s = (s.astype(np.float64) + g[p, i].astype(np.float64) * x[i:i + n].astype(np.float64)).astype(np.float32)
        out[p] = s
    return out.astype(np.float64)


def fir_f64(x):
    n = len(x) - 47
    return np.stack([sum(g[p, i].astype(np.float64) * x[i:i + n].astype(np.float64) for i in range(48)) for p in range(3)])


def fir_split(x, nprod):
    tm = np.abs(x).max()
    sc = 2.0 ** (14 - np.floor(np.log2(tm))) if tm > 0 else 1.0
    xs = (x.astype(np.float64) * sc).astype(np.float32)
    xh = xs.astype(np.float16)
    xl = (xs - xh.astype(np.float32)).astype(np.float16)
    G = (g.astype(np.float64) * 2 ** 15).astype(np.float32)
    gh = G.astype(np.float16)
    gl = (G - gh.astype(np.float32)).astype(np.float16)
    n = len(x) - 47
    out = []
    for p in range(3):
        s = np.zeros(n)
        for i in range(48):
            s += gh[p, i].astype(np.float64) * (xh[i:i + n].astype(np.float64) + xl[i:i + n].astype(np.float64))
            if nprod == 3:
                s += gl[p, i].astype(np.float64) * xh[i:i + n].astype(np.float64)
        out.append(s.astype(np.float32).astype(np.float64) / sc / 2 ** 15)
    return np.stack(out)


signals = [("noise", rng.uniform(-1, 1, 4000)), ("programme", rng.uniform(-1, 1, 4000) * 0.3 + 0.5 * np.sin(np.arange(4000) * 0.05)),
           ("quiet (1e-6)", rng.uniform(-1, 1, 4000) * 1e-6), ("fs/4 pattern", np.tile([1.0, 1.0, -1.0, -1.0], 1000)),
           ("impulse + 1e-7 noise", np.concatenate([np.zeros(100), [1.0], rng.uniform(-1, 1, 3899) * 1e-7]))]
print("%-22s %12s | max |err| / peak:  f32 chain   3 products   2 products" % ("signal", "peak"))
for name, x in signals:
    x = x.astype(np.float32)
    r = fir_f64(x)
    pk = np.abs(r).max()
    e = [np.abs(v - r).max() / pk for v in (fir_f32(x), fir_split(x, 3), fir_split(x, 2))]
    print("%-22s %12.6g | %28.2e %12.2e %12.2e" % (name, pk, *e))
