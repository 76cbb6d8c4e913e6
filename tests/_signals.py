"""Seeded synthetic 48 kHz stereo test signals, numpy only.

All return float32 arrays of shape [T, 2] (interleaved stereo frames).  The
same buffer is handed to the oracle and to the HIP engine, so libm differences
in sin() never enter a comparison.
"""
import numpy as np

_A = np.uint64(1664525)
_C = np.uint64(1013904223)
_M = np.uint64(0xFFFFFFFF)


def _lcg_jump(n):
    """(A_k, C_k), k = 1..n, with s_k = A_k * s_0 + C_k (mod 2^32)."""
    A = np.empty(n, np.uint64)
    Cc = np.empty(n, np.uint64)
    a, c = np.uint64(1), np.uint64(0)
    for k in range(n):
        a = (a * _A) & _M
        c = (c * _A + _C) & _M
        A[k], Cc[k] = a, c
    return A, Cc


_JUMP = None
_JB = 4096


def lcg_u32(n, seed):
    """n successive states of s <- 1664525 s + 1013904223 (mod 2^32), starting after `seed`."""
    global _JUMP
    if _JUMP is None:
        _JUMP = _lcg_jump(_JB)
    A, Cc = _JUMP
    out = np.empty(n, np.uint64)
    s = np.uint64(seed & 0xFFFFFFFF)
    for p in range(0, n, _JB):
        m = min(_JB, n - p)
        blk = (A[:m] * s + Cc[:m]) & _M
        out[p:p + m] = blk
        s = blk[m - 1]
    return out.astype(np.uint32)


def lcg_noise(T, seed, gain=1.0):
    """Uniform [-1, 1) noise, two draws per frame (L then R); bit-identical to mo_fill_lcg."""
    s = lcg_u32(2 * T, seed)
    u = ((s >> np.uint32(8)).astype(np.int64) - (1 << 23)).astype(np.float32) / np.float32(1 << 23)
    return (np.float32(gain) * u).reshape(T, 2)


def sine(T, freq, amp=1.0, fs=48000.0, phase=0.0, freq_r=None, amp_r=None):
    t = np.arange(T, dtype=np.float64) / fs
    L = amp * np.sin(2 * np.pi * freq * t + phase)
    R = (amp if amp_r is None else amp_r) * np.sin(2 * np.pi * (freq if freq_r is None else freq_r) * t + phase)
    return np.stack([L, R], 1).astype(np.float32)


def g0(T, fs=48000.0):
    """1 kHz 0 dBFS both channels -> 0.0 LUFS."""
    return sine(T, 1000.0, 1.0, fs)


def g1(T, fs=48000.0):
    """997 Hz at -23 dBFS."""
    return sine(T, 997.0, 10 ** (-23 / 20), fs)


def g2(T, seed=777, fs=48000.0):
    """Programme-like: slow envelope over noise + tone, exercises gating and LRA."""
    t = np.arange(T, dtype=np.float64) / fs
    env = 0.05 + 0.45 * (0.5 + 0.5 * np.sin(2 * np.pi * 0.2 * t))
    u = lcg_noise(T, seed).astype(np.float64)
    L = env * (0.5 * u[:, 0] + 0.5 * np.sin(2 * np.pi * 440.0 * t))
    R = env * (0.5 * u[:, 1] + 0.5 * np.sin(2 * np.pi * 3000.0 * t))
    return np.stack([L, R], 1).astype(np.float32)


def g3(T):
    """fs/4 sine at 45 degrees as the exact pattern {+1,+1,-1,-1}: inter-sample peak +3 dBTP."""
    pat = np.array([1, 1, -1, -1], np.float32)
    x = np.tile(pat, (T + 3) // 4)[:T]
    return np.stack([x, 0.5 * x], 1).astype(np.float32)


def g4(T, band, fs=48000.0):
    """Full-scale sine at the centre of 1/3-octave band `band` (0..29)."""
    return sine(T, 1000.0 * 2.0 ** ((band - 16) / 3.0), 1.0, fs)


def g5(n, seed=4242):
    """Bit-pattern soup for the bit meter: raw LCG words reinterpreted as f32 (NaN/Inf/denormals/±0)."""
    s = lcg_u32(n, seed).copy()
    # sprinkle exact specials so every branch of float_stats is hit
    s[::97] = 0x00000000
    s[1::97] = 0x80000000
    s[2::97] = 0x7F800000
    s[3::97] = 0xFF800000
    s[4::97] = 0x7FC00001
    s[5::97] = 0x00000001
    s[6::97] = 0x807FFFFF
    return s.view(np.float32)


def dc_plus_quiet(T, seed=99, dc=0.25, level=2.0 ** -10):
    """Large DC offset under a quiet noise programme — the K-filter's integrator states grow to
    ~1e4 x the output; catches time-parallel schemes that lose the cancellation."""
    return (lcg_noise(T, seed, level) + np.float32(dc)).astype(np.float32)
