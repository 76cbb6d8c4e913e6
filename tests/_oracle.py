"""ctypes access to the CPU oracle (oracle/_build/libmtr_oracle.so) and, where it was
built (build container only), the reference objects (oracle/_ref/libmeters_ref.so).

Test infrastructure: imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke() only — never by meters.lv2_amd.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "_build", "libmtr_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libmeters_ref.so")
REFERENCE_ROOT = "/root/reference"

HIST_LEN = 751
NBANDS = 30
BIM_LAST = 584
DIST_BIN = 361

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build_oracle():
    """(Re)build the oracle .so with gcc if missing or stale. Cheap (≈1 s).  MTR_ORACLE_SO: an instrumented build
    of the same sources (make check-asan)."""
    if os.environ.get("MTR_ORACLE_SO"):
        return os.environ["MTR_ORACLE_SO"]
    src = [os.path.join(ORACLE_DIR, f) for f in ("mtr_oracle.c", "mtr_oracle.h")]
    if (not os.path.exists(ORACLE_SO)
            or os.path.getmtime(ORACLE_SO) < max(os.path.getmtime(s) for s in src)):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "oracle"], stdout=subprocess.DEVNULL)
    return ORACLE_SO


def build_ref():
    """Build oracle/_ref from /root/reference when that tree exists (build container only)."""
    if not os.path.isdir(REFERENCE_ROOT):
        return None
    subprocess.check_call(["make", "-C", ORACLE_DIR, "ref"], stdout=subprocess.DEVNULL)
    return REF_SO


class _Batch:
    """The batch entry points shared by oracle (mo_*) and reference shim (ref_*)."""

    def __init__(self, lib, prefix):
        self.lib = lib
        self._ebu = getattr(lib, prefix + "batch_ebu")
        self._ebu.argtypes = [_f32p, C.c_uint32, C.c_float, C.c_uint32, _f32p, _i32p, _i32p, _i32p,
                              C.c_void_p]
        self._ebu.restype = None
        self._tp = getattr(lib, prefix + "batch_tp")
        self._tp.argtypes = [_f32p, C.c_uint32, C.c_float, C.c_uint32, _f32p]
        self._tp.restype = None
        self._sp = getattr(lib, prefix + "batch_spectr")
        self._sp.argtypes = [_f32p, C.c_uint32, C.c_double, C.c_uint32, _f32p, _f32p, _f32p, _f32p]
        self._sp.restype = None

    def ebu(self, x, fs=48000.0, block=1024, want_frag=False):
        """x: float32 [T,2]. Returns dict(out9, hist_M, hist_S, counts, frag_power)."""
        x = np.ascontiguousarray(x, np.float32)
        T = x.shape[0]
        out9 = np.zeros(9, np.float32)
        hm = np.zeros(HIST_LEN, np.int32)
        hs = np.zeros(HIST_LEN, np.int32)
        cnt = np.zeros(2, np.int32)
        nfrag = T // (int(fs) // 20)
        fp = np.zeros(max(nfrag, 1), np.float32)
        self._ebu(x, T, fs, block, out9, hm, hs, cnt,
                  fp.ctypes.data_as(C.c_void_p) if want_frag else None)
        return dict(out9=out9, hist_M=hm, hist_S=hs, counts=cnt,
                    frag_power=fp[:nfrag] if want_frag else None)

    def tp(self, x, fs=48000.0, block=1024):
        x = np.ascontiguousarray(x, np.float32)
        pk = np.zeros(2, np.float32)
        self._tp(x, x.shape[0], fs, block, pk)
        return pk

    def spectr(self, x, fs=48000.0, block=1024):
        x = np.ascontiguousarray(x, np.float32)
        val = np.zeros(NBANDS, np.float32)
        mx = np.zeros(NBANDS, np.float32)
        vdb = np.zeros(NBANDS, np.float32)
        mdb = np.zeros(NBANDS, np.float32)
        self._sp(x, x.shape[0], float(fs), block, val, mx, vdb, mdb)
        return dict(val=val, max=mx, val_db=vdb, max_db=mdb)


class MoKw(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("a0", "a1", "a2", "b1", "b2", "c3", "c4")]


class MoTp(C.Structure):
    _fields_ = [("m", C.c_float), ("p", C.c_float), ("z1", C.c_float), ("z2", C.c_float),
                ("res", C.c_int), ("w1", C.c_float), ("w2", C.c_float), ("w3", C.c_float),
                ("g", C.c_float), ("win", C.c_float * 48)]


class MoVu(C.Structure):
    _fields_ = [("z1", C.c_float), ("z2", C.c_float), ("m", C.c_float), ("res", C.c_int),
                ("w", C.c_float), ("g", C.c_float)]


class MoBiquad(C.Structure):
    _fields_ = [("W", C.c_double * 6), ("z", C.c_double * 2)]


class MoBand(C.Structure):
    _fields_ = [("f", MoBiquad * 6), ("stages", C.c_uint32), ("ac", C.c_int)]


class MoBitstats(C.Structure):
    _fields_ = [("hist", C.c_int32 * BIM_LAST), ("n_zero", C.c_int32), ("n_pos", C.c_int32),
                ("n_nan", C.c_int32), ("n_inf", C.c_int32), ("n_den", C.c_int32),
                ("vmin", C.c_float), ("vmax", C.c_float)]


class MoSigdist(C.Structure):
    _fields_ = [("bins", C.c_int32 * DIST_BIN), ("peak_cnt", C.c_int32), ("peak_bin", C.c_int32),
                ("avg", C.c_double), ("var_m", C.c_double), ("var_s", C.c_double),
                ("count", C.c_int64)]


class MoHist(C.Structure):
    _fields_ = [("histc", C.c_int * HIST_LEN), ("count", C.c_int), ("error", C.c_int)]


class MoEbu(C.Structure):
    """mo_ebu of oracle/mtr_oracle.h (MO_MAXCH = 5), for block-by-block use."""
    _fields_ = [("integr", C.c_int), ("nchan", C.c_int), ("fsamp", C.c_float), ("fragm", C.c_int), ("frcnt", C.c_int),
                ("frpwr", C.c_float), ("power", C.c_float * 64), ("wrind", C.c_int), ("div1", C.c_int), ("div2", C.c_int),
                ("loudness_M", C.c_float), ("maxloudn_M", C.c_float), ("loudness_S", C.c_float), ("maxloudn_S", C.c_float),
                ("integrated", C.c_float), ("integ_thr", C.c_float), ("range_min", C.c_float), ("range_max", C.c_float),
                ("range_thr", C.c_float), ("k", MoKw), ("z", (C.c_float * 4) * 5), ("hist_M", MoHist), ("hist_S", MoHist)]


class EbuStream:
    """One Ebu_r128_proc + two TruePeakdsp fed block by block, with the integration controls — what an
    EBUr128 plugin instance holds (src/ebulv2.cc:189-196)."""

    def __init__(self, lib, fs):
        self.lib, self.e, self.t = lib, MoEbu(), (MoTp(), MoTp())
        lib.mo_ebu_init.argtypes = [C.POINTER(MoEbu), C.c_int, C.c_float]
        lib.mo_ebu_process.argtypes = [C.POINTER(MoEbu), C.c_int, C.POINTER(C.c_void_p)]
        for f in ("mo_ebu_integr_start", "mo_ebu_integr_pause", "mo_ebu_integr_reset"):
            getattr(lib, f).argtypes = [C.POINTER(MoEbu)]
        lib.mo_tp_init.argtypes = [C.POINTER(MoTp), C.c_float]
        lib.mo_tp_process_max.argtypes = [C.POINTER(MoTp), _f32p, C.c_int]
        lib.mo_tp_read.argtypes = [C.POINTER(MoTp)]
        lib.mo_tp_read.restype = C.c_float
        lib.mo_ebu_init(C.byref(self.e), 2, fs)
        for t in self.t:
            lib.mo_tp_init(C.byref(t), fs)

    def start(self):
        self.lib.mo_ebu_integr_start(C.byref(self.e))

    def pause(self):
        self.lib.mo_ebu_integr_pause(C.byref(self.e))

    def reset(self):
        self.lib.mo_ebu_integr_reset(C.byref(self.e))

    def process(self, left, right, with_tp=True):
        """-> (out9, hist_M, hist_S, counts, (tp_l, tp_r) of this block or None)"""
        left = np.ascontiguousarray(left, np.float32)
        right = np.ascontiguousarray(right, np.float32)
        ptrs = (C.c_void_p * 2)(left.ctypes.data, right.ctypes.data)
        self.lib.mo_ebu_process(C.byref(self.e), left.size, ptrs)
        tp = None
        if with_tp:
            self.lib.mo_tp_process_max(C.byref(self.t[0]), left, left.size)
            self.lib.mo_tp_process_max(C.byref(self.t[1]), right, right.size)
            tp = (self.lib.mo_tp_read(C.byref(self.t[0])), self.lib.mo_tp_read(C.byref(self.t[1])))
        e = self.e
        out9 = np.array([e.loudness_M, e.maxloudn_M, e.loudness_S, e.maxloudn_S, e.integrated, e.integ_thr,
                         e.range_min, e.range_max, e.range_thr], np.float32)
        return (out9, np.array(e.hist_M.histc, np.int32), np.array(e.hist_S.histc, np.int32),
                (e.hist_M.count, e.hist_S.count), tp)


class TpStream:
    """One channel through TruePeakdsp::process_max + read, call by call (the per-call peak of the LV2 glue)."""

    def __init__(self, lib, fs):
        self.lib = lib
        self.t = MoTp()
        lib.mo_tp_init(C.byref(self.t), fs)

    def process(self, x):
        x = np.ascontiguousarray(x, np.float32)
        self.lib.mo_tp_process_max(C.byref(self.t), x, x.size)
        return self.lib.mo_tp_read(C.byref(self.t))


class Oracle(_Batch):
    def ebu_stream(self, fs=48000.0):
        return EbuStream(self.lib, fs)

    def tp_stream(self, fs=48000.0):
        return TpStream(self.lib, fs)

    def __init__(self):
        lib = C.CDLL(build_oracle())
        super().__init__(lib, "mo_")
        lib.mo_kw_init.argtypes = [C.POINTER(MoKw), C.c_float]
        lib.mo_tp_table.restype = C.POINTER(C.c_float)
        lib.mo_tp_init.argtypes = [C.POINTER(MoTp), C.c_float]
        lib.mo_tp_resample.argtypes = [C.POINTER(MoTp), _f32p, C.c_int, _f32p]
        lib.mo_tp_process.argtypes = [C.POINTER(MoTp), _f32p, C.c_int]
        lib.mo_tp_process_max.argtypes = [C.POINTER(MoTp), _f32p, C.c_int]
        lib.mo_tp_read.argtypes = [C.POINTER(MoTp)]
        lib.mo_tp_read.restype = C.c_float
        lib.mo_tp_read2.argtypes = [C.POINTER(MoTp), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        lib.mo_tp_reset.argtypes = [C.POINTER(MoTp)]
        lib.mo_vu_init.argtypes = [C.POINTER(MoVu), C.c_float]
        lib.mo_vu_process.argtypes = [C.POINTER(MoVu), _f32p, C.c_int]
        lib.mo_vu_read.argtypes = [C.POINTER(MoVu)]
        lib.mo_vu_read.restype = C.c_float
        lib.mo_band_setup.argtypes = [C.POINTER(MoBand), C.c_double, C.c_double, C.c_double, C.c_int]
        lib.mo_band_process.argtypes = [C.POINTER(MoBand), C.c_float]
        lib.mo_band_process.restype = C.c_float
        lib.mo_bitstats_reset.argtypes = [C.POINTER(MoBitstats)]
        lib.mo_bitstats_run.argtypes = [C.POINTER(MoBitstats), _f32p, C.c_uint32]
        lib.mo_sigdist_reset.argtypes = [C.POINTER(MoSigdist)]
        lib.mo_sigdist_run.argtypes = [C.POINTER(MoSigdist), _f32p, C.c_uint32]
        lib.mo_fill_lcg.argtypes = [_f32p, C.c_uint32, C.c_uint32, C.c_float]
        lib.mo_hist_reset.argtypes = [C.POINTER(MoHist)]
        lib.mo_hist_addpoint.argtypes = [C.POINTER(MoHist), C.c_float]
        lib.mo_hist_calc_integ.argtypes = [C.POINTER(MoHist), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        lib.mo_hist_calc_range.argtypes = [C.POINTER(MoHist), C.POINTER(C.c_float),
                                           C.POINTER(C.c_float), C.POINTER(C.c_float)]

    def kw_coef(self, fs):
        k = MoKw()
        self.lib.mo_kw_init(C.byref(k), fs)
        return np.array([k.a0, k.a1, k.a2, k.b1, k.b2, k.c3, k.c4], np.float32)

    def tp_table(self):
        return np.ctypeslib.as_array(self.lib.mo_tp_table(), shape=(120,)).copy()

    def tp_consts(self, fs):
        t = MoTp()
        self.lib.mo_tp_init(C.byref(t), fs)
        return np.array([t.w1, t.w2, t.w3, t.g], np.float32)

    def band_coef(self, rate, i):
        """36 doubles of band i (0..29) at `rate`, laid out [section][a0 a1 a2 b0 b1 b2]."""
        fb = MoBand()
        f_m = 2.0 ** ((i - 16) / 3.0) * 1000.0
        bw = f_m * 2.0 ** (1 / 6.0) - f_m * 2.0 ** (-1 / 6.0)
        self.lib.mo_band_setup(C.byref(fb), float(rate), f_m, bw, 6)
        return np.array([[fb.f[s].W[j] for j in range(6)] for s in range(6)], np.float64)

    def bitstats(self, x):
        b = MoBitstats()
        self.lib.mo_bitstats_reset(C.byref(b))
        x = np.ascontiguousarray(x, np.float32)
        self.lib.mo_bitstats_run(C.byref(b), x, x.size)
        return dict(hist=np.array(b.hist[:], np.int32),
                    counters=np.array([b.n_zero, b.n_pos, b.n_nan, b.n_inf, b.n_den], np.int32),
                    vmin=np.float32(b.vmin), vmax=np.float32(b.vmax))

    def sigdist(self, x):
        d = MoSigdist()
        self.lib.mo_sigdist_reset(C.byref(d))
        x = np.ascontiguousarray(x, np.float32)
        self.lib.mo_sigdist_run(C.byref(d), x, x.size)
        return dict(bins=np.array(d.bins[:], np.int32), peak_cnt=d.peak_cnt, peak_bin=d.peak_bin,
                    avg=d.avg, var_m=d.var_m, var_s=d.var_s, count=d.count)

    def fill_lcg(self, T, seed, gain=1.0):
        x = np.zeros((T, 2), np.float32)
        self.lib.mo_fill_lcg(x, T, seed, gain)
        return x

    def vu(self, x, fs=48000.0, block=None):
        """Returns the sequence of read() values after each block (block=None: one call)."""
        v = MoVu()
        self.lib.mo_vu_init(C.byref(v), fs)
        x = np.ascontiguousarray(x, np.float32)
        block = block or x.size
        out = []
        for p in range(0, x.size, block):
            seg = np.ascontiguousarray(x[p:p + block])
            self.lib.mo_vu_process(C.byref(v), seg, seg.size)
            out.append(self.lib.mo_vu_read(C.byref(v)))
        return np.array(out, np.float32)

    def tp_process_seq(self, x, fs=48000.0, block=1024):
        """Mono x through TruePeakdsp::process in blocks; returns [(m, p)] after each read."""
        t = MoTp()
        self.lib.mo_tp_init(C.byref(t), fs)
        x = np.ascontiguousarray(x, np.float32)
        out = []
        m, p = C.c_float(), C.c_float()
        for q in range(0, x.size, block):
            seg = np.ascontiguousarray(x[q:q + block])
            self.lib.mo_tp_process(C.byref(t), seg, seg.size)
            self.lib.mo_tp_read2(C.byref(t), C.byref(m), C.byref(p))
            out.append((m.value, p.value))
        return np.array(out, np.float32)

    def tp_resample(self, x, fs=48000.0):
        t = MoTp()
        self.lib.mo_tp_init(C.byref(t), fs)
        x = np.ascontiguousarray(x, np.float32)
        y = np.zeros(4 * x.size, np.float32)
        self.lib.mo_tp_resample(C.byref(t), x, x.size, y)
        return y


class Reference(_Batch):
    """The reference's own objects; only available where /root/reference could be compiled."""

    def __init__(self):
        if not os.path.exists(REF_SO):
            raise FileNotFoundError(REF_SO)
        lib = C.CDLL(REF_SO)
        super().__init__(lib, "ref_")
        lib.ref_ebu_new.restype = C.c_void_p
        lib.ref_ebu_new.argtypes = [C.c_int, C.c_float]
        lib.ref_ebu_free.argtypes = [C.c_void_p]
        lib.ref_ebu_coef.argtypes = [C.c_void_p, _f32p]
        lib.ref_tp_new.restype = C.c_void_p
        lib.ref_tp_new.argtypes = [C.c_float]
        lib.ref_tp_free.argtypes = [C.c_void_p]
        lib.ref_tp_table.argtypes = [C.c_void_p, _f32p]
        lib.ref_tp_consts.argtypes = [C.c_void_p, _f32p]
        lib.ref_tp_process.argtypes = [C.c_void_p, _f32p, C.c_int]
        lib.ref_tp_process_max.argtypes = [C.c_void_p, _f32p, C.c_int]
        lib.ref_tp_read.argtypes = [C.c_void_p]
        lib.ref_tp_read.restype = C.c_float
        lib.ref_tp_read2.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        lib.ref_tp_lastbuf.argtypes = [C.c_void_p, _f32p, C.c_int]
        lib.ref_tp_resampler_state.argtypes = [C.c_void_p] + [C.POINTER(C.c_uint)] * 3
        lib.ref_band_new.restype = C.c_void_p
        lib.ref_band_new.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int]
        lib.ref_band_free.argtypes = [C.c_void_p]
        lib.ref_band_coef.argtypes = [C.c_void_p, _f64p]
        lib.ref_vu_new.restype = C.c_void_p
        lib.ref_vu_new.argtypes = [C.c_float]
        lib.ref_vu_free.argtypes = [C.c_void_p]
        lib.ref_vu_process.argtypes = [C.c_void_p, _f32p, C.c_int]
        lib.ref_vu_read.argtypes = [C.c_void_p]
        lib.ref_vu_read.restype = C.c_float

    def kw_coef(self, fs):
        h = self.lib.ref_ebu_new(2, fs)
        out = np.zeros(7, np.float32)
        self.lib.ref_ebu_coef(h, out)
        self.lib.ref_ebu_free(h)
        return out

    def tp_table(self):
        h = self.lib.ref_tp_new(48000.0)
        out = np.zeros(120, np.float32)
        self.lib.ref_tp_table(h, out)
        self.lib.ref_tp_free(h)
        return out

    def tp_consts(self, fs):
        h = self.lib.ref_tp_new(fs)
        out = np.zeros(4, np.float32)
        self.lib.ref_tp_consts(h, out)
        self.lib.ref_tp_free(h)
        return out

    def tp_resampler_state(self, fs=48000.0):
        h = self.lib.ref_tp_new(fs)
        a, b, c = C.c_uint(), C.c_uint(), C.c_uint()
        self.lib.ref_tp_resampler_state(h, C.byref(a), C.byref(b), C.byref(c))
        self.lib.ref_tp_free(h)
        return a.value, b.value, c.value

    def band_coef(self, rate, i):
        f_m = 2.0 ** ((i - 16) / 3.0) * 1000.0
        bw = f_m * 2.0 ** (1 / 6.0) - f_m * 2.0 ** (-1 / 6.0)
        h = self.lib.ref_band_new(float(rate), f_m, bw, 6)
        out = np.zeros(36, np.float64)
        self.lib.ref_band_coef(h, out)
        self.lib.ref_band_free(h)
        return out.reshape(6, 6)

    def vu(self, x, fs=48000.0, block=None):
        h = self.lib.ref_vu_new(fs)
        x = np.ascontiguousarray(x, np.float32)
        block = block or x.size
        out = []
        for p in range(0, x.size, block):
            seg = np.ascontiguousarray(x[p:p + block])
            self.lib.ref_vu_process(h, seg, seg.size)
            out.append(self.lib.ref_vu_read(h))
        self.lib.ref_vu_free(h)
        return np.array(out, np.float32)

    def tp_process_seq(self, x, fs=48000.0, block=1024):
        h = self.lib.ref_tp_new(fs)
        x = np.ascontiguousarray(x, np.float32)
        out = []
        m, p = C.c_float(), C.c_float()
        for q in range(0, x.size, block):
            seg = np.ascontiguousarray(x[q:q + block])
            self.lib.ref_tp_process(h, seg, seg.size)
            self.lib.ref_tp_read2(h, C.byref(m), C.byref(p))
            out.append((m.value, p.value))
        self.lib.ref_tp_free(h)
        return np.array(out, np.float32)

    def tp_resample(self, x, fs=48000.0):
        h = self.lib.ref_tp_new(fs)
        x = np.ascontiguousarray(x, np.float32)
        ys = []
        for q in range(0, x.size, 8192):
            seg = np.ascontiguousarray(x[q:q + 8192])
            self.lib.ref_tp_process_max(h, seg, seg.size)
            y = np.zeros(4 * seg.size, np.float32)
            self.lib.ref_tp_lastbuf(h, y, y.size)
            ys.append(y)
        self.lib.ref_tp_free(h)
        return np.concatenate(ys)


def have_reference():
    return os.path.exists(REF_SO)
