"""ctypes binding of include/mtr_engine.h (libmtr_engine.so).

Names follow the reference's classes: Engine.integr_start/.integr_pause/.integr_reset mirror
Ebu_r128_proc (ebumeter/ebu_r128_proc.h:77-79), Engine.results() returns the getters of
ebu_r128_proc.h:81-94 per stream, Engine.process() is the batched process()/process_max()/
spectrum_run inner loop.  Device buffers are passed as raw pointers (torch tensors' data_ptr()).
"""
import ctypes as C
import os
import re

import numpy as np

METER_EBU, METER_TRUEPEAK, METER_SPECTR30, METER_TPBALLIST = 0x01, 0x02, 0x04, 0x08
METER_BITSTATS, METER_SIGDIST, METER_DR14, METER_KMETER = 0x10, 0x20, 0x40, 0x80
BIM_LAST, DIST_BIN = 584, 361
HIST_LEN, NBANDS = 751, 30

_HERE = os.path.dirname(os.path.abspath(__file__))
# MTR_LIB: an alternative build of the same library (instrumented kernels, tools/f4_prof.py); never a different backend
lib_path = os.environ.get("MTR_LIB") or os.path.join(_HERE, "lib", "libmtr_engine.so")


class EngineError(RuntimeError):
    """A C-ABI call returned a status other than MTR_OK; `.code` is that status (ERR_TIMEOUT, ERR_STATE, ...)."""
    code = 0


ABI_VERSION = 2            # MTR_ABI_VERSION of include/mtr_engine.h this file binds

ERR_ARG, ERR_UNSUPPORTED, ERR_NODEVICE, ERR_HIP, ERR_NOMEM, ERR_TIMEOUT, ERR_STATE = -1, -2, -3, -4, -5, -6, -7


class _Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("meters", C.c_uint32), ("n_streams", C.c_uint32),
                ("n_channels", C.c_uint32), ("sample_rate", C.c_float), ("device", C.c_int32),
                ("max_frames", C.c_uint32), ("tune_run", C.c_uint32), ("tune_segments", C.c_uint32),
                ("tune_layout", C.c_uint32), ("tune_fir", C.c_uint32), ("tune_prune", C.c_uint32)]


class PlanInfo(C.Structure):
    """mtr_plan_info: how an engine would tile and route a call (mtr_plan_query: host arithmetic, no device)."""
    _fields_ = [(n, C.c_uint32) for n in ("layout", "uses_seg", "head_frames", "body_fragments", "segments", "fragments_per_lane",
                                          "warm_steps", "n_tiles", "head_tiles", "n_fragments_ended", "kw_segments", "frames_left_after")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class Dr14Result(C.Structure):
    """mtr_dr14_result: what dr14_run leaves on the dr14 plugins' ports in dr_operation_mode."""
    _fields_ = [("m_rms", C.c_float * 2), ("m_peak", C.c_float * 2), ("dr", C.c_float * 2),
                ("dr_total", C.c_float), ("block_count", C.c_float)]


class StreamResult(C.Structure):
    """mtr_stream_result: Ebu_r128_proc's getters in declaration order, then true peak."""
    _fields_ = [("loudness_M", C.c_float), ("maxloudn_M", C.c_float), ("loudness_S", C.c_float),
                ("maxloudn_S", C.c_float), ("integrated", C.c_float), ("integ_thr", C.c_float),
                ("range_min", C.c_float), ("range_max", C.c_float), ("range_thr", C.c_float),
                ("hist_M_count", C.c_int32), ("hist_S_count", C.c_int32),
                ("truepeak", C.c_float * 2), ("truepeak_call", C.c_float * 2),
                ("tpb_level", C.c_float * 2), ("tpb_peak", C.c_float * 2)]


def _preload_hip_runtime():
    """One HIP runtime and one RCCL per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so and librccl.so
    (same SONAMEs as /opt/rocm's); if our library pulled in the system copies first and torch then loaded
    its own, the second runtime finds no GPU.  When torch is installed (it is only plumbing here:
    device buffers, torch.distributed) load ITS copies first so both sides share them; without torch
    the library's rpath (/opt/rocm/lib) applies."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.submodule_search_locations:
        for name in ("libamdhip64.so", "librccl.so"):
            cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", name)
            if os.path.exists(cand):
                try:
                    C.CDLL(cand, mode=C.RTLD_GLOBAL if name == "libamdhip64.so" else C.RTLD_LOCAL)
                except OSError:
                    pass                      # the system copy (rpath) serves


def _load():
    if not os.path.exists(lib_path):
        raise ImportError(
            f"{lib_path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). meters.lv2_amd has no CPU or pure-Python fallback.")
    _preload_hip_runtime()
    L = C.CDLL(lib_path)
    vp, u32, u64, i32, f32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32, C.c_float
    L.mtr_last_error.restype = C.c_char_p
    L.mtr_version.restype = C.c_char_p
    # the entry points bound below are those of ABI version ABI_VERSION: an older library (MTR_LIB pointing at a stale
    # build) is named as such here instead of failing somewhere below with an AttributeError
    have = L.mtr_abi_version() if hasattr(L, "mtr_abi_version") else 0
    if have < ABI_VERSION:
        raise ImportError(f"{lib_path} speaks ABI version {have}, this binding needs {ABI_VERSION} (include/mtr_engine.h): rebuild it "
                          "with `python -c 'import __graft_entry__ as g; g.build()'`")
    L.mtr_engine_create.argtypes = [C.POINTER(_Config), C.POINTER(vp)]
    L.mtr_engine_destroy.argtypes = [vp]
    L.mtr_engine_destroy.restype = None
    for n in ("reset", "integr_start", "integr_pause", "integr_reset", "truepeak_reset",
              "spectr_reset_peak", "sync"):
        getattr(L, "mtr_engine_" + n).argtypes = [vp]
    L.mtr_engine_spectr_set_speed.argtypes = [vp, f32]
    L.mtr_engine_process_device.argtypes = [vp, vp, u64, u64, vp]
    L.mtr_engine_process_host.argtypes = [vp, vp, u64, u64]
    L.mtr_engine_set_host_chunk_bytes.argtypes = [vp, u64]
    L.mtr_engine_process_planar_host.argtypes = [vp, C.POINTER(vp), u32]
    L.mtr_engine_results.argtypes = [vp, u32, u32, C.POINTER(StreamResult)]
    L.mtr_engine_histograms.argtypes = [vp, u32, u32, vp, vp]
    L.mtr_engine_fragment_powers.argtypes = [vp, u32, u32, vp, u32, C.POINTER(u32)]
    L.mtr_engine_spectrum.argtypes = [vp, u32, u32, vp, vp, vp, vp]
    L.mtr_engine_aggregate_device.argtypes = [vp, vp, vp, vp]
    L.mtr_engine_bitstats.argtypes = [vp, u32, u32, vp, vp, vp]
    L.mtr_engine_sigdist.argtypes = [vp, u32, u32, vp, vp, vp, vp]
    L.mtr_engine_intstat_reset.argtypes = [vp]
    L.mtr_engine_dr14_results.argtypes = [vp, u32, u32, vp]
    L.mtr_engine_dr14_reset.argtypes = [vp]
    L.mtr_engine_kmeter_read.argtypes = [vp, u32, u32, vp, vp]
    L.mtr_engine_kmeter_reset.argtypes = [vp]
    L.mtr_engine_prune_stats.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    L.mtr_engine_refine_stats.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    L.mtr_engine_layout.argtypes = [vp]
    L.mtr_engine_seg_stats.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    L.mtr_plan_query.argtypes = [vp, u32, u64, u32, vp]
    L.mtr_comm_unique_id.argtypes = [vp]
    L.mtr_comm_init.argtypes = [C.POINTER(vp), i32, i32, vp, i32]
    L.mtr_comm_init_timeout.argtypes = [C.POINTER(vp), i32, i32, vp, i32, u32, C.POINTER(f32)]
    L.mtr_comm_probe.argtypes = [vp, u32, C.POINTER(f32)]
    L.mtr_comm_set_timeout.argtypes = [vp, u32]
    L.mtr_comm_nranks.argtypes = [vp]
    L.mtr_comm_device.argtypes = [vp]
    L.mtr_engine_set_deferred_tail.argtypes = [vp, C.c_int]
    L.mtr_engine_join.argtypes = [vp, vp]
    L.mtr_engine_deferred_stats.argtypes = [vp, C.POINTER(u64)]
    L.mtr_rccl_version.argtypes = []
    L.mtr_engine_state_bytes.argtypes = [vp, u32]
    L.mtr_engine_state_bytes.restype = C.c_size_t
    L.mtr_engine_state_export.argtypes = [vp, u32, u32, vp, C.c_size_t]
    L.mtr_engine_state_import.argtypes = [vp, u32, vp, C.c_size_t]
    L.mtr_state_blob_count.argtypes = [vp, C.c_size_t]
    L.mtr_state_blob_count.restype = u32
    L.mtr_comm_destroy.argtypes = [vp]
    L.mtr_comm_destroy.restype = None
    L.mtr_engine_reduce.argtypes = [vp, vp, vp, vp, vp]
    L.mtr_hist_loudness.argtypes = [vp, vp] + [C.POINTER(f32)] * 5
    L.mtr_hist_loudness.restype = None
    L.mtr_engine_timing_enable.argtypes = [vp, C.c_int]
    L.mtr_engine_timing_query.argtypes = [vp, C.POINTER(f32), C.POINTER(f32), C.POINTER(f32), C.POINTER(u32)]
    L.mtr_engine_timing_calls.argtypes = [vp, vp, u32, C.POINTER(u32)]
    L.mtr_kweight_coef.argtypes = [f32, vp]
    L.mtr_fir_table.argtypes = [vp]
    L.mtr_band_coef.argtypes = [C.c_double, u32, vp]
    L.mtr_synth_fill_device.argtypes = [vp, u32, u64, u64, u32, f32, C.c_int, vp]
    return L


lib = _load()
if b"TIMING-ONLY" in lib.mtr_version() and os.environ.get("MTR_ALLOW_TIMING_ONLY_BUILD") != "1":
    # a library built with -DMTR_TIMING_ONLY_BUILD may carry kernels with a role switched off (tools/: elimination runs): its
    # results are wrong by construction, and only the timing tools — which set the variable — may load it
    raise ImportError(f"{lib_path} is a TIMING-ONLY build ({lib.mtr_version().decode()}): it computes wrong results and is only "
                      "for the elimination runs under tools/ (MTR_ALLOW_TIMING_ONLY_BUILD=1)")


def _check(rc, what):
    if rc != 0:
        err = EngineError(f"{what} failed ({rc}): {lib.mtr_last_error().decode()}")
        err.code = rc
        raise err


def exported_symbols():
    """Every function include/mtr_engine.h declares, parsed from the header itself."""
    hdr = os.path.join(os.path.dirname(_HERE), "include", "mtr_engine.h")
    txt = re.sub(r"/\*.*?\*/", "", open(hdr).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(mtr_[a-z0-9_]+)\s*\(", txt)))


def kweight_coef(fs):
    out = np.zeros(7, np.float32)
    _check(lib.mtr_kweight_coef(fs, out.ctypes.data), "mtr_kweight_coef")
    return out


def fir_table():
    out = np.zeros(120, np.float32)
    _check(lib.mtr_fir_table(out.ctypes.data), "mtr_fir_table")
    return out


def band_coef(rate, band):
    out = np.zeros(36, np.float64)
    _check(lib.mtr_band_coef(float(rate), band, out.ctypes.data), "mtr_band_coef")
    return out.reshape(6, 6)


def hist_loudness(hist_M, hist_S):
    """(integrated, integ_thr, range_min, range_max, range_thr) of (summed) histograms."""
    hm = np.ascontiguousarray(hist_M, np.int32)
    hs = np.ascontiguousarray(hist_S, np.int32)
    o = [C.c_float() for _ in range(5)]
    lib.mtr_hist_loudness(hm.ctypes.data, hs.ctypes.data, *[C.byref(x) for x in o])
    return tuple(x.value for x in o)


def plan_query(n_streams, n_frames, sample_rate=48000.0, meters=METER_EBU | METER_TRUEPEAK, frames_left_in_fragment=0, n_slots=0,
               n_channels=2, tune_run=0, tune_segments=0, tune_layout=0, tune_fir=0, tune_prune=0):
    """How an engine of this configuration would tile and route a call (no GPU needed)."""
    cfg = _Config(struct_size=C.sizeof(_Config), meters=meters, n_streams=n_streams, n_channels=n_channels, sample_rate=sample_rate,
                  device=0, max_frames=0, tune_run=tune_run, tune_segments=tune_segments, tune_layout=tune_layout, tune_fir=tune_fir,
                  tune_prune=tune_prune)
    info = PlanInfo()
    _check(lib.mtr_plan_query(C.byref(cfg), frames_left_in_fragment, n_frames, n_slots, C.byref(info)), "mtr_plan_query")
    return info.as_dict()


def synth_fill_device(ptr, n_streams, n_frames, stride, seed, fs=48000.0, kind=1, stream=0):
    _check(lib.mtr_synth_fill_device(ptr, n_streams, n_frames, stride, seed, fs, kind, stream),
           "mtr_synth_fill_device")


COMM_ID_BYTES = 128


def comm_unique_id():
    """Rank 0: the 128-byte RCCL id every rank needs for Comm(...)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _check(lib.mtr_comm_unique_id(buf), "comm_unique_id")
    return bytes(buf.raw)


def rccl_version():
    """ncclGetVersion of the RCCL this process runs, e.g. 22606."""
    v = lib.mtr_rccl_version()
    if v < 0:
        _check(v, "rccl_version")
    return v


class Comm:
    """mtr_comm: one RCCL communicator per rank (ncclCommInitRank is collective: every rank constructs its own).
    timeout_ms > 0: mtr_comm_init_timeout — the creation (and every later call on the communicator) is polled to that
    deadline and raises EngineError with code ERR_TIMEOUT instead of hanging; `.init_ms` is how long the creation took."""

    def __init__(self, rank, world, unique_id, device=0, timeout_ms=0):
        self._h = C.c_void_p()
        self.world = world
        buf = C.create_string_buffer(bytes(unique_id), COMM_ID_BYTES)
        ms = C.c_float()
        _check(lib.mtr_comm_init_timeout(C.byref(self._h), rank, world, buf, device, int(timeout_ms), C.byref(ms)), "comm_init")
        self.init_ms = ms.value

    def probe(self, timeout_ms=0):
        """The job's first collective (one 4-byte all-reduce on the communicator's own stream), bounded; returns its ms."""
        ms = C.c_float()
        _check(lib.mtr_comm_probe(self._h, int(timeout_ms), C.byref(ms)), "comm_probe")
        return ms.value

    def set_timeout(self, timeout_ms):
        _check(lib.mtr_comm_set_timeout(self._h, int(timeout_ms)), "comm_set_timeout")

    def nranks(self):
        """ncclCommCount: the ranks RCCL itself sees in this communicator."""
        n = lib.mtr_comm_nranks(self._h)
        if n < 0:
            _check(n, "comm_nranks")
        return n

    def device(self):
        """ncclCommCuDevice: the device RCCL bound this rank to."""
        d = lib.mtr_comm_device(self._h)
        if d < 0:
            _check(d, "comm_device")
        return d

    def close(self):
        if self._h:
            lib.mtr_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class Engine:
    def __init__(self, n_streams, sample_rate=48000.0, meters=METER_EBU | METER_TRUEPEAK,
                 n_channels=2, device=0, tune_run=0, tune_segments=0, tune_layout=0, tune_fir=0, tune_prune=0):
        cfg = _Config(struct_size=C.sizeof(_Config), meters=meters, n_streams=n_streams,
                      n_channels=n_channels, sample_rate=sample_rate, device=device,
                      max_frames=0, tune_run=tune_run, tune_segments=tune_segments,
                      tune_layout=tune_layout, tune_fir=tune_fir, tune_prune=tune_prune)
        self._h = C.c_void_p()
        self.n_streams, self.meters, self.sample_rate = n_streams, meters, sample_rate
        _check(lib.mtr_engine_create(C.byref(cfg), C.byref(self._h)), "mtr_engine_create")

    def close(self):
        if getattr(self, "_h", None):
            lib.mtr_engine_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def reset(self):
        _check(lib.mtr_engine_reset(self._h), "reset")

    def integr_start(self):
        _check(lib.mtr_engine_integr_start(self._h), "integr_start")

    def integr_pause(self):
        _check(lib.mtr_engine_integr_pause(self._h), "integr_pause")

    def integr_reset(self):
        _check(lib.mtr_engine_integr_reset(self._h), "integr_reset")

    def truepeak_reset(self):
        _check(lib.mtr_engine_truepeak_reset(self._h), "truepeak_reset")

    def spectr_set_speed(self, v):
        _check(lib.mtr_engine_spectr_set_speed(self._h, v), "spectr_set_speed")

    def spectr_reset_peak(self):
        _check(lib.mtr_engine_spectr_reset_peak(self._h), "spectr_reset_peak")

    def process_device(self, ptr, n_frames, stride=None, stream=0):
        _check(lib.mtr_engine_process_device(self._h, ptr, n_frames, stride or n_frames, stream), "process_device")

    def process(self, x):
        """x: host float32 [S, T, 2] (or [S, T] mono)."""
        x = np.ascontiguousarray(x, np.float32)
        assert x.shape[0] == self.n_streams
        _check(lib.mtr_engine_process_host(self._h, x.ctypes.data, x.shape[1], x.shape[1]), "process_host")

    def set_host_chunk_bytes(self, n):
        """Bytes of audio per chunk of process() (host memory crosses the link chunk by chunk under the kernels)."""
        _check(lib.mtr_engine_set_host_chunk_bytes(self._h, n), "set_host_chunk_bytes")

    def process_planar(self, chans):
        arrs = [np.ascontiguousarray(c, np.float32) for c in chans]
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        _check(lib.mtr_engine_process_planar_host(self._h, ptrs, arrs[0].size), "process_planar_host")

    def sync(self):
        _check(lib.mtr_engine_sync(self._h), "sync")

    def set_deferred_tail(self, mode):
        """0 auto / 1 never / 2 always: k_gate (and the reduction) of a call on the engine's side stream, beside the next call."""
        _check(lib.mtr_engine_set_deferred_tail(self._h, int(mode)), "set_deferred_tail")

    def join(self, stream=0):
        """`stream` waits for what the engine's side stream holds (before reading reduce()'s buffers in stream order)."""
        _check(lib.mtr_engine_join(self._h, stream), "join")

    def deferred_calls(self):
        n = C.c_uint64()
        _check(lib.mtr_engine_deferred_stats(self._h, C.byref(n)), "deferred_stats")
        return n.value

    def results(self, first=0, count=None):
        count = self.n_streams - first if count is None else count
        out = (StreamResult * count)()
        _check(lib.mtr_engine_results(self._h, first, count, out), "results")
        return out

    def out9(self, first=0, count=None):
        """[count, 9] float32: M, maxM, S, maxS, I, I_thr, Rmin, Rmax, R_thr."""
        r = self.results(first, count)
        return np.array([[getattr(x, f[0]) for f in StreamResult._fields_[:9]] for x in r], np.float32)

    def truepeak(self, first=0, count=None):
        r = self.results(first, count)
        return np.array([[x.truepeak[0], x.truepeak[1]] for x in r], np.float32)

    def histograms(self, first=0, count=None):
        count = self.n_streams - first if count is None else count
        hm = np.zeros((count, HIST_LEN), np.int32)
        hs = np.zeros((count, HIST_LEN), np.int32)
        _check(lib.mtr_engine_histograms(self._h, first, count, hm.ctypes.data, hs.ctypes.data), "histograms")
        return hm, hs

    def fragment_powers(self, first=0, count=None):
        count = self.n_streams - first if count is None else count
        n = C.c_uint32()
        _check(lib.mtr_engine_fragment_powers(self._h, first, count, None, 0, C.byref(n)), "fragment_powers")
        out = np.zeros((count, max(n.value, 1)), np.float32)
        _check(lib.mtr_engine_fragment_powers(self._h, first, count, out.ctypes.data, out.shape[1], C.byref(n)),
               "fragment_powers")
        return out[:, :n.value]

    def spectrum(self, first=0, count=None):
        count = self.n_streams - first if count is None else count
        a = [np.zeros((count, NBANDS), np.float32) for _ in range(4)]
        _check(lib.mtr_engine_spectrum(self._h, first, count, *[x.ctypes.data for x in a]), "spectrum")
        return dict(val=a[0], max=a[1], val_db=a[2], max_db=a[3])

    def bitstats(self, first=0, count=None):
        count = self.n_streams - first if count is None else count
        hist = np.zeros((count, BIM_LAST), np.int32)
        cnt = np.zeros((count, 5), np.int32)
        mm = np.zeros((count, 2), np.float32)
        _check(lib.mtr_engine_bitstats(self._h, first, count, hist.ctypes.data, cnt.ctypes.data, mm.ctypes.data), "bitstats")
        return dict(hist=hist, counters=cnt, vmin=mm[:, 0], vmax=mm[:, 1])

    def dr14(self, first=0, count=None):
        count = self.n_streams - first if count is None else count
        out = (Dr14Result * count)()
        _check(lib.mtr_engine_dr14_results(self._h, first, count, out), "dr14_results")
        return out

    def kmeter_read(self, first=0, count=None):
        """Kmeterdsp::read for every stream: (rms, peak) as [count, 2] arrays; arms the new-maximum flag."""
        count = self.n_streams - first if count is None else count
        rms = np.zeros((count, 2), np.float32)
        peak = np.zeros((count, 2), np.float32)
        _check(lib.mtr_engine_kmeter_read(self._h, first, count, rms.ctypes.data, peak.ctypes.data), "kmeter_read")
        return rms, peak

    def kmeter_reset(self):
        _check(lib.mtr_engine_kmeter_reset(self._h), "kmeter_reset")

    def dr14_reset(self):
        _check(lib.mtr_engine_dr14_reset(self._h), "dr14_reset")

    def sigdist(self, first=0, count=None):
        count = self.n_streams - first if count is None else count
        bins = np.zeros((count, DIST_BIN), np.int32)
        peak = np.zeros((count, 2), np.int32)
        mom = np.zeros((count, 3), np.float64)
        n = np.zeros(count, np.int64)
        _check(lib.mtr_engine_sigdist(self._h, first, count, bins.ctypes.data, peak.ctypes.data, mom.ctypes.data,
                                      n.ctypes.data), "sigdist")
        return dict(bins=bins, peak_cnt=peak[:, 0], peak_bin=peak[:, 1], avg=mom[:, 0], var_m=mom[:, 1],
                    var_s=mom[:, 2], count=n)

    def intstat_reset(self):
        _check(lib.mtr_engine_intstat_reset(self._h), "intstat_reset")

    def state_bytes(self, count):
        return int(lib.mtr_engine_state_bytes(self._h, count))

    def state_export(self, first=0, count=None):
        """Everything streams [first, first + count) carry from call to call, as one opaque blob (bytes)."""
        count = self.n_streams - first if count is None else count
        buf = C.create_string_buffer(self.state_bytes(count))
        _check(lib.mtr_engine_state_export(self._h, first, count, buf, len(buf)), "state_export")
        return bytes(buf.raw)

    def state_import(self, blob, first=0):
        """Put a blob's streams into slots [first, first + its count) of this engine (same meters / channels / rate)."""
        blob = bytes(blob)
        _check(lib.mtr_engine_state_import(self._h, first, blob, len(blob)), "state_import")
        return int(lib.mtr_state_blob_count(blob, len(blob)))

    def aggregate_device(self, hist_ptr, max_ptr, stream=0):
        _check(lib.mtr_engine_aggregate_device(self._h, hist_ptr, max_ptr, stream), "aggregate_device")

    def layout(self):
        return lib.mtr_engine_layout(self._h)

    def seg_stats(self):
        """layout 7: (calls whose whole fragments ran k_seg, frames per stream it covered)"""
        a, b = C.c_uint64(), C.c_uint64()
        _check(lib.mtr_engine_seg_stats(self._h, C.byref(a), C.byref(b)), "seg_stats")
        return a.value, b.value

    def reduce(self, comm, hist_ptr, max_ptr, stream=0):
        """aggregate_device + the RCCL all-reduce across the ranks of `comm` (a Comm), in place, on `stream`."""
        _check(lib.mtr_engine_reduce(self._h, comm._h, hist_ptr, max_ptr, stream), "reduce")

    def prune_stats(self):
        a, b = C.c_uint64(), C.c_uint64()
        _check(lib.mtr_engine_prune_stats(self._h, C.byref(a), C.byref(b)), "prune_stats")
        return a.value, b.value

    def refine_stats(self):
        """tune_prune = 2: (channel-blocks screened with the first product, completed with the other two)"""
        a, b = C.c_uint64(), C.c_uint64()
        _check(lib.mtr_engine_refine_stats(self._h, C.byref(a), C.byref(b)), "refine_stats")
        return a.value, b.value

    def timing_enable(self, on=True):
        _check(lib.mtr_engine_timing_enable(self._h, int(on)), "timing_enable")

    def timing_calls(self, cap=4096):
        """[calls, 4] float32 ms per timed call since the last query: fused, gate, behind the gate, whole call."""
        n = C.c_uint32()
        out = np.zeros((cap, 4), np.float32)
        _check(lib.mtr_engine_timing_calls(self._h, out.ctypes.data, cap, C.byref(n)), "timing_calls")
        return out[:min(n.value, cap)]

    def timing_query(self):
        f, g, b, n = C.c_float(), C.c_float(), C.c_float(), C.c_uint32()
        _check(lib.mtr_engine_timing_query(self._h, C.byref(f), C.byref(g), C.byref(b), C.byref(n)), "timing_query")
        return dict(ms_fused=f.value, ms_gate=g.value, ms_bank=b.value, calls=n.value)
