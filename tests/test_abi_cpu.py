"""CPU-side checks of the product library: it loads, exports every symbol include/mtr_engine.h
declares, its host-side set-up math matches the golden vectors bit-for-bit, and it refuses to
run without a GPU instead of falling back to anything.  No compute kernels are launched."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import meters.lv2_amd as M

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "golden_v1.npz"))


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32 if a.dtype == np.float32 else np.uint64)


def test_library_exports_every_declared_symbol():
    names = M.exported_symbols()
    assert len(names) >= 28
    for n in names:
        assert hasattr(M.lib, n), f"{n} declared in include/mtr_engine.h but not exported"
    assert M.lib.mtr_abi_version() == M.engine.ABI_VERSION == 2
    hdr = open(os.path.join(os.path.dirname(HERE), "include", "mtr_engine.h")).read()
    assert re.search(r"#define\s+MTR_ABI_VERSION\s+2\b", hdr)
    for n in ("mtr_engine_set_deferred_tail", "mtr_engine_join", "mtr_engine_deferred_stats", "mtr_comm_nranks", "mtr_comm_device"):
        assert n in names                                   # (round 6)


def test_product_never_touches_the_oracle():
    """Nothing the product ships (package, public headers) names the oracle, the reference-built checker
    (oracle/_ref/libmeters_ref.so — it travels to the GPU box for bench.py's cpu_baseline) or loads a library by
    name at run time: the only dlopen-like call allowed is the package's own ctypes.CDLL of libmtr_engine.so."""
    root = os.path.dirname(HERE)
    banned = (r"mtr_oracle", r"oracle/", r"_oracle", r"_ref\b", r"libmeters_ref", r"dlopen", r"dlsym")
    for top in ("meters.lv2_amd", "include"):
        for dirpath, dirs, files in os.walk(os.path.join(root, top)):
            dirs[:] = [d for d in dirs if d not in ("lib", "lib_prof", "__pycache__")]      # built artefacts
            for f in files:
                if f.endswith((".py", ".c", ".h", ".hip", ".cc", "Makefile")):
                    txt = open(os.path.join(dirpath, f), errors="ignore").read()
                    for word in banned:
                        assert not re.search(word, txt), (dirpath, f, word)
                    if f.endswith(".py"):
                        assert txt.count("CDLL(") <= (2 if f == "engine.py" else 0), (dirpath, f)


def test_setup_math_matches_reference_golden():
    for j, r in enumerate(G["rates"]):
        assert np.array_equal(_bits(M.kweight_coef(float(r))), _bits(G["kw_coef"][j]))
        for b in range(30):
            assert np.array_equal(_bits(M.band_coef(float(r), b)), _bits(G["band_coef"][j, b])), (r, b)
    assert np.array_equal(_bits(M.fir_table()), _bits(G["tp_table"]))


def test_hist_loudness_matches_oracle(oracle):
    from _oracle import MoHist
    rng = np.random.default_rng(5)
    for trial in range(400):
        # (round 6: mtr_setup_hist_loudness is the repo's own formulation of the gated mean and the percentile walk — held bit for
        # bit against the pinned oracle over narrow and wide programmes, a few points and summed histograms of a million, gates
        # that fall below bin 0 and above the loudest point, counts either side of the 50 / 20 point minima)
        centre = rng.integers(40, 745)
        width = [3, 40, 150][trial % 3]
        size = [rng.integers(10, 400), rng.integers(15, 60), rng.integers(100000, 1000000)][(trial // 3) % 3]
        pts = np.clip(rng.normal(centre, width, size=size).astype(int), 0, 750)
        hm = np.bincount(pts, minlength=751).astype(np.int32)
        hs = np.bincount(np.clip(pts + rng.integers(-30, 30, pts.size), 0, 750), minlength=751).astype(np.int32)
        got = M.hist_loudness(hm, hs)
        h = MoHist()
        for i in range(751):
            h.histc[i] = int(hm[i])
        h.count = int(hm.sum())
        vi, th = C.c_float(-200.0), C.c_float(-200.0)
        oracle.lib.mo_hist_calc_integ(C.byref(h), C.byref(vi), C.byref(th))
        for i in range(751):
            h.histc[i] = int(hs[i])
        h.count = int(hs.sum())
        v0, v1, t2 = C.c_float(-200.0), C.c_float(-200.0), C.c_float(-200.0)
        oracle.lib.mo_hist_calc_range(C.byref(h), C.byref(v0), C.byref(v1), C.byref(t2))
        want = (vi.value, th.value, v0.value, v1.value, t2.value)
        assert np.array_equal(_bits(np.array(got, np.float32)), _bits(np.array(want, np.float32))), trial


def test_bad_arguments_are_rejected():
    assert M.lib.mtr_kweight_coef(48000.0, None) == -1
    assert M.lib.mtr_band_coef(48000.0, 30, np.zeros(36).ctypes.data) == -1
    assert M.lib.mtr_engine_create(None, None) == -1
    assert b"mtr_engine_create" in M.lib.mtr_last_error()
    # a configuration of another ABI's size is refused by the planner as by create ()
    import ctypes as C
    from meters.lv2_amd import engine as E
    cfg = E._Config(struct_size=C.sizeof(E._Config) - 4, meters=3, n_streams=8, n_channels=2, sample_rate=48000.0)
    info = E.PlanInfo()
    assert M.lib.mtr_plan_query(C.byref(cfg), 0, 48000, 0, C.byref(info)) == -1
    assert b"struct_size" in M.lib.mtr_last_error()
    cfg.struct_size = C.sizeof(E._Config)
    assert M.lib.mtr_plan_query(C.byref(cfg), 0, 48000, 0, C.byref(info)) == 0
    # the per-call timing getter and the host-path chunk size need an engine
    assert M.lib.mtr_engine_timing_calls(None, None, 0, None) == -1
    assert M.lib.mtr_engine_set_host_chunk_bytes(None, 1 << 20) == -1


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(M.EngineError) as ei:
        M.Engine(4)
    assert "(-3)" in str(ei.value) and "no CPU path" in str(ei.value)


def test_new_entry_points_reject_what_they_must_without_a_device():
    """The round-5 additions of the ABI, as far as they go without a GPU: the state-blob helpers are pure, the communicator
    calls check their arguments before they touch RCCL, and RCCL's version is readable."""
    assert M.lib.mtr_state_blob_count(None, 0) == 0
    assert M.lib.mtr_state_blob_count(b"x" * 200, 200) == 0            # not a blob: no magic
    assert M.lib.mtr_engine_state_bytes(None, 5) == 0
    assert M.lib.mtr_engine_state_export(None, 0, 1, None, 0) == -1
    assert M.lib.mtr_engine_state_import(None, 0, None, 0) == -1
    assert M.lib.mtr_comm_init_timeout(None, 0, 1, None, 0, 1000, None) == -1
    assert M.lib.mtr_comm_probe(None, 1000, None) == -1 and M.lib.mtr_comm_set_timeout(None, 1000) == -1
    assert M.engine.rccl_version() >= 21800
    assert (M.engine.ERR_TIMEOUT, M.engine.ERR_STATE) == (-6, -7)


def test_a_timing_only_build_names_itself_and_is_refused(tmp_path):
    """A library built with -DMTR_TIMING_ONLY_BUILD may carry kernels with a role switched off (tools/: elimination runs — wrong
    results by construction): mtr_version () says so, and the package refuses to load it unless the caller is a timing tool
    (MTR_ALLOW_TIMING_ONLY_BUILD=1).  Built here from the engine's host TU alone, linked with the shipped kernels."""
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    hipcc = "/opt/rocm/bin/hipcc"
    objdir = os.path.join(root, "meters.lv2_amd", "lib", "obj")
    if not (os.path.exists(hipcc) and os.path.isdir(objdir) and shutil.which("g++")):
        pytest.skip("needs hipcc and the built objects (the build container)")
    obj = str(tmp_path / "mtr_engine_timing.o")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O1", "-std=c++20", "-fPIC", "-DMTR_TIMING_ONLY_BUILD", "-I" + os.path.join(root, "include"),
                    "-I" + os.path.join(root, "meters.lv2_amd", "csrc"), "-c", os.path.join(root, "meters.lv2_amd", "csrc", "mtr_engine.hip"), "-o", obj],
                   check=True, capture_output=True, timeout=600)
    so = str(tmp_path / "libmtr_engine.so")
    others = [os.path.join(objdir, f) for f in os.listdir(objdir) if f.endswith(".o") and f != "mtr_engine.o"]
    subprocess.run(["g++", "-shared", "-fPIC", "-o", so, obj] + others + ["-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lamdhip64", "-lrccl", "-lm"],
                   check=True, capture_output=True, timeout=300)
    code = "import sys; sys.path.insert(0, %r); import meters.lv2_amd as M; print(M.lib.mtr_version().decode())" % root
    env = dict(os.environ, MTR_LIB=so)
    env.pop("MTR_ALLOW_TIMING_ONLY_BUILD", None)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "TIMING-ONLY build" in r.stderr, (r.stdout, r.stderr[-800:])
    r = subprocess.run([sys.executable, "-c", code], env=dict(env, MTR_ALLOW_TIMING_ONLY_BUILD="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "TIMING-ONLY BUILD" in r.stdout, (r.stdout, r.stderr[-800:])
