"""meters.lv2_amd — MI355X-native batch engine for the meters.lv2 DSP hot path.

The product is the C-ABI shared library ``lib/libmtr_engine.so`` (HIP kernels for gfx950 + host
code, built from ``csrc/`` by ``__graft_entry__.build()``) and the LV2 plugin ``lib/meters_amd.so``
on top of it.  This Python package is only the thin ctypes binding the tests and bench.py drive
the library through; there is no Python or CPU compute path here, and importing it fails loudly
when the library has not been built.
"""
from .engine import (  # noqa: F401
    Engine, EngineError, StreamResult, lib, lib_path, Comm, comm_unique_id,
    METER_EBU, METER_TRUEPEAK, METER_SPECTR30, METER_TPBALLIST, METER_BITSTATS, METER_SIGDIST, METER_DR14, METER_KMETER,
    fir_table, kweight_coef, band_coef, hist_loudness, synth_fill_device, exported_symbols, plan_query,
)
