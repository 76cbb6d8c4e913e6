#!/bin/bash
# tools/r06_delay_sweep.sh — the experiment behind the deferred tail's two constants (profiles/r06_tail.md): the idle time in
# front of the deferred gate (MTR_TAIL_DELAY_US) x the gate's grid (MTR_TAIL_GATE_GRID: 0 = one workgroup per stream)
for g in 0 512 256; do for d in 0 20 100; do echo "== MTR_TAIL_GATE_GRID=$g MTR_TAIL_DELAY_US=$d"; MTR_TAIL_GATE_GRID=$g MTR_TAIL_DELAY_US=$d python tools/r06_tail_probe.py serial deferred gate_only; done; done
