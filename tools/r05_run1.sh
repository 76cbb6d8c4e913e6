#!/bin/bash
# round 5, GPU call 1: the new k_tpb (correctness first), its roles and counters against round 4's, then the whole suite
O=gpurun_out/r05a; mkdir -p $O; export TMPDIR=/tmp
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids"
echo "== ballistics tests"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_seg.py tests/test_gpu_hostpath.py tests/test_lv2_plugin.py -m gpu -q --maxfail=20 -k "ballistics or tpb or dBTP or golden_other or hostpath" > $O/t_tpb.txt 2>&1; grep -v "$F" $O/t_tpb.txt | tail -25
echo "== roles"; for v in "" _map0 _r4 _nochain _noprod _nosplit _nodma; do echo "=== tpb_prof$v"; timeout 120 ./tools/tpb_prof$v 8192 96000 2>&1 | grep -v amdgpu.ids; done > $O/tpb_roles.txt 2>&1; grep -A16 "=== tpb_prof$" $O/tpb_roles.txt; grep "ns per chunk\|===" $O/tpb_roles.txt
echo "== A/B"; bash tools/tpb_ab.sh lib lib_map0 lib_r4 2>&1 | tee $O/tpb_ab.txt
echo "== fuzz"; timeout 500 python tools/fuzz_tpb.py 9000 300 2>&1 | grep -v "$F" | tee $O/fuzz_tpb.txt | tail -6
echo "== suite"; timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 > $O/gputests.txt 2>&1; echo "pytest rc $?" >> $O/gputests.txt; grep -v "$F" $O/gputests.txt | tail -60
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v "$F" | tail -3 | tee $O/smoke.txt
echo "== pmc"; bash tools/prof_seg.sh r05a_tpb --meters tpb --steps 4 > /dev/null 2>&1; cp gpurun_out/prof_r05a_tpb/summary.txt $O/r05a_tpb_pmc.txt; grep "k_tpb" $O/r05a_tpb_pmc.txt | head -40
find gpurun_out/prof_r05a* -name "*.csv" -size +2M -delete
