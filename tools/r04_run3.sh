#!/bin/bash
# round 4, GPU call 3: k_tpb variants (prefetch across the barrier, pair maps, role eliminations, map splits), parity of the pair-map build
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
for v in v1 pm pm_s1 pm_s2 pm_s3 pm_nochain pm_noprod pm_nomaps pm_nofetch pm_chainonly pm_prodonly pm_mapsonly; do
  echo "=== tpb_prof_$v"; timeout 120 ./tools/tpb_prof_$v 8192 96000 2>&1 | grep -v amdgpu.ids
done > $O/tpb_variants3.txt 2>&1
cat $O/tpb_variants3.txt
timeout 900 python -m pytest tests/test_gpu_hostpath.py tests/test_gpu_parity.py -m gpu -q -k "host or ballistics" > $O/gputests3.txt 2>&1; tail -5 $O/gputests3.txt
MTR_LIB=$PWD/meters.lv2_amd/lib_pm/libmtr_engine.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_lv2_plugin.py tests/test_lv2_dr14.py -m gpu -q -k "ballistics or dBTP or dr14 or TPnRMS" > $O/gputests3_pm.txt 2>&1; tail -5 $O/gputests3_pm.txt
MTR_LIB=$PWD/meters.lv2_amd/lib_pm/libmtr_engine.so timeout 900 python tools/fuzz_tpb.py 0 400 > $O/fuzz_tpb3_pm.txt 2>&1; tail -3 $O/fuzz_tpb3_pm.txt
timeout 300 python tools/bank_mono_probe.py > $O/bank_mono3.txt 2>&1; cat $O/bank_mono3.txt
