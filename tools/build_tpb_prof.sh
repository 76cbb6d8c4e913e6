#!/bin/bash
# tools/build_tpb_prof.sh — tools/tpb_prof (cycles per wave and role of k_tpb) and its elimination builds (a role switched off:
# timing only, wrong results — they carry MTR_TIMING_ONLY_BUILD)
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++20 -Iinclude -Imeters.lv2_amd/csrc -mllvm -amdgpu-mfma-vgpr-form"
/opt/rocm/bin/hipcc $F tools/tpb_prof.hip -o tools/tpb_prof &
for v in NOCHAIN NOPROD NOSPLIT NODMA; do
	/opt/rocm/bin/hipcc $F -DMTR_TIMING_ONLY_BUILD -DMTR_TPB_DBG_$v=1 tools/tpb_prof.hip -o tools/tpb_prof_$(echo $v | tr A-Z a-z) &
done
wait
ls -la tools/tpb_prof*
