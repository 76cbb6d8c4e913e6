#!/bin/bash
# tools/seg_ab.sh [lib dirs...] — same-box comparison of engine builds on the bench shape (layout 7, EBU R128 + true peak and
# true peak alone): kernel milliseconds by HIP events.  Default: the shipped build against the elimination builds.
[ $# -eq 0 ] && set -- lib lib_NOADV lib_NOPROD
for L in "$@"; do
	[ -f meters.lv2_amd/$L/libmtr_engine.so ] || continue
	echo "== $L"
	MTR_ALLOW_TIMING_ONLY_BUILD=1 MTR_LIB=$PWD/meters.lv2_amd/$L/libmtr_engine.so python tools/seg_try.py big 2>&1 | grep -v amdgpu.ids | grep "layout 7\|max"
done
