#!/bin/bash
# tools/build_ab.sh [rev] — build the engine of a committed revision (default HEAD) into meters.lv2_amd/lib_ab, the "B" of tools/ab.sh
rev=${1:-HEAD}
top=$(git rev-parse --show-toplevel)
tmp=$(mktemp -d)
git -C "$top" archive "$rev" meters.lv2_amd/csrc include tools | tar -x -C "$tmp"
rm -rf "$top/meters.lv2_amd/lib_ab"
make -s -C "$tmp/meters.lv2_amd/csrc" OUT="$top/meters.lv2_amd/lib_ab" "$top/meters.lv2_amd/lib_ab/libmtr_engine.so" && echo "lib_ab = $rev"
rm -rf "$tmp"
