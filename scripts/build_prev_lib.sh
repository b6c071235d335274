#!/bin/bash
# Builds the library of another revision beside the working tree's, for scripts/ab_step.sh:
#   bash scripts/build_prev_lib.sh [rev=HEAD]  ->  daliid_amd/libdaliid_prev.so   (git-ignored; remove it before the final GPU call)
REV=${1:-HEAD}
T=/tmp/daliid_prev_build; rm -rf $T; mkdir -p $T
git archive $REV daliid_amd/csrc include | tar -x -C $T || exit 1
make -C $T/daliid_amd/csrc -j8 > $T/build.log 2>&1 || { tail -5 $T/build.log; exit 1; }
cp $T/daliid_amd/libdaliid_hip.so daliid_amd/libdaliid_prev.so && echo "built daliid_amd/libdaliid_prev.so from $REV"
