#!/bin/bash
# Builds an experiment variant of the library into tools/_exp/lib_<tag>.so:
#   tools/exp_build.sh <tag> [extra hipcc flags, e.g. -DSMLVM_FWD_WPE16=6]
# (tools/_exp is git-ignored; the .so files travel to the GPU box with gpurun.)
set -e
TAG=$1; shift
SRC=sparse-marginalization-lvm_amd/csrc
OUT=tools/_exp/obj_$TAG
mkdir -p $OUT
for f in capi sparsemax compact binary_topk sparsemap lowrank_ce segmented; do
    extra=""
    [ $f = sparsemap ] && extra="-ffp-contract=off"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function $extra "$@" -c $SRC/$f.hip -o $OUT/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_exp/lib_$TAG.so $OUT/*.o
rm -rf $OUT
echo built tools/_exp/lib_$TAG.so
