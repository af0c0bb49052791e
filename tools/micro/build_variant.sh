#!/bin/bash
# A second build of the library for same-box A/B runs (tools/micro/lib_ab2.sh):
#   bash tools/micro/build_variant.sh NAME "EXTRA_CXXFLAGS" [SRCDIR]
# compiles SRCDIR/*.hip (default advoc_amd/csrc; a `git worktree` of another commit works too) with the Makefile's flags plus
# EXTRA_CXXFLAGS into /tmp/advoc_variant_NAME/ and links advoc_amd/csrc/libadvoc_hip_NAME.so (git-ignored, travels with gpurun).
set -e
NAME=$1; EXTRA=$2; SRC=${3:-advoc_amd/csrc}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OBJ=/tmp/advoc_variant_$NAME
mkdir -p $OBJ
INC=$(cd $SRC/../../include && pwd)
pids=()
for f in $SRC/*.hip; do
  o=$OBJ/$(basename ${f%.hip}).o
  if [ ! -f $o ] || [ $f -nt $o ] || [ -n "$FORCE" ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$INC -Wall -Wno-unused-function -Wno-pass-failed $EXTRA -c $f -o $o &
    pids+=($!)
    if [ ${#pids[@]} -ge 8 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/advoc_amd/csrc/libadvoc_hip_$NAME.so $OBJ/*.o
ls -la $ROOT/advoc_amd/csrc/libadvoc_hip_$NAME.so
