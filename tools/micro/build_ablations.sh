#!/bin/bash
# Compile-time ablations of the patch kernels' K loop on the clock-probe build (workgroup life in shader cycles):
#   bash tools/micro/build_ablations.sh "0 4 1 5 256 512 768 769 773 837 2 66"
# needs the probe variant's objects (bash tools/micro/build_variant.sh clk "-DADVOC_CLOCK_PROBE"); writes
# advoc_amd/csrc/libadvoc_hip_abl<N>.so with igemm_patch.hip compiled under -DADVOC_P3_ABL=<N>.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd); cd $ROOT
BASE=/tmp/advoc_variant_clk
pids=()
for n in $1; do
  mkdir -p /tmp/abl_$n
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Wall -Wno-unused-function -Wno-pass-failed -Wno-unused-variable -Wno-unused-but-set-variable \
      -DADVOC_CLOCK_PROBE -DADVOC_P3_ABL=$n $2 -c advoc_amd/csrc/igemm_patch.hip -o /tmp/abl_$n/igemm_patch.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o advoc_amd/csrc/libadvoc_hip_abl$n.so $(ls $BASE/*.o | grep -v igemm_patch.o) /tmp/abl_$n/igemm_patch.o ) &
  pids+=($!)
  if [ ${#pids[@]} -ge 6 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
ls advoc_amd/csrc/libadvoc_hip_abl*.so
