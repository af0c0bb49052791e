#!/bin/bash
# Compile-time ablations of igemm_patch.hip on the PRODUCT objects (no clock probe): wall-time A/B per layer (tools/layer_times.py)
#   bash tools/micro/build_abl_plain.sh "16 8 4"   -> advoc_amd/csrc/libadvoc_hip_pabl<N>.so
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd); cd $ROOT
pids=()
for n in $1; do
  mkdir -p /tmp/pabl_$n
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Wall -Wno-unused-function -Wno-pass-failed -Wno-unused-variable -Wno-unused-but-set-variable -Wno-unused-value \
      -DADVOC_P3_ABL=$n $2 -c advoc_amd/csrc/igemm_patch.hip -o /tmp/pabl_$n/igemm_patch.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o advoc_amd/csrc/libadvoc_hip_pabl$n.so $(ls advoc_amd/csrc/*.o | grep -v igemm_patch.o) /tmp/pabl_$n/igemm_patch.o ) &
  pids+=($!)
  if [ ${#pids[@]} -ge 6 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
ls advoc_amd/csrc/libadvoc_hip_pabl*.so
