#!/bin/bash
# As build_ablations.sh for the image weight gradient: wgrad_h3.hip under -DADVOC_WH3_ABL=<n> on the clock-probe objects.
#   bash tools/micro/build_ablations_w.sh "1 4 32" [extra flags]
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd); cd $ROOT
BASE=/tmp/advoc_variant_clk
for n in $1; do
  mkdir -p /tmp/wabl_$n
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Wno-pass-failed -Wno-c++20-extensions -Wno-unused-variable -Wno-unused-but-set-variable \
      -DADVOC_CLOCK_PROBE -DADVOC_WH3_ABL=$n $2 -c advoc_amd/csrc/wgrad_h3.hip -o /tmp/wabl_$n/wgrad_h3.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o advoc_amd/csrc/libadvoc_hip_wabl$n.so $(ls $BASE/*.o | grep -v wgrad_h3.o) /tmp/wabl_$n/wgrad_h3.o ) &
done
wait
ls advoc_amd/csrc/libadvoc_hip_wabl*.so
