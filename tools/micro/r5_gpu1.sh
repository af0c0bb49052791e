set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_conv.py tests/test_hip_numerics.py tests/test_hip_fullsize.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r5a_tests.txt
cat gpurun_out/r5a_tests.txt
timeout 900 bash tools/micro/lib_ab2.sh advoc_amd/csrc/libadvoc_hip_base.so advoc_amd/csrc/libadvoc_hip_late.so advoc_amd/csrc/libadvoc_hip.so advoc_amd/csrc/libadvoc_hip_base.so advoc_amd/csrc/libadvoc_hip_late.so advoc_amd/csrc/libadvoc_hip.so > gpurun_out/r5a_ab.txt 2>&1
cat gpurun_out/r5a_ab.txt
