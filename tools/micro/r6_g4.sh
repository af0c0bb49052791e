cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_hip_conv.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r6d_tests.txt
cat gpurun_out/r6d_tests.txt
bash tools/micro/lib_ab2.sh advoc_amd/csrc/libadvoc_hip_nodirect.so advoc_amd/csrc/libadvoc_hip.so advoc_amd/csrc/libadvoc_hip_nodirect.so advoc_amd/csrc/libadvoc_hip.so > gpurun_out/r6d_ab.txt 2>&1
cat gpurun_out/r6d_ab.txt
