set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_fullsize.py -x -q -s -m gpu -k "batch_norm" 2>&1 | grep -v "^$" | tail -30 > gpurun_out/r5i_bn_test.txt
L=advoc_amd/csrc/libadvoc_hip
timeout 900 bash tools/micro/lib_ab2.sh ${L}_base.so ${L}.so ${L}_base.so ${L}.so > gpurun_out/r5i_ab.txt 2>&1
