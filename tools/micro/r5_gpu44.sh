cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_conv.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python tools/micro/patch_repeat.py 20 2>&1 | tail -3
L=advoc_amd/csrc/libadvoc_hip
timeout 1200 bash tools/micro/lib_ab2.sh ${L}_steploop.so ${L}.so ${L}_steploop.so ${L}.so > gpurun_out/r5_slice_ab.txt 2>&1
head -8 gpurun_out/r5_slice_ab.txt | cut -c1-160
