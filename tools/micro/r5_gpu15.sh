cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_conv.py -x -q -m gpu -k "sole_reader or a_priori or patch_kernels or operand_image_kernels" 2>&1 | tail -25 > gpurun_out/r5o_newtest.txt
timeout 1800 python -m pytest tests/test_hip_model.py tests/test_hip_fullsize.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r5o_model_tests.txt
L=advoc_amd/csrc/libadvoc_hip
timeout 900 bash tools/micro/lib_ab2.sh ${L}_base.so ${L}.so ${L}_base.so ${L}.so > gpurun_out/r5o_ab.txt 2>&1
