cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_conv.py -x -q -m gpu -k "test_layer_patch_kernels" 2>&1 | tail -15 > gpurun_out/r6f_tests.txt
cat gpurun_out/r6f_tests.txt
ROWS=12 bash tools/micro/env_ab2.sh "ADVOC_H3_PATCH_2WG=0" "ADVOC_H3_PATCH_2WG=1" "ADVOC_H3_PATCH_2WG=2" "ADVOC_H3_PATCH_2WG=1 ADVOC_H3_PATCH_2WG_DELAY=0" "ADVOC_H3_PATCH_2WG=1 ADVOC_H3_PATCH_2WG_DELAY=50" "ADVOC_H3_PATCH_2WG=0" > gpurun_out/r6f_ab.txt 2>&1
cat gpurun_out/r6f_ab.txt
