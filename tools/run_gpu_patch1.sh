cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_conv.py -x -q -m gpu -k "patch" 2>&1 | tail -15
timeout 900 python tools/micro/patch_sweep.py 2>&1 | grep -v amdgpu | tee gpurun_out/patch_sweep1.txt
