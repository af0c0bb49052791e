cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_hip_conv.py -m gpu -x -q -k "row_mode or weight_gradient or k_slices or two_streams" 2>&1 | tail -5
timeout 1800 python -m pytest tests/test_hip_fullsize.py tests/test_hip_model.py tests/test_hip_bench.py -m gpu -x -q 2>&1 | tail -5
