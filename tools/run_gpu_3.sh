cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_hip_conv.py -x -q -m gpu -k "weight_gradient or split_bf16" 2>&1 | tail -12
