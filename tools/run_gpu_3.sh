cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_conv.py -x -q -m gpu -k "operand_image or split_bf16" 2>&1 | tail -8
python tools/micro/x6d_sweep.py 2>&1 | tee gpurun_out/x6d_sweep2.txt
