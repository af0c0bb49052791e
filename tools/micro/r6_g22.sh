cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_conv.py tests/test_hip_model.py -x -q -m gpu -k "thin or edge or layer_1 or encoder_1 or forward_losses or two_stage or direct or small_k or first or sole_reader or all_directions" 2>&1 | tail -3
L=advoc_amd/csrc/libadvoc_hip; bash tools/micro/lib_ab2.sh ${L}_base.so $L.so ${L}_base.so $L.so 2>&1 | grep "ms_per_step\|thin_k\|thin_wgrad" > gpurun_out/r6v_thin_fetch.txt; cat gpurun_out/r6v_thin_fetch.txt
