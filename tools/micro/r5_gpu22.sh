cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_conv.py -m gpu -x -q -k "tile_choice_follows or two_streams_share or fused_taps or thin or all_directions" > gpurun_out/r5_tests_new.txt 2>&1
tail -4 gpurun_out/r5_tests_new.txt
bash tools/run_gpu_prof_r05.sh b > gpurun_out/r5_prof_b.log 2>&1
tail -30 gpurun_out/r5_prof_b.log
