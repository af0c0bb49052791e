cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -x -q > gpurun_out/r5_tests_final.txt 2>&1
tail -4 gpurun_out/r5_tests_final.txt
timeout 600 python tools/micro/patch_repeat.py 20 > gpurun_out/r5_repeat_final.txt 2>&1
tail -11 gpurun_out/r5_repeat_final.txt
