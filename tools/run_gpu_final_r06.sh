# The round-end call: whole GPU suite, bitwise repeatability of the patch kernels, then the evidence set (run_gpu_prof_r06.sh final)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -x -q > gpurun_out/r6_tests_final.txt 2>&1
tail -4 gpurun_out/r6_tests_final.txt
timeout 600 python tools/micro/patch_repeat.py 20 > gpurun_out/r6_repeat_final.txt 2>&1
tail -12 gpurun_out/r6_repeat_final.txt
bash tools/run_gpu_prof_r06.sh final > gpurun_out/r6_prof_final.log 2>&1
tail -30 gpurun_out/r6_prof_final.log
