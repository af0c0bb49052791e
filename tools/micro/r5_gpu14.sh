cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r5n_all_tests.txt
bash tools/run_gpu_prof_r05.sh a > gpurun_out/r5n_prof.log 2>&1
