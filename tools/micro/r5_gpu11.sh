cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
T=tests/test_hip_fullsize.py::test_full_model_train_loops_at_bench_size_match_the_float64_oracle
for i in 1 2 3; do
timeout 600 python -m pytest $T -x -q -s -m gpu 2>&1 | grep -E "step [12]:|passed|failed|AssertionError|assert " | cut -c1-400 >> gpurun_out/r5k_on.txt
ADVOC_Y_IMAGE_ONLY=0 timeout 600 python -m pytest $T -x -q -s -m gpu 2>&1 | grep -E "step [12]:|passed|failed|AssertionError|assert " | cut -c1-400 >> gpurun_out/r5k_off.txt
done
