cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_conv.py -x -q -m gpu -k "sole_reader or a_priori or weight_magnitude or persistent" 2>&1 | tail -25 > gpurun_out/r5l_newtest.txt
T=tests/test_hip_fullsize.py::test_full_model_train_loops_at_bench_size_match_the_float64_oracle
for i in 1 2 3 4; do
timeout 600 python -m pytest $T -x -q -s -m gpu 2>&1 | grep -E "step [12]:|passed|failed|AssertionError" | cut -c1-300 >> gpurun_out/r5l_full.txt
done
timeout 1500 python -m pytest tests/test_hip_model.py tests/test_hip_fullsize.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r5l_model_tests.txt
