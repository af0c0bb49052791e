set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=advoc_amd/csrc/libadvoc_hip
timeout 1200 bash tools/micro/lib_ab2.sh ${L}.so ${L}_nosb.so ${L}_pindma.so ${L}_nostag.so ${L}.so ${L}_nosb.so ${L}_pindma.so ${L}_nostag.so > gpurun_out/r5c_ab.txt 2>&1
T=tests/test_hip_fullsize.py::test_full_model_train_loops_at_bench_size_match_the_float64_oracle
for i in 1 2 3; do
timeout 600 python -m pytest $T -x -q -s -m gpu 2>&1 | grep -E "step [12]:|passed|failed|Assertion" >> gpurun_out/r5c_flaky_new.txt
ADVOC_HIP_LIB=$PWD/${L}_base.so timeout 600 python -m pytest $T -x -q -s -m gpu 2>&1 | grep -E "step [12]:|passed|failed|Assertion" >> gpurun_out/r5c_flaky_base.txt
done
