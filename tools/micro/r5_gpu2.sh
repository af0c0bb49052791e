set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=tests/test_hip_fullsize.py::test_full_model_train_loops_at_bench_size_match_the_float64_oracle
timeout 600 python -m pytest $T -x -q -m gpu 2>&1 | tail -40 > gpurun_out/r5b_test_new.txt
ADVOC_HIP_LIB=$PWD/advoc_amd/csrc/libadvoc_hip_base.so timeout 600 python -m pytest $T -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r5b_test_base.txt
L=advoc_amd/csrc/libadvoc_hip
timeout 900 bash tools/micro/lib_ab2.sh ${L}_base.so ${L}_pinnone.so ${L}.so ${L}_pinall.so ${L}_base.so ${L}_pinnone.so ${L}.so ${L}_pinall.so > gpurun_out/r5b_ab.txt 2>&1
