cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
L=advoc_amd/csrc/libadvoc_hip
for v in stg2 stg2all; do
  ADVOC_HIP_LIB=$PWD/${L}_$v.so timeout 900 python -m pytest tests/test_hip_conv.py -m gpu -x -q -k "patch_kernels or remainder_columns or a_priori or sole_reader or producer_written" > gpurun_out/r5u_tests_$v.txt 2>&1
  tail -2 gpurun_out/r5u_tests_$v.txt
done
timeout 1500 bash tools/micro/lib_ab2.sh ${L}.so ${L}_stg2.so ${L}_stg2all.so ${L}.so ${L}_stg2.so ${L}_stg2all.so > gpurun_out/r5u_ab.txt 2>&1
head -12 gpurun_out/r5u_ab.txt | cut -c1-200
timeout 300 python -m pytest tests/test_hip_conv.py -m gpu -x -q -k "two_streams_share or k_slices_summed" 2>&1 | tail -2
