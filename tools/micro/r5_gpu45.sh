cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
L=advoc_amd/csrc/libadvoc_hip
timeout 2400 bash tools/micro/lib_ab2.sh ${L}.so ${L}_h3pm.so ${L}_h3ft.so ${L}_h3pmft.so ${L}.so ${L}_h3pm.so ${L}_h3ft.so ${L}_h3pmft.so > gpurun_out/r5_h3_ab.txt 2>&1
grep "ms_per_step\|gather_gemm" gpurun_out/r5_h3_ab.txt | cut -c1-220
