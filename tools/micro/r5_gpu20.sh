cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
L=advoc_amd/csrc/libadvoc_hip
timeout 1200 bash tools/micro/lib_ab2.sh ${L}.so ${L}_h3iglp0.so ${L}.so ${L}_h3iglp0.so > gpurun_out/r5t_ab.txt 2>&1
