set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
L=advoc_amd/csrc/libadvoc_hip
timeout 1200 bash tools/micro/lib_ab2.sh ${L}.so ${L}_dblb.so ${L}_iglp0.so ${L}_iglp1.so ${L}.so ${L}_dblb.so ${L}_iglp0.so ${L}_iglp1.so > gpurun_out/r5h_ab.txt 2>&1
