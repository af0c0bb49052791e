cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
bash tools/micro/lib_ab2.sh advoc_amd/csrc/libadvoc_hip.so advoc_amd/csrc/libadvoc_hip_aux2.so advoc_amd/csrc/libadvoc_hip_aux16.so advoc_amd/csrc/libadvoc_hip_aux18.so advoc_amd/csrc/libadvoc_hip.so > gpurun_out/r6m_store_aux.txt 2>&1
head -14 gpurun_out/r6m_store_aux.txt
