cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
L=advoc_amd/csrc/libadvoc_hip
bash tools/micro/lib_ab2.sh $L.so ${L}_tsa2.so ${L}_fla2.so ${L}_twnt.so ${L}_allnt.so $L.so > gpurun_out/r6n_nt.txt 2>&1
grep "ms_per_step\|thin\|fused\|bias_grad\|operand_images" gpurun_out/r6n_nt.txt
