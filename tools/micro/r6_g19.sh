cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_conv.py -x -q -m gpu -k "test_layer_patch_kernels and p3_d4" 2>&1 | tail -2
ADVOC_HIP_LIB=$PWD/advoc_amd/csrc/libadvoc_hip_hburst.so timeout 600 python -m pytest tests/test_hip_conv.py -x -q -m gpu -k "test_layer_patch_kernels and (p3_d4 or p3_rem_d4)" 2>&1 | tail -2
bash tools/micro/ablation_survey.sh "clk clkhb" > /dev/null 2>&1
grep "== \|<1," gpurun_out/r5z_abl_all.txt > gpurun_out/r6s_halo_burst_cycles.txt
bash tools/micro/lib_ab2.sh advoc_amd/csrc/libadvoc_hip.so advoc_amd/csrc/libadvoc_hip_hburst.so advoc_amd/csrc/libadvoc_hip.so advoc_amd/csrc/libadvoc_hip_hburst.so 2>&1 | grep "ms_per_step\|<1," >> gpurun_out/r6s_halo_burst_cycles.txt
cat gpurun_out/r6s_halo_burst_cycles.txt
