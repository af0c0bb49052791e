cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
bash tools/micro/ablation_survey.sh "clk abl8192" > /dev/null 2>&1
cp gpurun_out/r5z_abl_all.txt gpurun_out/r6b_abl.txt
timeout 900 python -m pytest tests/test_hip_conv.py -x -q -m gpu 2>&1 | tail -3
