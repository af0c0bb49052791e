cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_hip_conv.py -x -q -m gpu 2>&1 | tail -3
bash tools/micro/ablation_survey.sh "clknd clk" > /dev/null 2>&1
grep -v "<1,\|e+19\|e+07" gpurun_out/r5z_abl_all.txt | grep "== \|,0,0,.>" > gpurun_out/r6e_direct_cycles.txt
cat gpurun_out/r6e_direct_cycles.txt
