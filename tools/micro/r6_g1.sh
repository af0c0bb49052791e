cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
python bench.py --train-only --no-cpu-baseline --steps 20 > gpurun_out/r6a_bench.json 2> gpurun_out/r6a_bench.err
python tools/layer_times.py regular 64 > gpurun_out/r6a_layers.txt 2>&1
bash tools/micro/ablation_survey.sh "clk abl64 abl128 abl32" > /dev/null 2>&1
cp gpurun_out/r5z_abl_all.txt gpurun_out/r6a_abl.txt
tail -c 600 gpurun_out/r6a_bench.json
