cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r03
timeout 900 python bench.py > gpurun_out/prof_r03/bench.json 2> gpurun_out/prof_r03/bench.err; echo bench rc=$?; head -c 1500 gpurun_out/prof_r03/bench.json
ADVOC_BENCH_VERBOSE=1 timeout 600 python bench.py --train-only --no-cpu-baseline --steps 10 > /dev/null 2> gpurun_out/prof_r03/layer_times.txt
bash tools/run_gpu_prof.sh
