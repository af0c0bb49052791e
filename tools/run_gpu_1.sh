set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_conv.py -x -q -m gpu 2>&1 | tail -25
ADVOC_BENCH_VERBOSE=1 timeout 600 python bench.py --steps 10 --warmup 2 --train-only --no-cpu-baseline > gpurun_out/b1_x6d.json 2> gpurun_out/b1_x6d.err; tail -3 gpurun_out/b1_x6d.err; cat gpurun_out/b1_x6d.json | head -c 600
ADVOC_X6D=0 ADVOC_BENCH_VERBOSE=1 timeout 600 python bench.py --steps 10 --warmup 2 --train-only --no-cpu-baseline > gpurun_out/b1_old.json 2> gpurun_out/b1_old.err; tail -3 gpurun_out/b1_old.err; cat gpurun_out/b1_old.json | head -c 600
