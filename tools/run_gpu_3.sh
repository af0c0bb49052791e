cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_hip_conv.py tests/test_hip_model.py tests/test_hip_parallel.py -x -q -m gpu 2>&1 | tail -8
ADVOC_BENCH_VERBOSE=1 timeout 600 python bench.py --steps 10 --warmup 2 --train-only --no-cpu-baseline > gpurun_out/b6_h3.json 2> gpurun_out/b6_h3.err; head -12 gpurun_out/b6_h3.err; tail -2 gpurun_out/b6_h3.err; head -c 300 gpurun_out/b6_h3.json
