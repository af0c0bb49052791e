cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_conv.py -x -q -m gpu -k "weight_magnitude or persistent or a_priori or sole_reader" 2>&1 | tail -5 > gpurun_out/r5q_tests.txt
OUT=gpurun_out/r5q_trace; rm -rf $OUT; mkdir -p $OUT
CMD="env ADVOC_WGRAD_STREAM=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --train-only --prof-steps 0"
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
T=$(ls $OUT/trace/*/*kernel_trace.csv | head -1)
python tools/trace_steady.py $T 3 > gpurun_out/r5q_steady_census.md
rm -rf $OUT
for i in 1 2; do python bench.py --train-only --no-cpu-baseline --steps 40 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('default', r['ms_per_step'])" >> gpurun_out/r5q_ab.txt; done
