set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=gpurun_out/r5g_trace; rm -rf $OUT; mkdir -p $OUT
CMD="env ADVOC_WGRAD_STREAM=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --train-only --prof-steps 0"
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1; echo trace rc=$?
T=$(ls $OUT/trace/*/*kernel_trace.csv | head -1)
python tools/trace_steady.py $T 3 > gpurun_out/r5g_steady_census.md
rm -rf $OUT
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r5g_all_tests.txt
