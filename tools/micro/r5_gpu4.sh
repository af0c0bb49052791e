set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
L=advoc_amd/csrc/libadvoc_hip
timeout 900 python tools/micro/patch_repeat.py 40 > gpurun_out/r5d_repeat.txt 2>&1
timeout 900 bash tools/micro/lib_ab2.sh ${L}.so ${L}_setprio.so ${L}.so ${L}_setprio.so > gpurun_out/r5d_ab.txt 2>&1
timeout 1500 python -m pytest tests/test_hip_bench.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r5d_bench_tests.txt
OUT=gpurun_out/r5d_trace; rm -rf $OUT; mkdir -p $OUT
CMD="env ADVOC_WGRAD_STREAM=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --train-only --prof-steps 0"
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1; echo trace rc=$?
T=$(ls $OUT/trace/*/*kernel_trace.csv | head -1)
python tools/trace_steady.py $T 3 > gpurun_out/r5d_steady_census.md
rm -rf $OUT
