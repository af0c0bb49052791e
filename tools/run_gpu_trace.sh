cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/trace_now; rm -rf $OUT; mkdir -p $OUT
CMD="env ADVOC_WGRAD_STREAM=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --train-only --prof-steps 0"
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1; echo trace rc=$?
T=$(ls $OUT/trace/*/*kernel_trace.csv | head -1)
python tools/trace_summary.py $T --skip-first 0 --by-grid "gather_gemm_h3_kernel<2, 1, 2, 2>" > $OUT/kernel_trace.md; head -3 $T
rm -rf $OUT/trace
tail -45 $OUT/kernel_trace.md | cut -c1-150
