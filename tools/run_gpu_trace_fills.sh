cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/trace_fills; rm -rf $OUT; mkdir -p $OUT
CMD="env ADVOC_WGRAD_STREAM=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --train-only --prof-steps 0"
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1; echo trace rc=$?
T=$(ls $OUT/trace/*/*kernel_trace.csv | head -1)
for s in FillFunctor fillBuffer copyBuffer; do python tools/trace_summary.py $T --skip-first 0 --by-grid "$s" | grep -A40 "by grid\|grid" | head -40; done > $OUT/fills.md
python tools/trace_summary.py $T --skip-first 0 > $OUT/kernel_trace.md
python tools/trace_gaps.py $T | tee $OUT/gaps.txt
rm -rf $OUT/trace
cat $OUT/fills.md | cut -c1-200
