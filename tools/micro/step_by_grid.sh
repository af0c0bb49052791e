# kernel-only times of one kernel family inside the train step, by grid size:  bash tools/micro/step_by_grid.sh 'wgrad_h3'
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/bygrid; mkdir -p $OUT; rm -rf $OUT/t1
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/t1 -- env ADVOC_WGRAD_STREAM=0 ADVOC_WGRAD_LOG=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --train-only --prof-steps 0 > $OUT/t1.log 2>&1
for K in "$@"; do
  python tools/trace_summary.py $(ls $OUT/t1/*/*kernel_trace.csv | head -1) --skip-first 0 --by-grid "$K" | sed -n '/| kernel | grid/,$p' | tee $OUT/by_grid_$(echo $K | tr -c 'a-zA-Z0-9_\n' '_').md
done
rm -rf $OUT/t1
