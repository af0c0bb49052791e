# kernel-only times of one kernel family inside the AdVoc-small train step, by grid size, under environment settings:
#   bash tools/micro/small_by_grid.sh 'gather_gemm_h3_kernel<2, 1' "" "ADVOC_EMIT_IMAGES=0"
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/bygrid; mkdir -p $OUT
K="$1"; shift
for setting in "$@"; do
  rm -rf $OUT/t1
  env ADVOC_WGRAD_STREAM=0 $setting timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/t1 -- python bench.py --model small --steps 3 --warmup 1 --no-cpu-baseline --train-only --prof-steps 0 > $OUT/t1.log 2>&1
  echo "== ${setting:-(default)}"
  python tools/trace_summary.py $(ls $OUT/t1/*/*kernel_trace.csv | head -1) --skip-first 0 --by-grid "$K" | sed -n '/| kernel | grid/,$p'
done 2>&1 | tee $OUT/small_by_grid.txt
rm -rf $OUT/t1
