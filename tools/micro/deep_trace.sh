# kernel-only times of the deep-layer launches: (1) in the train step, by grid size, (2) under forced K splits
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/deep; mkdir -p $OUT; rm -rf $OUT/t1 $OUT/t2 $OUT/split_kernel_only.txt
if [ "$1" != "nostep" ]; then
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/t1 -- env ADVOC_WGRAD_STREAM=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --train-only --prof-steps 0 > $OUT/t1.log 2>&1
python tools/trace_summary.py $(ls $OUT/t1/*/*kernel_trace.csv | head -1) --skip-first 0 --by-grid 'gather_gemm_h3_kernel<2, 1' | sed -n '/| kernel | grid/,$p' > $OUT/step_by_grid.md; cat $OUT/step_by_grid.md
fi
for TILE in ${TILES:-0}; do
rm -rf $OUT/t2
TILE=$TILE timeout 700 rocprofv3 --kernel-trace --output-format csv -d $OUT/t2 -- python tools/micro/deep_split_trace.py run $OUT/cfg.txt > $OUT/t2.log 2>&1
echo "ADVOC_H3_TILE=$TILE" | tee -a $OUT/split_kernel_only.txt
python tools/micro/deep_split_trace.py table $OUT/cfg.txt $(ls $OUT/t2/*/*kernel_trace.csv | head -1) | tee -a $OUT/split_kernel_only.txt
done
rm -rf $OUT/t1 $OUT/t2
