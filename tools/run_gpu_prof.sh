cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/prof_r03; mkdir -p $OUT
CMD="env ADVOC_WGRAD_STREAM=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --train-only --prof-steps 0"
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1; echo trace rc=$?
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1; echo fetch rc=$?
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1; echo write rc=$?
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- $CMD > $OUT/sq.log 2>&1; echo sq rc=$?
T=$(ls $OUT/trace/*/*kernel_trace.csv | head -1); F=$(ls $OUT/fetch/*/*counter_collection.csv | head -1); W=$(ls $OUT/write/*/*counter_collection.csv | head -1); S=$(ls $OUT/sq/*/*counter_collection.csv | head -1)
python tools/trace_summary.py $T --skip-first 0 > $OUT/kernel_trace.md
python tools/pmc_summary.py $F $W --json $OUT/traffic.json > $OUT/pmc_hbm.md
python tools/sq_summary.py $S > $OUT/pmc_mfma.md
head -20 $OUT/kernel_trace.md; head -12 $OUT/pmc_hbm.md; head -12 $OUT/pmc_mfma.md
# keep only the summaries (raw csv is large)
rm -rf $OUT/trace $OUT/fetch $OUT/write $OUT/sq
