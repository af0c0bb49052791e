cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/prof_lds; rm -rf $OUT; mkdir -p $OUT
CMD="env ADVOC_WGRAD_STREAM=0 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --train-only --prof-steps 0"
timeout 400 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/lds -- $CMD > $OUT/lds.log 2>&1; echo lds rc=$?
S=$(ls $OUT/lds/*/*counter_collection.csv | head -1)
python tools/lds_summary.py $S > $OUT/pmc_lds.md
head -24 $OUT/pmc_lds.md
rm -rf $OUT/lds
