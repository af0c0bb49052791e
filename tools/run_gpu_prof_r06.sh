# Round-5 evidence in one call (gpurun): the default bench line, per-layer times, kernel trace, matrix-pipe and HBM counter
# passes (each in its own rocprofv3 run), the LWS legs.  Summaries only are kept; copy them to profiles/r06_<tag>_*.
#   bash tools/run_gpu_prof_r06.sh [tag]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-a}
OUT=gpurun_out/prof_r06; rm -rf $OUT; mkdir -p $OUT
timeout 900 python bench.py > $OUT/r06_${TAG}_bench.json 2> $OUT/bench.err; echo bench rc=$?; head -c 600 $OUT/r06_${TAG}_bench.json; echo
ADVOC_BENCH_VERBOSE=1 timeout 600 python bench.py --train-only --no-cpu-baseline --steps 10 > /dev/null 2> $OUT/r06_${TAG}_layer_times.txt
CMD="env ADVOC_WGRAD_STREAM=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --train-only --prof-steps 0"
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1; echo trace rc=$?
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1; echo fetch rc=$?
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1; echo write rc=$?
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- $CMD > $OUT/sq.log 2>&1; echo sq rc=$?
T=$(ls $OUT/trace/*/*kernel_trace.csv | head -1); F=$(ls $OUT/fetch/*/*counter_collection.csv | head -1); W=$(ls $OUT/write/*/*counter_collection.csv | head -1); S=$(ls $OUT/sq/*/*counter_collection.csv | head -1)
python tools/trace_summary.py $T --skip-first 0 > $OUT/r06_${TAG}_kernel_trace.md
python tools/trace_gaps.py $T > $OUT/r06_${TAG}_launch_gaps.txt
python tools/trace_steady.py $T 2 $OUT/r06_${TAG}_step_sequence.txt > $OUT/r06_${TAG}_steady_census.md
python tools/pmc_summary.py $F $W --json $OUT/r06_traffic.json > $OUT/r06_${TAG}_pmc_hbm.md
python tools/sq_summary.py $S > $OUT/r06_${TAG}_pmc_mfma.md
rm -rf $OUT/trace $OUT/fetch $OUT/write $OUT/sq
# shader clock inside the real step (probe build: tools/micro/build_variant.sh clk "-DADVOC_CLOCK_PROBE")
if [ -f advoc_amd/csrc/libadvoc_hip_clk.so ]; then
  ADVOC_HIP_LIB=$PWD/advoc_amd/csrc/libadvoc_hip_clk.so timeout 600 python bench.py --train-only --no-cpu-baseline --steps 4 --warmup 3 --prof-steps 0 > $OUT/clk.log 2>&1
  python tools/clock_summary.py $OUT/clk.log > $OUT/r06_${TAG}_shader_clock.md; rm -f $OUT/clk.log
fi
python tools/micro/lws_time.py > $OUT/r06_${TAG}_lws_time.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/ltrace -- python tools/micro/lws_time.py > $OUT/ltrace.log 2>&1; echo lws trace rc=$?
T=$(ls $OUT/ltrace/*/*kernel_trace.csv | head -1)
python tools/trace_summary.py $T --skip-first 0 > $OUT/r06_${TAG}_lws_kernel_trace.md
rm -rf $OUT/ltrace
head -16 $OUT/r06_${TAG}_kernel_trace.md | cut -c1-140; head -8 $OUT/r06_${TAG}_pmc_hbm.md | cut -c1-200; head -8 $OUT/r06_${TAG}_pmc_mfma.md | cut -c1-200; cat $OUT/r06_${TAG}_launch_gaps.txt; grep clips $OUT/r06_${TAG}_lws_time.txt
