cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/prof_lws; mkdir -p $OUT
python tools/micro/lws_time.py > $OUT/lws_time.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python tools/micro/lws_time.py > $OUT/trace.log 2>&1; echo trace rc=$?
T=$(ls $OUT/trace/*/*kernel_trace.csv | head -1)
python tools/trace_summary.py $T --skip-first 0 > $OUT/kernel_trace.md
head -14 $OUT/kernel_trace.md; cat $OUT/lws_time.txt | grep clips
rm -rf $OUT/trace
