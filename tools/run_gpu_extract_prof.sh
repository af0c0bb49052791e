cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/prof_extract; rm -rf $OUT; mkdir -p $OUT
CMD="python tools/micro/extract_time.py"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- $CMD > $OUT/sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq2 -- $CMD > $OUT/sq2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT/sq3 -- $CMD > $OUT/sq3.log 2>&1
python tools/sq_summary.py $(ls $OUT/sq/*/*counter_collection.csv | head -1) > $OUT/sq.md; cat $OUT/sq.md
python - <<'PY'
import csv, glob, collections
for d in ('sq2', 'sq3'):
  f = glob.glob('gpurun_out/prof_extract/%s/*/*counter_collection.csv' % d)[0]
  agg = collections.defaultdict(lambda: collections.defaultdict(list))
  for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'][:40]
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
  for k, dd in agg.items():
    if 'stft' in k or 'mel_pinv' in k:
      print(d, k, {c: round(sum(v)/len(v)) for c, v in dd.items()}, len(next(iter(dd.values()))))
PY
rm -rf $OUT/sq $OUT/sq2 $OUT/sq3
