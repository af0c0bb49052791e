cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/pmc2
run() {  # tag, counters, env...
  tag=$1; shift
  ctr=$1; shift
  env "$@" rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/pmc2/$tag -- python tools/micro/h3_one.py d4 3 f > /dev/null 2>&1
  f=$(ls gpurun_out/pmc2/$tag/*/*counter_collection.csv 2>/dev/null | head -1)
  python - "$f" "$tag" <<'PY'
import csv, sys, collections
f, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
  k = r['Kernel_Name']
  if 'gather_gemm' not in k: continue
  agg['gg'][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
  print(tag, ' '.join('%s=%.4g' % (c, sum(v) / len(v)) for c, v in sorted(d.items())))
PY
}
for v in "h3k1 ADVOC_H3=1" "h3k0 ADVOC_H3=1 ADVOC_IGEMM_KORDER=0"; do
  set -- $v; tag=$1; shift
  run ${tag}_sq1 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "$@"
  run ${tag}_sq2 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "$@"
  run ${tag}_tcp "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" "$@"
  run ${tag}_tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "$@"
  run ${tag}_tcc2 "TCC_EA0_RDREQ_sum TCC_BUSY_avr TCC_TAG_STALL_sum" "$@"
  run ${tag}_ta "TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "$@"
done 2>&1 | tee gpurun_out/pmc2/summary.txt
grep -E "^TA_|^TCP_|^TCC_" gpurun_out/pmc/counters.txt | tr '\n' ' ' | head -c 6000
