cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/small
for P in 0 1 0 1; do
  ADVOC_H3_DEEP_PLAN=$P python bench.py --model small --train-only --no-cpu-baseline --steps 100 2>/tmp/err.txt | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('PLAN=$P small ms_per_step %.3f' % r['ms_per_step'])" || tail -3 /tmp/err.txt
done 2>&1 | tee gpurun_out/small/plan_ab.txt
timeout 2300 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | tee gpurun_out/small/gpu_tests.txt
