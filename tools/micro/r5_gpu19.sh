cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
for i in 1 2 3; do
ADVOC_DX_ACCUM=0 python bench.py --train-only --no-cpu-baseline --steps 40 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('DX_ACCUM=0', r['ms_per_step'])" >> gpurun_out/r5s_ab.txt
python bench.py --train-only --no-cpu-baseline --steps 40 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('default  ', r['ms_per_step'])" >> gpurun_out/r5s_ab.txt
done
