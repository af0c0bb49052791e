cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
for i in 1 2; do
ADVOC_DX_BOUNDED=0 ADVOC_Y_IMAGE_ONLY=0 python bench.py --train-only --no-cpu-baseline --steps 40 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('r5 image mechanisms OFF', r['ms_per_step'])" >> gpurun_out/r5p_ab.txt
python bench.py --train-only --no-cpu-baseline --steps 40 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('default', r['ms_per_step'])" >> gpurun_out/r5p_ab.txt
done
