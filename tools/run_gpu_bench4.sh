cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/b10
for A in 16 8 32 16; do
ADVOC_H3_PATCH_ABLATE=$A timeout 900 python bench.py --no-cpu-baseline --train-only --steps 20 > gpurun_out/b10/bench_$A.json 2> gpurun_out/b10/bench.err
python - <<PY
import json
d=json.load(open('gpurun_out/b10/bench_$A.json'))
print('abl $A', d['value'], d['ms_per_step'], ' '.join('%s %.3f' % (k['kernel'][:26], k['avg_launch_ms']) for k in d['roofline']['kernels'][:12] if 'patch' in k['kernel']))
PY
done
