cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/b12
timeout 900 python -m pytest tests/test_hip_spectral.py tests/test_loader.py tests/test_hip_bench.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python bench.py --no-cpu-baseline --steps 10 > gpurun_out/b12/bench.json 2> gpurun_out/b12/bench.err; echo bench rc=$?
python - <<'PY'
import json
d=json.load(open('gpurun_out/b12/bench.json'))
print(d['value'], d['ms_per_step'])
print(json.dumps(d['extractor'], indent=1))
PY
