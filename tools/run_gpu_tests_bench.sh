cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/tb
timeout 1500 python -m pytest tests/test_hip_conv.py tests/test_hip_fullsize.py tests/test_hip_model.py -m gpu -q -x 2>&1 | tail -4
timeout 900 python bench.py --no-cpu-baseline --train-only > gpurun_out/tb/bench.json 2> gpurun_out/tb/bench.err; echo bench rc=$?
python - <<'PY'
import json
d=json.load(open('gpurun_out/tb/bench.json'))
print(d['value'], d['ms_per_step'])
for k in d['roofline']['kernels'][:24]:
    print('%-52s %5.1f/step %7.3f ms/step %8.3f ms %7.1f %s frac %.3f' % (k['kernel'][:52], k['launches_per_step'], k['launches_per_step']*k['avg_launch_ms'], k['avg_launch_ms'], k['achieved'], k['unit'], k['frac']))
PY
