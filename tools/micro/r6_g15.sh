cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_hip_bench.py tests/test_hip_parallel.py tests/test_hip_rccl_single.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r6o_tests.txt
cat gpurun_out/r6o_tests.txt
timeout 900 python bench.py --no-cpu-baseline --steps 20 > gpurun_out/r6o_bench.json 2> gpurun_out/r6o_bench.err; tail -3 gpurun_out/r6o_bench.err
python - <<'PY'
import json
r=json.load(open('gpurun_out/r6o_bench.json'))
print('ms', r['ms_per_step'])
ro=r['roofline']; print({k:ro.get(k) for k in ('kernel','frac','clock_ghz','frac_of_peak_at_this_clock','traffic','algorithmic_bytes_per_launch','launches','clock_launches_probed')})
e=r['extractor']; print('stft', e['frac'], e['buffer_sets_rotated'], e['at_train_feed']['frac'], e['at_train_feed']['buffer_sets_rotated'])
t=e['triple']; print('triple bulk', t['frac'], t['buffer_sets_rotated'], 'feed', t['at_train_feed']['frac'], t['at_train_feed']['buffer_sets_rotated'], t['at_train_feed']['avg_ms'], 'in_step', t.get('in_step'))
print([ (k['kernel'],k['launches_per_step'],round(k['frac'],3)) for k in ro['kernels'] if 'wgrad' in k['kernel']])
PY
