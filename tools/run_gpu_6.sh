cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -12
( time timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err ) 2>&1 | tail -3
tail -3 gpurun_out/bench_default.err; python -c "
import json; r=json.load(open('gpurun_out/bench_default.json'))
print({k: r[k] for k in ('value','ms_per_step','steps')})
print('roofline', {k: r['roofline'][k] for k in ('kernel','achieved','peak','frac','share_of_step','avg_launch_ms')})
print('extractor', r['extractor'])
print('inference', {k:v for k,v in r['inference'].items() if k!='note' and k!='joint_sc09'})
print('joint', r['inference']['joint_sc09'])
print('small', r['small']); print('loader', r['loader']); print('cpu', r['cpu_baseline'])
"
