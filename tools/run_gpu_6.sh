cd $GRAFT_REPO_ROOT
( time timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err ) 2>&1 | tail -3
python -c "
import json; r=json.load(open('gpurun_out/bench_default.json'))
print({k: r[k] for k in ('value','ms_per_step','steps')})
print('roofline', {k: r['roofline'][k] for k in ('kernel','achieved','peak','frac','share_of_step','avg_launch_ms','traffic')})
for k in r['roofline']['kernels'][:12]: print('   ', k['kernel'], round(k['share_of_conv_stack'],3), k['bound'], round(k['achieved'],1), round(k['frac'],3))
print('extractor', r['extractor']['frac'], r['extractor']['at_train_feed']['frac'], r['extractor']['triple']['frac'])
print('inference', {k:v for k,v in r['inference'].items() if k not in ('note','joint_sc09','lws_note')})
print('small', r['small']); print('loader', r['loader']['value']); print('cpu', r['cpu_baseline'])
"
bash tools/run_gpu_prof.sh 2>&1 | tail -70
