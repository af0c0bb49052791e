cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/tb
run() {
  env "$@" timeout 600 python bench.py --no-cpu-baseline --train-only > gpurun_out/tb/b.json 2> gpurun_out/tb/bench.err
  python -c "
import json
d=json.load(open('gpurun_out/tb/b.json'))
print('$*', d['ms_per_step'], [(k['kernel'], round(k['avg_launch_ms'],3)) for k in d['roofline']['kernels'] if 'thin_wgrad' in k['kernel']])
"
}
run X=0
run ADVOC_THIN_WGRAD_NT=2
run ADVOC_THIN_WGRAD_NT=1
run X=0
