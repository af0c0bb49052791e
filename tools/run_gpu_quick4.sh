cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/tb
timeout 1200 python -m pytest tests/test_hip_conv.py tests/test_hip_model.py -m gpu -q -x  2>&1 | tail -5
for on in 1 0 1 0; do
ADVOC_WEIGHT_IMAGES=$on timeout 600 python bench.py --no-cpu-baseline --train-only --prof-steps 0 > gpurun_out/tb/bench_h$on.json 2> gpurun_out/tb/bench.err; echo on=$on rc=$?
python -c "
import json
d=json.load(open('gpurun_out/tb/bench_h$on.json'))
print(d['value'], d['ms_per_step'])
"
done
