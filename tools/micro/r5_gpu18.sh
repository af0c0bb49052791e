cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_conv.py -x -q -m gpu -k "a_priori or sole_reader or patch_kernels" 2>&1 | tail -25 > gpurun_out/r5r_tests.txt
timeout 1800 python -m pytest tests/test_hip_model.py tests/test_hip_fullsize.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r5r_model_tests.txt
for i in 1 2; do python bench.py --train-only --no-cpu-baseline --steps 40 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('default', r['ms_per_step'])" >> gpurun_out/r5r_ab.txt; done
