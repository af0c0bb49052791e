set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_conv.py -x -q -m gpu -k "sole_reader or a_priori or producer_written or thin_producers" 2>&1 | tail -25 > gpurun_out/r5j_newtest.txt
timeout 1500 python -m pytest tests/test_hip_model.py tests/test_hip_fullsize.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r5j_model_tests.txt
for i in 1 2; do
ADVOC_Y_IMAGE_ONLY=0 python bench.py --train-only --no-cpu-baseline --steps 40 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('Y_IMAGE_ONLY=0', r['ms_per_step'])" >> gpurun_out/r5j_ab.txt
python bench.py --train-only --no-cpu-baseline --steps 40 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('Y_IMAGE_ONLY=1', r['ms_per_step'])" >> gpurun_out/r5j_ab.txt
done
