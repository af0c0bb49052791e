cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 3300 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r6q_gpu_tests.txt
cat gpurun_out/r6q_gpu_tests.txt
timeout 600 python bench.py --no-cpu-baseline --train-only --steps 20 2>/dev/null | python -c "import json,sys; r=json.load(sys.stdin); print('ms_per_step', r['ms_per_step'])"
