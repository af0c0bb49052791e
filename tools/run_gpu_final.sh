cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/final/pytest.log 2>&1; echo pytest rc=$?; tail -4 gpurun_out/final/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; echo bench rc=$?; cat gpurun_out/final/bench.json | head -c 6000
bash tools/run_gpu_prof.sh
