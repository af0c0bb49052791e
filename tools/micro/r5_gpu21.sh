cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
# shader clock inside the real train step (probe build), then the whole GPU suite on the product build
ADVOC_HIP_LIB=$PWD/advoc_amd/csrc/libadvoc_hip_clk.so timeout 600 python bench.py --train-only --no-cpu-baseline --steps 4 --warmup 3 --prof-steps 0 > gpurun_out/r5_clk.txt 2>&1
grep -c "^clk" gpurun_out/r5_clk.txt
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r5_tests_full.txt 2>&1
tail -5 gpurun_out/r5_tests_full.txt
