cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out; rm -f gpurun_out/r5w_clk.txt
ADVOC_HIP_LIB=$PWD/advoc_amd/csrc/libadvoc_hip_stg3.so timeout 900 python -m pytest tests/test_hip_conv.py -m gpu -x -q -k "patch_kernels or a_priori or sole_reader" > gpurun_out/r5w_tests_stg3.txt 2>&1
tail -2 gpurun_out/r5w_tests_stg3.txt
ADVOC_HIP_LIB=$PWD/advoc_amd/csrc/libadvoc_hip_stg3.so timeout 300 python tools/micro/patch_repeat.py 20 d4b:f d4b:d > gpurun_out/r5w_repeat_stg3.txt 2>&1
tail -3 gpurun_out/r5w_repeat_stg3.txt
for v in clk clkstg1 clkstg2 clkstg3 clk clkstg1 clkstg2 clkstg3; do
  ADVOC_HIP_LIB=$PWD/advoc_amd/csrc/libadvoc_hip_$v.so timeout 600 python bench.py --train-only --no-cpu-baseline --steps 4 --warmup 3 --prof-steps 0 > /tmp/clk_$v.txt 2>&1
  echo "== $v" >> gpurun_out/r5w_clk.txt
  python tools/clock_summary.py /tmp/clk_$v.txt | grep "kernel<1," >> gpurun_out/r5w_clk.txt
done
cat gpurun_out/r5w_clk.txt
