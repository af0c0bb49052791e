cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
for v in clk clkstg2 clk clkstg2; do
  ADVOC_HIP_LIB=$PWD/advoc_amd/csrc/libadvoc_hip_$v.so timeout 600 python bench.py --train-only --no-cpu-baseline --steps 4 --warmup 3 --prof-steps 0 > /tmp/clk_$v.txt 2>&1
  echo "== $v" >> gpurun_out/r5v_clk.txt
  python tools/clock_summary.py /tmp/clk_$v.txt | grep "kernel<1," >> gpurun_out/r5v_clk.txt
done
cat gpurun_out/r5v_clk.txt
