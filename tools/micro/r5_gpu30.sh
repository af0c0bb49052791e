cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_hip_conv.py -m gpu -x -q -k "patch or a_priori or sole_reader or producer_written or operand_image_kernels" 2>&1 | tail -3
timeout 300 python tools/micro/patch_repeat.py 10 2>&1 | tail -3
V=clkflat
for v in clk $V clk $V; do
  ADVOC_HIP_LIB=$PWD/advoc_amd/csrc/libadvoc_hip_$v.so timeout 600 python bench.py --train-only --no-cpu-baseline --steps 4 --warmup 3 --prof-steps 0 > /tmp/clk_$v.txt 2>&1
  echo "== $v"
  python tools/clock_summary.py /tmp/clk_$v.txt | grep "kernel<1," | awk -F'|' '{printf "%s %s cyc %s GHz %s us\n",$2,$3,$5,$8}'
done
