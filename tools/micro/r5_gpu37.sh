cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_hip_conv.py -m gpu -x -q -k "weight_gradient or k_slices or two_streams or all_directions or random_layer or operand_image" 2>&1 | tail -3
for v in clk clkrow clk clkrow; do
  ADVOC_HIP_LIB=$PWD/advoc_amd/csrc/libadvoc_hip_$v.so timeout 600 python bench.py --train-only --no-cpu-baseline --steps 4 --warmup 3 --prof-steps 0 > /tmp/clk_$v.txt 2>&1
  echo "== $v"
  python tools/clock_summary.py /tmp/clk_$v.txt | grep "wgrad" | awk -F'|' '{printf "%s %s cyc n %s  %s GHz %s us\n",$2,$3,$4,$5,$8}' | awk '$5>=5'
done
timeout 900 bash tools/micro/env_ab2.sh "ADVOC_WGRAD_H3_ROWS=0" "ADVOC_WGRAD_H3_ROWS=1" "ADVOC_WGRAD_H3_ROWS=0" "ADVOC_WGRAD_H3_ROWS=1" > gpurun_out/r5_rows_ab.txt 2>&1
head -8 gpurun_out/r5_rows_ab.txt | cut -c1-200
