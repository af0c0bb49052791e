cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
V=$1
ADVOC_HIP_LIB=$PWD/advoc_amd/csrc/libadvoc_hip_$V.so timeout 900 python -m pytest tests/test_hip_conv.py -m gpu -x -q -k "patch_kernels or a_priori or sole_reader" 2>&1 | tail -2
ADVOC_HIP_LIB=$PWD/advoc_amd/csrc/libadvoc_hip_$V.so timeout 300 python tools/micro/patch_repeat.py 10 d4b:f d4b:d 2>&1 | grep -v "^clk" | tail -2
for v in clk $V clk $V; do
  ADVOC_HIP_LIB=$PWD/advoc_amd/csrc/libadvoc_hip_$v.so timeout 600 python bench.py --train-only --no-cpu-baseline --steps 4 --warmup 3 --prof-steps 0 > /tmp/clk_$v.txt 2>&1
  echo "== $v"
  python tools/clock_summary.py /tmp/clk_$v.txt | grep "kernel<1," | awk -F'|' '{printf "%s %s cyc n %s  %s GHz %s us\n",$2,$3,$4,$5,$8}'
done
