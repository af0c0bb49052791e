cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out; rm -f gpurun_out/r5_wabl2.txt
for v in $1; do
  ADVOC_HIP_LIB=$PWD/advoc_amd/csrc/libadvoc_hip_$v.so timeout 600 python bench.py --train-only --no-cpu-baseline --steps 4 --warmup 3 --prof-steps 0 > /tmp/clk_$v.txt 2>&1
  echo "== $v" >> gpurun_out/r5_wabl2.txt
  python tools/clock_summary.py /tmp/clk_$v.txt | grep "wgrad\|kernel<1,0" | awk -F'|' '{printf "%s %s cyc n %s  %s GHz %s us\n",$2,$3,$4,$5,$8}' | awk '$5>=14' | head -4 >> gpurun_out/r5_wabl2.txt
done
cat gpurun_out/r5_wabl2.txt
