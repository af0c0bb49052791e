cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out; rm -f gpurun_out/r5_unr.txt
for v in $1; do
  ADVOC_HIP_LIB=$PWD/advoc_amd/csrc/libadvoc_hip_$v.so timeout 600 python bench.py --train-only --no-cpu-baseline --steps 4 --warmup 3 --prof-steps 0 > /tmp/clk_$v.txt 2>&1
  echo "== $v" >> gpurun_out/r5_unr.txt
  python tools/clock_summary.py /tmp/clk_$v.txt | grep "patch_gemm" | awk -F'|' '{printf "%s %s cyc %s n\n",$2,$3,$4}' >> gpurun_out/r5_unr.txt
done
