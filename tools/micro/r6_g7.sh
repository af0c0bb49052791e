cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out; rm -f gpurun_out/r6g_cycles.txt
for cfg in "clk 0" "clk 1" "abl8192 1" "clk 2" "abl8192 2"; do
  set -- $cfg
  ADVOC_H3_PATCH_2WG=$2 ADVOC_HIP_LIB=$PWD/advoc_amd/csrc/libadvoc_hip_$1.so timeout 600 python bench.py --train-only --no-cpu-baseline --steps 4 --warmup 3 --prof-steps 0 > /tmp/clk.txt 2>&1
  echo "== $1 2WG=$2" >> gpurun_out/r6g_cycles.txt
  python tools/clock_summary.py /tmp/clk.txt | grep "patch_gemm_h3_kernel<[46]" | awk -F'|' '{printf "%s %s cyc %s n %s GHz %s us\n",$2,$3,$4,$5,$8}' >> gpurun_out/r6g_cycles.txt
done
cat gpurun_out/r6g_cycles.txt
