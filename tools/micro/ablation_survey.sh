# Ablation survey inside the real train step: cycles per patch instance under compile-time ablation libraries
#   bash tools/micro/build_variant.sh clk "-DADVOC_CLOCK_PROBE"; bash tools/micro/build_ablations.sh "1 8 16"
#   gpurun -- bash tools/micro/ablation_survey.sh "clk abl1 abl8 abl16"   -> gpurun_out/r5z_abl_all.txt
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out; rm -f gpurun_out/r5z_abl_all.txt
for v in $1; do
  ADVOC_HIP_LIB=$PWD/advoc_amd/csrc/libadvoc_hip_$v.so timeout 600 python bench.py --train-only --no-cpu-baseline --steps 4 --warmup 3 --prof-steps 0 > /tmp/clk_$v.txt 2>&1
  echo "== $v" >> gpurun_out/r5z_abl_all.txt
  python tools/clock_summary.py /tmp/clk_$v.txt | grep "patch_gemm" | awk -F'|' '{printf "%s %s cyc %s n %s GHz %s us\n",$2,$3,$4,$5,$8}' >> gpurun_out/r5z_abl_all.txt
done
cat gpurun_out/r5z_abl_all.txt
