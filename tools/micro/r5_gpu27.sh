cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=gpurun_out/r5y_ablate_cycles.txt
echo "D layer_4 forward, 64 images, patch_gemm_h3_kernel<1,0> (clock-probe build; 1.573 M matrix-pipe cycles per SIMD): workgroup life in SHADER CYCLES with parts of the K loop COMPILED OUT (-DADVOC_P3_ABL=n, tools/micro/build_ablations.sh; results are garbage, the schedule of what is left is the compiler's; bits: 1 no DMA, 2 no MFMA, 4 waits without the barrier, 64 no epilogue, 256 no A fragment reads, 512 no B fragment reads)" > $OUT
echo "| ablate | what is left out | cycles | us | GHz |" >> $OUT
echo "|---|---|---|---|---|" >> $OUT
run() {
  ADVOC_HIP_LIB=$PWD/advoc_amd/csrc/libadvoc_hip_abl$1.so timeout 300 python tools/micro/power_throttle_probe.py 0 > /tmp/ab.txt 2>&1
  python - "$1" "$2" <<'PY' >> gpurun_out/r5y_ablate_cycles.txt
import re,sys
v=[(int(m.group(1)),int(m.group(2))) for m in (re.match(r'clk <1,0,0,0,0> wg\s+\d+: (\d+) cycles in (\d+) ticks',l) for l in open('/tmp/ab.txt')) if m]
v=v[len(v)//4:]
if v:
  c=sum(a for a,_ in v)/len(v); t=sum(b for _,b in v)/len(v)
  print('| %s | %s | %.4g | %.0f | %.3f |' % (sys.argv[1],sys.argv[2],c,t/100,0.1*c/t))
else:
  print('| %s | %s | failed |' % (sys.argv[1],sys.argv[2]), open('/tmp/ab.txt').read()[-300:])
PY
}
run 0 "nothing (= the product kernel)"
run 64 "epilogue"
run 4 "barriers (waits kept)"
run 1 "DMA"
run 5 "DMA, barriers"
run 256 "A fragment reads"
run 512 "B fragment reads"
run 768 "all fragment reads"
run 769 "fragment reads, DMA"
run 773 "fragment reads, DMA, barriers: MFMAs + epilogue"
run 837 "everything but the MFMAs"
run 2 "MFMAs"
run 66 "MFMAs, epilogue"
run 0 "nothing (again)"
cat $OUT
