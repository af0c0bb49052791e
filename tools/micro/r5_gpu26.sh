cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f /tmp/pt.txt; for c in 0 1 2 3 0 1 2 3; do ADVOC_HIP_LIB=$PWD/advoc_amd/csrc/libadvoc_hip_clk.so timeout 600 python tools/micro/power_throttle_probe.py $c > /tmp/pt1.txt 2>&1; grep CASE /tmp/pt1.txt >> /tmp/pt.txt; grep '^clk' /tmp/pt1.txt >> /tmp/pt.txt; done
python - <<'PY' > gpurun_out/r5x_power_throttle.txt
import re
case=None; rows={}
for l in open('/tmp/pt.txt'):
  if l.startswith('CASE'): case=l[5:].strip(); rows.setdefault(case,[])
  m=re.match(r'clk <1,0,0,0,0> wg\s+\d+: (\d+) cycles in (\d+) ticks',l)
  if m and case: rows[case].append((int(m.group(1)),int(m.group(2))))
print('D layer_4 forward, 64 images, patch_gemm_h3_kernel<1,0>: 50.3 M MFMAs = 1.573 M pipe cycles per SIMD; one process per case, twice')
print('| operands | workgroup life, cycles | us | clock GHz | matrix pipe occupied |')
print('|---|---|---|---|---|')
for k,v in rows.items():
  c=sum(a for a,_ in v)/len(v); t=sum(b for _,b in v)/len(v)
  print('| %s | %.3g | %.0f | %.3f | %.3f |' % (k,c,t/100,0.1*c/t,1.573e6/c))
PY
cat gpurun_out/r5x_power_throttle.txt; tail -3 /tmp/pt.txt
