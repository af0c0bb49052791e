cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
for n in $1; do
  ADVOC_HIP_LIB=$PWD/advoc_amd/csrc/libadvoc_hip_abl$n.so timeout 300 python tools/micro/power_throttle_probe.py 0 > /tmp/ab.txt 2>&1
  python - $n <<'PY'
import re,sys
v=[(int(m.group(1)),int(m.group(2))) for m in (re.match(r'clk <1,0,0,0,0> wg\s+\d+: (\d+) cycles in (\d+) ticks',l) for l in open('/tmp/ab.txt')) if m]
v=v[len(v)//4:]
if v:
  c=sum(a for a,_ in v)/len(v); t=sum(b for _,b in v)/len(v)
  print('| %s | %.4g | %.0f | %.3f |' % (sys.argv[1],c,t/100,0.1*c/t))
else: print(sys.argv[1],'failed',open('/tmp/ab.txt').read()[-300:])
PY
done
