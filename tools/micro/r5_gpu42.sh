cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in clk clkslice; do echo "== $v"; ADVOC_HIP_LIB=$PWD/advoc_amd/csrc/libadvoc_hip_$v.so AMD_LOG_LEVEL=1 timeout 300 python tools/micro/dbg_slice.py 2>&1 | grep -v "^clk " | tail -12; done
