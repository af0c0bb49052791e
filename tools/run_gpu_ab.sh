cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for rep in 1 2; do
for lib in advoc_amd/csrc/ab/libadvoc_hip_old.so advoc_amd/csrc/libadvoc_hip.so; do
echo "== $lib"
ADVOC_HIP_LIB=$PWD/$lib timeout 600 python tools/micro/patch_sweep.py enc2m:d enc3m:d d2m:d d3m:d dec2m:fd dec3m:d d4:fd enc2m:f 2>&1 | cut -c1-60,100-200
done
done
