cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
V=$1
ADVOC_HIP_LIB=$PWD/advoc_amd/csrc/libadvoc_hip_clk.so timeout 600 python bench.py --train-only --no-cpu-baseline --steps 4 --warmup 3 --prof-steps 0 2>&1 | grep -v "^clk " | tail -15
ADVOC_HIP_LIB=$PWD/advoc_amd/csrc/libadvoc_hip_$V.so timeout 900 python -m pytest tests/test_hip_conv.py -m gpu -x -q -k "patch_kernels or a_priori or sole_reader" 2>&1 | grep -v "^clk " | tail -30
