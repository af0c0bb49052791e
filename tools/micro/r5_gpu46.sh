cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_hip_conv.py -m gpu -x -q -k "row_mode or weight_gradient or k_slices or two_streams or random_layer" 2>&1 | tail -4
timeout 900 bash tools/micro/env_ab2.sh "ADVOC_WGRAD_H3_ROWS=0" "ADVOC_WGRAD_H3_ROWS=1" "ADVOC_WGRAD_H3_ROWS=0" "ADVOC_WGRAD_H3_ROWS=1" > gpurun_out/r5_rows_ab.txt 2>&1
grep "ms_per_step\|wgrad" gpurun_out/r5_rows_ab.txt | cut -c1-120
ADVOC_HIP_LIB=$PWD/advoc_amd/csrc/libadvoc_hip_clk.so timeout 600 python bench.py --train-only --no-cpu-baseline --steps 4 --warmup 3 --prof-steps 0 > /tmp/clk.txt 2>&1
python tools/clock_summary.py /tmp/clk.txt | grep wgrad | head -8
