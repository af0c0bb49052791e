cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_conv.py -x -q -m gpu -k "test_layer_patch_kernels and two_per_cu_all" 2>&1 | tail -3
for cfg in "0 100" "1 100" "1 0" "2 0"; do set -- $cfg
ADVOC_P4W_DEBUG=1 ADVOC_H3_PATCH_2WG=$1 ADVOC_H3_PATCH_2WG_DELAY=$2 python tools/layer_times.py regular 64 > /tmp/l.txt 2>&1
echo "== 2WG=$1 delay=$2"; grep "p4w:" /tmp/l.txt | sort | uniq -c | head -3; grep "patch_gemm_h3_kernel<[46]" /tmp/l.txt | awk '{printf "%s %s %s %s | ", $1,$2,$3$4,$6} END {print ""}'; grep "^total" /tmp/l.txt
done > gpurun_out/r6i_layers.txt 2>&1
cat gpurun_out/r6i_layers.txt
