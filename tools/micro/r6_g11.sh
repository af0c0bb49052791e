cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for cfg in "0 100" "1 0" "1 50" "1 100" "1 150" "1 200"; do set -- $cfg
ADVOC_P4W_DEBUG=1 ADVOC_H3_PATCH_2WG=$1 ADVOC_H3_PATCH_2WG_DELAY=$2 python tools/layer_times.py regular 64 > /tmp/l.txt 2>&1
echo "== 2WG=$1 delay=$2"; grep "p4w:" /tmp/l.txt | sort | uniq -c | head -3; grep "bwdD patch_gemm_h3_kernel<[46]" /tmp/l.txt | sort | awk '{printf "%s %s | ", $1,$6} END {print ""}'
done > gpurun_out/r6k_delay.txt 2>&1
cat gpurun_out/r6k_delay.txt
