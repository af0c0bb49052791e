cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for cfg in "0 100 1" "1 100 1" "1 100 2" "1 100 4" "1 100 8" "1 60 4" "1 140 4"; do set -- $cfg
ADVOC_H3_PATCH_2WG=$1 ADVOC_H3_PATCH_2WG_DELAY=$2 ADVOC_H3_PATCH_2WG_WAYS=$3 python tools/layer_times.py regular 64 > /tmp/l.txt 2>&1
echo "== 2WG=$1 delay=$2 ways=$3"; grep "bwdD patch_gemm_h3_kernel<[46]" /tmp/l.txt | sort | awk '{printf "%s %s | ", $1,$6} END {print ""}'
done > gpurun_out/r6l_ways.txt 2>&1
cat gpurun_out/r6l_ways.txt
