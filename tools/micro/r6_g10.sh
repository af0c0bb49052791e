cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for v in "" pabl8192 pabl16 pabl8208 pabl8 pabl4 pabl768; do
for w in 0 1; do
lib=$PWD/advoc_amd/csrc/libadvoc_hip${v:+_$v}.so
ADVOC_HIP_LIB=$lib ADVOC_H3_PATCH_2WG=$w ADVOC_H3_PATCH_2WG_DELAY=0 python tools/layer_times.py regular 64 > /tmp/l.txt 2>&1
echo "== ${v:-product} 2WG=$w"; grep "bwdD patch_gemm_h3_kernel<[46]" /tmp/l.txt | sort | awk '{printf "%s %s | ", $1,$6} END {print ""}'
done; done > gpurun_out/r6j_abl_layers.txt 2>&1
cat gpurun_out/r6j_abl_layers.txt
