cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
ADVOC_P4W_DEBUG=1 ADVOC_H3_PATCH_2WG=1 python tools/layer_times.py regular 64 2>&1 | grep -v amdgpu.ids > /tmp/l1.txt
grep "p4w:" /tmp/l1.txt | sort | uniq -c | head
ADVOC_H3_PATCH_2WG=0 python tools/layer_times.py regular 64 2>&1 | grep -v amdgpu.ids > /tmp/l0.txt
ADVOC_H3_PATCH_2WG=2 python tools/layer_times.py regular 64 2>&1 | grep -v amdgpu.ids > /tmp/l2.txt
for f in /tmp/l0.txt /tmp/l1.txt /tmp/l2.txt; do echo == $f; grep "patch_gemm_h3_kernel<[46]" $f; done > gpurun_out/r6h_layers.txt
cat gpurun_out/r6h_layers.txt
