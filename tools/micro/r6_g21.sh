cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for s in 0 4 8 16 24 32; do
ADVOC_H3_DEEP_SPLIT=$s python tools/layer_times.py regular 64 > /tmp/l.txt 2>&1
echo "== DEEP_SPLIT=$s"; grep "encoder_[678] \|decoder_[678] " /tmp/l.txt | grep -v bwdW | sort | awk '{printf "%s.%s %s | ", $1,$2,$(NF-4)} END {print ""}'; grep "^total" /tmp/l.txt
done > gpurun_out/r6u_deep_split.txt 2>&1
cat gpurun_out/r6u_deep_split.txt
