cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for w in 0 3 0 3; do
ADVOC_H3_PATCH_2WG=$w python tools/layer_times.py regular 64 > /tmp/l.txt 2>&1
echo "== 2WG=$w"; grep "encoder_5 \|decoder_5 " /tmp/l.txt | grep -v bwdW | sort | cut -c1-100; grep "^total" /tmp/l.txt
done > gpurun_out/r6t_small_grid_p4w.txt 2>&1
cat gpurun_out/r6t_small_grid_p4w.txt
ADVOC_H3_PATCH_2WG=3 timeout 900 python -m pytest tests/test_hip_fullsize.py -x -q -m gpu -k "test_full_model_train_loops_at_bench_size" 2>&1 | tail -3
