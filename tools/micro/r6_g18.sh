cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 3000 python -m pytest tests/test_hip_fullsize.py tests/test_hip_model.py tests/test_hip_spectral.py tests/test_hip_lws.py tests/test_hip_inversion.py tests/test_hip_melspecgan.py tests/test_hip_numerics.py tests/test_infer.py tests/test_loader.py tests/test_cli.py -q -m gpu -s 2>&1 | grep "gates that differ\|generator output\|passed\|failed\|FAILED\|gate-frozen" | tail -30 > gpurun_out/r6r_fullsize.txt
cat gpurun_out/r6r_fullsize.txt
